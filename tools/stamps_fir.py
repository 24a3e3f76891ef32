"""Per-phase cycles of fir_ols_pk32_kernel (C4: 8192 taps, 8 x 2 880 000 samples) per block and wave, from a `make STAMPS=1` build:
  python tools/stamps_fir.py"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "soundml_amd", "lib_stamps", "libsoundml_amd.so"))
i64, vp, ci, f64 = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_double
ch, ns, ntaps = int(os.environ.get("FIR_CH", "8")), 2880000, 8192
x = torch.empty(ch, ns, device="cuda").uniform_(-1, 1)
y = torch.empty_like(x)
taps, beta, plan = (f64 * ntaps)(), f64(), vp()
lib.smx_fir_kaiser_beta.argtypes = [f64, ctypes.POINTER(f64)]
assert lib.smx_fir_kaiser_beta(100.0, ctypes.byref(beta)) == 0
lib.smx_fir_design_lowpass.argtypes = [i64, f64, f64, vp]
assert lib.smx_fir_design_lowpass(ntaps, 0.25, beta.value, taps) == 0
lib.smx_fir_plan_create.argtypes = [vp, i64, ctypes.POINTER(vp)]
assert lib.smx_fir_plan_create(taps, ntaps, ctypes.byref(plan)) == 0
lib.smx_fir_apply_f32_dev.argtypes = [vp, vp, i64, i64, i64, vp, i64, vp]
run = lambda: lib.smx_fir_apply_f32_dev(plan, vp(x.data_ptr()), ch, ns, ns, vp(y.data_ptr()), ns, None)
for _ in range(20): assert run() == 0
torch.cuda.synchronize()
ev = []
for _ in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize(); ev.append(a.elapsed_time(b))
S, nwg = 24, 256
buf = np.zeros(nwg * 16 * S, dtype=np.uint64)
assert lib.smx_debug_read_stamps_fir(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), buf.size) == 0
st = buf.reshape(nwg, 16, S).astype(np.float64)[:, :8, :]
names = ["radix-16 across the block + LDS writes", "wait: barrier 1", "forward sub-transform (reads .. second radix-32)", "exchange 1, pairs, exchange 2",
         "inverse sub-transform + LDS writes + next block's requests", "wait: barrier 2", "last pass's LDS reads + barrier 3", "twiddles + radix-16", "stores"]
blocks = st[:, 0, 22]
four = blocks >= 4
mean = st[four].mean(axis=(0, 1))
nb = blocks[four].mean()
tot = mean[:9].sum()
print("launch %.4f ms (median of 8); workgroups with %.0f blocks: %d of %d; %.0f stamped ticks per block and wave; clock %.2f GHz; loop %.1f us"
      % (sorted(ev)[4], nb, int(four.sum()), nwg, tot / nb, mean[20] / mean[21] / 10.0, mean[21] / 100.0))
for i, nm in enumerate(names):
    print("  %-60s %8.0f per block  %5.1f %%   (slowest wave %.0f, fastest %.0f)" % (nm, mean[i] / nb, 100 * mean[i] / tot, st[four][:, :, i].mean(axis=0).max() / nb, st[four][:, :, i].mean(axis=0).min() / nb))

"""Diagnostic: per-phase cycle shares of the fused STFT kernel (stamps build, make STAMPS=1)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "soundml_amd", "lib_stamps", "libsoundml_amd.so"))
i64, vp = ctypes.c_int64, ctypes.c_void_p
h = vp()
lib.smx_stft_config_create.argtypes = [i64, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(vp)]
assert lib.smx_stft_config_create(2048, -(2**63), 512, 0, 0, 0.0, 0, 0, None, ctypes.byref(h)) == 0
clips, n = 256, 480000
x = torch.rand(clips, n, device="cuda") * 2 - 1
out = torch.empty(clips, 1025, 938, device="cuda")
lib.smx_stft_power_range_f32_dev.argtypes = [vp, vp, i64, i64, i64, i64, i64, ctypes.c_double, vp, vp]
for _ in range(2):
    # interior frames only (2..936): one launch of the fused kernel
    rc = lib.smx_stft_power_range_f32_dev(h, vp(x.data_ptr()), clips, n, n, 2, 936, 2.0, vp(out.data_ptr()), None)
    assert rc == 0
torch.cuda.synchronize()
ev = []
for _ in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.smx_stft_power_range_f32_dev(h, vp(x.data_ptr()), clips, n, n, 2, 936, 2.0, vp(out.data_ptr()), None)
    e1.record(); torch.cuda.synchronize(); ev.append(e0.elapsed_time(e1))
wall_ms = sorted(ev)[len(ev) // 2]
S = 24
nwg = 256
buf = np.zeros(nwg * 16 * S, dtype=np.uint64)
assert lib.smx_debug_read_stamps(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), buf.size) == 0
st = buf.reshape(nwg, 16, S).astype(np.float64)
tot = st.sum(axis=2)
names = ["loop top", "window", "A pass1", "A pass2", "A twiddle", "X re (+ready)", "X im", "B pass1", "B pass2", "B twiddle",
         "C q<8", "C q>=8", "P q0-3 (+ready)", "P q4-7", "P q8-11", "P q12-15", "(hook15)", "nyquist+signal", "prefetch issue", "wait filled+flush"]
mean = st.mean(axis=(0, 1))
tiles = 256 * 934 / 16 / nwg
print("SMX_ABLATE=%s  wall %.3f ms per launch; s_memtime ticks per wave: total %.0f over %.1f tiles (%.0f per tile); ticks / wall = %.0f MHz"
      % (os.environ.get("SMX_ABLATE", "0"), wall_ms, tot.mean(), tiles, tot.mean() / tiles, tot.mean() / wall_ms / 1e3))
for i, nm in enumerate(names):
    print("  %-16s %9.0f  %5.1f%%   (per tile %.0f)" % (nm, mean[i], 100 * mean[i] / mean.sum(), mean[i] / tiles))
print("per-wave totals (mean over WGs):", np.round(tot.mean(axis=0) / tiles))
print("per-wave flush phase:", np.round(st[:, :, 19].mean(axis=0) / tiles))

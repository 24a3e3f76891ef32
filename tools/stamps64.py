"""Diagnostic: per-phase cycle sums of stft2048_power_wide_kernel (the float64 interior at fft 2048; stamps build: make STAMPS=1)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "soundml_amd", "lib_stamps", "libsoundml_amd.so"))
i64, vp = ctypes.c_int64, ctypes.c_void_p
h = vp()
lib.smx_stft_config_create.argtypes = [i64, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(vp)]
assert lib.smx_stft_config_create(2048, -(2**63), 512, 0, 0, 0.0, 0, 0, None, ctypes.byref(h)) == 0
assert lib.smx_set_interior(1) == 0
clips, n = 256, 480000
frames = 1 + n // 512
x = torch.rand(clips, n, device="cuda") * 2 - 1
out = torch.empty(clips, 1025, frames, device="cuda")
lib.smx_stft_power_range_f32_dev.argtypes = [vp, vp, i64, i64, i64, i64, i64, ctypes.c_double, vp, vp]
def run():
    assert lib.smx_stft_power_range_f32_dev(h, vp(x.data_ptr()), clips, n, n, 0, frames, 2.0, vp(out.data_ptr()), None) == 0
for _ in range(2): run()
torch.cuda.synchronize()
ev = []
for _ in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize(); ev.append(e0.elapsed_time(e1))
S, nwg = 24, 256
buf = np.zeros(nwg * 16 * S, dtype=np.uint64)
assert lib.smx_debug_read_stamps64(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), buf.size) == 0
st = buf.reshape(nwg, 16, S).astype(np.float64)[:, :8, :]
names = ["0 loop top", "1 window (LDS table, samples arrive)", "2 pass 1 (radix 16)", "3 exchange 1", "4 pass 2 (radix 16 + twiddle powers)",
         "5 request + exchange 2", "6 pass 3 (radix 4)", "7 partner exchange (x2)", "8 post-pass + |.|^p (x2)", "9 results", "10 flush (waits, reads, stores)", "11 loop tail"]
mean = st.mean(axis=(0, 1))
slots = 2 * 256 * ((frames + 15) // 16) / nwg
print("wall %.3f ms; %.1f frames per wave; loop %.0f ticks per frame-slot (clock %.2f GHz)" %
      (sorted(ev)[len(ev) // 2], slots, mean[20] / slots, mean[20] / mean[21] / 10.0 if mean[21] else 0))
for i, nm in enumerate(names):
    print("  %-48s %8.0f per frame  %5.1f %%" % (nm, mean[i] / slots, 100 * mean[i] / mean[:12].sum()))
print("  inside the counter waits, per tile: filled %.0f, drained %.0f" % (2 * mean[12] / slots, 2 * mean[13] / slots))

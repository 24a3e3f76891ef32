"""Diagnostic: per-phase cycle shares of stft2048_power32_kernel (stamps build, make STAMPS=1)."""
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
def run():
    assert lib.smx_stft_power_range_f32_dev(h, vp(x.data_ptr()), clips, n, n, 2, 934, 2.0, vp(out.data_ptr()), None) == 0
for _ in range(2): run()
torch.cuda.synchronize()
ev = []
for _ in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize(); ev.append(e0.elapsed_time(e1))
wall_ms = sorted(ev)[len(ev) // 2]
S, nwg = 24, 256
buf = np.zeros(nwg * 16 * S, dtype=np.uint64)
assert lib.smx_debug_read_stamps(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), buf.size) == 0
st = buf.reshape(nwg, 16, S).astype(np.float64)[:, :8, :]
names = ["0 loop top", "1 window", "2 A radix-32 + twiddle", "3 wait drained", "4 transposition issue", "5 B radix-32 (+ wait for the transposition)",
         "6 exchange issue, wait filled, flush reads", "7 post-pass + results (+ flush stores, prefetch issue)", "8 signal"]
mean = st.mean(axis=(0, 1))
tiles = 256 * 59 / nwg   # 934 frames -> 59 tiles per clip
tot = mean[:9].sum()
print("wall %.3f ms per launch; ticks per wave in the loop %.0f over %.1f tiles (%.0f per tile); whole loop %.0f ticks, %.0f x 10 ns -> clock %.2f GHz"
      % (wall_ms, tot, tiles, tot / tiles, mean[20], mean[21], mean[20] / mean[21] / 10.0 if mean[21] else 0))
print("  drained wait per tile by wave:", " ".join("%.0f" % (st[:, w, 12].mean() / tiles) for w in range(st.shape[1])))
print("  whole loop ticks by wave:     ", " ".join("%.0f" % (st[:, w, 20].mean() / tiles) for w in range(st.shape[1])))
t14 = st[:, :, 14]
rel = t14 - t14.min(axis=1, keepdims=True)
print("  start of tile 33 relative to the workgroup's first wave, cycles, mean by wave:", " ".join("%.0f" % rel[:, w].mean() for w in range(st.shape[1])), " (max spread mean %.0f)" % rel.max(axis=1).mean())
print("  inside the counter waits: drained %.0f per tile, filled %.0f per tile" % (mean[12] / tiles, mean[13] / tiles))
for i, nm in enumerate(names):
    print("  %-24s %9.0f  %5.1f%%   (per tile %.0f)" % (nm, mean[i], 100 * mean[i] / tot, mean[i] / tiles))

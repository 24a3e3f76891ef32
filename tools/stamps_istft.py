"""Per-phase cycles of istft2048_pipe_kernel (Stft.invert at C2) per tile and wave, from a `make STAMPS=1` build:
  python tools/stamps_istft.py"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "soundml_amd", "lib_stamps", "libsoundml_amd.so"))
i64, vp, ci = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
h = vp()
lib.smx_stft_config_create.argtypes = [i64, i64, i64, ci, ci, ctypes.c_double, ci, ci, vp, ctypes.POINTER(vp)]
assert lib.smx_stft_config_create(2048, -(2**63), 512, 0, 0, 0.0, 0, 0, None, ctypes.byref(h)) == 0
clips, n, frames = 256, 480000, 938
z = torch.randn(clips, 1025, frames, 2, device="cuda")
out = torch.empty(clips, n, device="cuda")
f = lib.smx_stft_invert_f32_dev
f.argtypes = [vp, vp, i64, i64, i64, ci, i64, vp, vp]
def run():
    assert f(h, vp(z.data_ptr()), clips, 1025, frames, 1, n, vp(out.data_ptr()), None) == 0
for _ in range(20): run()
torch.cuda.synchronize()
ev = []
for _ in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize(); ev.append(a.elapsed_time(b))
S, nwg = 24, 256
buf = np.zeros(nwg * 16 * S, dtype=np.uint64)
assert lib.smx_debug_read_stamps_istft(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), buf.size) == 0
st = buf.reshape(nwg, 16, S).astype(np.float64)[:, :8, :]
tiles = st[:, 0, 22].mean()
names = ["loop edge", "GL factors", "wait: slots free (drained)", "staging stores (+ arrival of the staged registers)", "next tile's requests issued",
         "wait: all staged", "two frames (pre-pass .. samples)", "wait: all frames in (filled)", "overlap-add reads + sums", "envelope + stores"]
mean = st.mean(axis=(0, 1))
tot = mean[:10].sum()
print("launch %.4f ms (median of 8); %.1f tiles per workgroup; %.0f stamped ticks per tile and wave (s_memtime, 100 MHz x clock ratio); clock %.2f GHz"
      % (sorted(ev)[4], tiles, tot / tiles, mean[20] / mean[21] / 10.0))
for i, nm in enumerate(names):
    print("  %-52s %8.0f per tile  %5.1f %%   (slowest wave %.0f)" % (nm, mean[i] / tiles, 100 * mean[i] / tot, st[:, :, i].mean(axis=0).max() / tiles))

"""Is a workgroup's tile time tied to its CU or to the addresses it works on?  (round 5: the slowest workgroup of a C2 launch runs its
tiles 4 % slower than the mean and sets the launch's end.)  A `make COARSE=1` build; three launches of the fft-2048 power kernel on 255
clips: (a) clips 0..254, (b) the same again, (c) clips 1..255 of the same arrays -- workgroup w then works on the addresses workgroup
w + 1 had.  Prints the correlation of the per-workgroup steady tile times between the launches, as launched and shifted by one.
  python tools/wg_speed_probe.py"""
import ctypes, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "soundml_amd", "lib_clock", "libsoundml_amd.so"))
i64, vp = ctypes.c_int64, ctypes.c_void_p
h = vp()
lib.smx_stft_config_create.argtypes = [i64, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(vp)]
assert lib.smx_stft_config_create(2048, -(2**63), 512, 0, 0, 0.0, 0, 0, None, ctypes.byref(h)) == 0
lib.smx_stft_power_range_f32_dev.argtypes = [vp, vp, i64, i64, i64, i64, i64, ctypes.c_double, vp, vp]
clips, n = 256, 480000
frames = 1 + n // 512
x = torch.rand(clips, n, device="cuda") * 2 - 1
out = torch.empty(clips, 1025, frames, device="cuda")
def run(first):
    assert lib.smx_stft_power_range_f32_dev(h, vp(x[first:].data_ptr()), clips - 1, n, n, 0, frames, 2.0, vp(out[first:].data_ptr()), None) == 0
def tile_times(first):
    t0 = time.time()
    while time.time() - t0 < 1.0:
        for _ in range(10): run(first)
        torch.cuda.synchronize()
    run(first); torch.cuda.synchronize()
    S, nwg = 24, 256
    buf = np.zeros(nwg * 16 * S, dtype=np.uint64)
    assert lib.smx_debug_read_stamps(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), buf.size) == 0
    st = buf.reshape(nwg, 16, S)[:, :8, :].astype(np.int64)
    nt = st[:, 0, 22]
    ok = nt > 2
    per = (st[:, :, 11].max(axis=1) - st[:, :, 9].max(axis=1)) / np.maximum(nt - 2, 1) / 100.0
    return per, ok, nt
a, oka, nta = tile_times(0)
b, okb, _ = tile_times(0)
c, okc, _ = tile_times(1)
ok = oka & okb & okc
print("workgroups with a range: %d; tiles per workgroup %d..%d; steady tile us: mean %.3f, sd %.3f, max %.3f" % (ok.sum(), nta[ok].min(), nta[ok].max(), a[ok].mean(), a[ok].std(), a[ok].max()))
cc = lambda u, v: float(np.corrcoef(u, v)[0, 1])
print("same clips, two runs:            corr %.3f" % cc(a[ok], b[ok]))
print("clips shifted by one, as launched (same CU, other addresses):  corr %.3f" % cc(a[ok], c[ok]))
idx = np.where(ok)[0]
idx = idx[(idx + 1 < 256)]
idx = idx[ok[idx + 1]]
print("clips shifted by one, compared at the same ADDRESSES (workgroup w of (c) against w + 1 of (a)):  corr %.3f" % cc(a[idx + 1], c[idx]))
by_xcd = [a[ok & (np.arange(256) % 8 == k)].mean() for k in range(8)]
print("by XCD (blockIdx %% 8): " + " ".join("%.3f" % v for v in by_xcd))

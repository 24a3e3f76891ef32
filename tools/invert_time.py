"""Diagnostic: device time of Stft.invert on the C2-shaped spectrum batch (256 x 1025 x 938 complex64)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "soundml_amd", "lib", "libsoundml_amd.so"))
i64, vp, ci = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
clips, n = int(os.environ.get("CLIPS", 256)), int(os.environ.get("N", 480000))
FFT, HOP = int(os.environ.get("FFT", 2048)), int(os.environ.get("HOP", 512))
frames = 1 + n // HOP
z = torch.randn(clips, FFT // 2 + 1, frames, 2, device="cuda")
out = torch.empty(clips, n, device="cuda")
h = vp()
lib.smx_stft_config_create.argtypes = [i64, i64, i64, ci, ci, ctypes.c_double, ci, ci, vp, ctypes.POINTER(vp)]
assert lib.smx_stft_config_create(FFT, -(2**63), HOP, 0, 0, 0.0, 0, 0, None, ctypes.byref(h)) == 0
if os.environ.get("INTERIOR") == "float64":
    assert lib.smx_set_interior(1) == 0
f = lib.smx_stft_invert_f32_dev
f.argtypes = [vp, vp, i64, i64, i64, ci, i64, vp, vp]
def run():
    assert f(h, vp(z.data_ptr()), clips, FFT // 2 + 1, frames, 1, n, vp(out.data_ptr()), None) == 0
for _ in range(2): run()
torch.cuda.synchronize()
if os.environ.get("AB_ORDER"):   # interleaved A/B of the XCD-contiguous tile order against the linear one
    res = {"0": [], "1": []}
    for _ in range(10):
        for mode in ("0", "1"):
            os.environ["SMX_ISTFT_LINEAR"] = mode
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); torch.cuda.synchronize(); res[mode].append(a.elapsed_time(b))
    for mode, name in (("0", "xcd-contiguous"), ("1", "linear")):
        t = sorted(res[mode])
        print("%s: median %.3f ms  min %.3f ms" % (name, t[len(t) // 2], t[0]))
    os.environ["SMX_ISTFT_LINEAR"] = "0"
ts = []
for _ in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ts.sort()
print("invert fft %d hop %d (complex64 in, float32 out) %d clips x %d frames: median %.3f ms  min %.3f ms  %.1f Mframes/s  %.0f GB/s algorithmic"
      % (FFT, HOP, clips, frames, ts[4], ts[0], clips * frames / ts[4] / 1e3, clips * frames * ((FFT // 2 + 1) * 8 + HOP * 4) / ts[4] / 1e6))

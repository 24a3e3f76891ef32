"""Diagnostic: device time of the complex-output entry point (Stft.transform) on the C2 batch."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "soundml_amd", "lib", "libsoundml_amd.so"))
i64, vp = ctypes.c_int64, ctypes.c_void_p
clips, n = int(os.environ.get("CLIPS", 256)), 480000
frames = 1 + n // 512
x = torch.rand(clips, n, device="cuda") * 2 - 1
out = torch.empty(clips, 1025, frames, 2, device="cuda")
h = vp()
lib.smx_stft_config_create.argtypes = [i64, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(vp)]
assert lib.smx_stft_config_create(2048, -(2**63), 512, 0, 0, 0.0, 0, 0, None, ctypes.byref(h)) == 0
f = lib.smx_stft_transform_range_f32_dev
f.argtypes = [vp, vp, i64, i64, i64, i64, i64, vp, vp]
def run():
    assert f(h, vp(x.data_ptr()), clips, n, n, 0, frames, vp(out.data_ptr()), None) == 0
for _ in range(2): run()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ts.sort()
print("transform (complex64 out) %d clips: median %.3f ms  min %.3f ms  %.1f Mframes/s  %.0f GB/s algorithmic"
      % (clips, ts[5], ts[0], clips * frames / ts[5] / 1e3, clips * frames * 10248 / ts[5] / 1e6))

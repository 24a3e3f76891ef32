"""Diagnostic: device time of power_spectrum for geometries that take the generic kernels (fft != 2048)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundml_amd import Stft
from soundml_amd._lib import lib, check
vp = ctypes.c_void_p
for fft, hop, clips, n in ((1024, 256, 256, 441000), (512, 128, 256, 441000), (4096, 1024, 256, 480000), (8192, 2048, 256, 480000), (256, 64, 256, 160000), (400, 160, 256, 160000)):
    c = Stft.Config.create(fft_size=fft, hop=hop)
    frames = Stft.frames(c, n)
    x = torch.rand(clips, n, device="cuda") * 2 - 1
    out = torch.empty(clips, fft // 2 + 1, frames, device="cuda")
    def run():
        check(lib.smx_stft_power_range_f32_dev(c._h, vp(x.data_ptr()), clips, n, n, 0, frames, 2.0, vp(out.data_ptr()), None))
    for _ in range(2): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = sorted(ts)[2]
    print("fft %5d hop %4d: %d clips x %d frames in %.3f ms  (%.1f Mframes/s, %.0f GB/s algorithmic)"
          % (fft, hop, clips, frames, ms, clips * frames / ms / 1e3, clips * frames * (hop * 4 + (fft // 2 + 1) * 4) / ms / 1e6))
    del x, out

"""Diagnostic: device time of Stft.transform (complex64 out) at the generic sizes."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundml_amd import Stft
from soundml_amd._lib import lib, check
vp = ctypes.c_void_p
for fft, hop, clips, n in ((1024, 256, 256, 441000), (512, 128, 256, 441000)):
    c = Stft.Config.create(fft_size=fft, hop=hop)
    frames = Stft.frames(c, n)
    x = torch.rand(clips, n, device="cuda") * 2 - 1
    out = torch.empty(clips, fft // 2 + 1, frames, 2, device="cuda")
    def run():
        check(lib.smx_stft_transform_range_f32_dev(c._h, vp(x.data_ptr()), clips, n, n, 0, frames, vp(out.data_ptr()), None))
    for _ in range(2): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = sorted(ts)[3]
    print("transform fft %5d hop %4d: %d clips x %d frames in %.3f ms  (%.1f Mframes/s, %.0f GB/s algorithmic)"
          % (fft, hop, clips, frames, ms, clips * frames / ms / 1e3, clips * frames * (hop * 4 + (fft // 2 + 1) * 8) / ms / 1e6))
    del x, out

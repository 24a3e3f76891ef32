"""fft 1024 / 512 power spectrogram, 256 clips of C1's length: the flush in whole aligned 128-byte lines (default) against the plain
per-tile flush (SMX_POWER_SKEW=0), sustained and interleaved: one second of launches first, then blocks of 60 launches alternating.
  python tools/ab_lanes_skew.py"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundml_amd import Stft
from soundml_amd._lib import lib, check
vp = ctypes.c_void_p
for fft, hop, n in ((1024, 256, 441000), (512, 128, 441000), (1024, 256, 442368), (1024, 255, 441000)):
    c = Stft.Config.create(fft_size=fft, hop=hop)
    frames = Stft.frames(c, n)
    x = torch.rand(256, n, device="cuda") * 2 - 1
    out = torch.empty(256, fft // 2 + 1, frames, device="cuda")
    run = lambda: check(lib.smx_stft_power_range_f32_dev(c._h, vp(x.data_ptr()), 256, n, n, 0, frames, 2.0, vp(out.data_ptr()), None))
    t0 = time.time()
    while time.time() - t0 < 1.0:
        for _ in range(50): run()
        torch.cuda.synchronize()
    res = {"1": [], "0": []}
    for rnd in range(6):
        for mode in ("1", "0"):
            os.environ["SMX_POWER_SKEW"] = mode
            for _ in range(10): run()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(60): run()
            b.record(); torch.cuda.synchronize()
            res[mode].append(a.elapsed_time(b) / 60)
    os.environ.pop("SMX_POWER_SKEW", None)
    med = lambda v: sorted(v)[len(v) // 2]
    print("fft %4d hop %3d n %6d (%d frames a clip): aligned lines %.4f ms, plain flush %.4f ms  (%+.1f %%)" % (fft, hop, n, frames, med(res["1"]), med(res["0"]), 100 * (med(res["1"]) / med(res["0"]) - 1)))
    del x, out

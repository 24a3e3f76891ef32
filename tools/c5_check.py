"""BASELINE C5 on one GPU: 4096 clips x 30 s x 48 kHz (23.6 GB in, 47.2 GB power out).  Checks sizes,
spot-checks frames of far-apart clips against the oracle, and times the pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import soundml_amd as S
from soundml_amd import Stft
from oracle import soundml_oracle as O
clips, n = int(os.environ.get("CLIPS", 4096)), int(os.environ.get("SAMPLES", 1440000))
c = Stft.Config.create(fft_size=2048, hop=512)
frames = Stft.frames(c, n)
g = torch.Generator(device="cuda"); g.manual_seed(42)
x = torch.empty(clips, n, device="cuda")
for i in range(0, clips, 256):
    x[i:i + 256] = torch.rand(min(256, clips - i), n, device="cuda", generator=g) * 2 - 1
torch.cuda.synchronize()
t0 = time.perf_counter(); p = Stft.power_spectrum(c, x); torch.cuda.synchronize(); first = time.perf_counter() - t0
import ctypes
from soundml_amd._lib import lib, check
def run():   # preallocated output, straight through the C ABI (what bench.py times)
    check(lib.smx_stft_power_range_f32_dev(c._h, ctypes.c_void_p(x.data_ptr()), clips, n, n, 0, frames, 2.0,
                                           ctypes.c_void_p(p.data_ptr()), None))
run(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ms = sorted(ts)[2]
assert tuple(p.shape) == (clips, 1025, frames)
o = O.stft_config(2048, hop=512)
worst = 0.0
for clip in (0, 1, clips // 2 - 1, clips - 1):
    xc = x[clip].cpu().numpy()
    for fa, fb in ((0, 3), (frames // 2, frames // 2 + 16), (frames - 3, frames)):
        want = np.abs(O.transform_range(o, xc, fa, fb, np.complex128)) ** 2
        got = p[clip, :, fa:fb].cpu().numpy().astype(np.float64)
        worst = max(worst, float(np.max(np.abs(got - want)) / np.max(want)))
assert worst < 1e-5, worst
print("C5-style run on 1 GPU: %d frames in %.2f ms (%.1f Mframes/s, %.0f GB/s algorithmic); first call %.1f ms; max rel err %.2e"
      % (clips * frames, ms, clips * frames / ms / 1e3, clips * frames * 6148 / ms / 1e6, first * 1e3, worst))

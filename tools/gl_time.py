"""Diagnostic: Griffin-Lim (32 iterations) on the C2-shaped magnitude batch; run under rocprofv3 --kernel-trace --stats
to see the split between synthesis, analysis and the elementwise steps."""
import ctypes, hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundml_amd import Stft
from soundml_amd._lib import lib, check
vp = ctypes.c_void_p
clips, n = int(os.environ.get("CLIPS", 256)), int(os.environ.get("N", 480000))
FFT, HOP = int(os.environ.get("FFT", 2048)), int(os.environ.get("HOP", 512))
c = Stft.Config.create(fft_size=FFT, hop=HOP)
torch.manual_seed(0)
frames = Stft.frames(c, n)
if os.environ.get("MAG") == "stft":   # magnitudes of a real transform (what tools/bench_extra.py feeds)
    x = torch.rand(clips, n, device="cuda") * 2 - 1
    mag = Stft.transform(c, x).abs().contiguous()
    del x
else:
    mag = torch.rand(clips, FFT // 2 + 1, frames, device="cuda")
out = torch.empty(clips, n, device="cuda")
if os.environ.get("INTERIOR") == "float64":
    check(lib.smx_set_interior(1))
def run():
    check(lib.smx_stft_griffin_lim_f32_dev(c._h, vp(mag.data_ptr()), clips, FFT // 2 + 1, frames, 32, 0.99, None, 1, n, vp(out.data_ptr()), None))
run(); torch.cuda.synchronize()
ts = []
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print("griffin_lim fft %d hop %d, 32 iterations, %d clips x %d samples: %.1f ms (%.3f ms per clip)" % (FFT, HOP, clips, n, sorted(ts)[1], sorted(ts)[1] / clips))
print("sha256 of the output:", hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16])

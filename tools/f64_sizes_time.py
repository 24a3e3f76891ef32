"""The float64 interior (the reference's own numerics) at several sizes: power / complex / invert, device resident float32 audio."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundml_amd as S
from soundml_amd import Stft
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
clips = int(os.environ.get("CLIPS", 64))
for fft, hop, n in ((2048, 512, 480000), (1024, 256, 480000), (400, 160, 480000), (1000, 250, 480000), (441, 220, 480000)):
    x = torch.rand(clips, n, device="cuda") * 2 - 1
    c = Stft.Config.create(fft_size=fft, hop=hop)
    row = []
    for interior in ("float32", "float64"):
        S.set_interior(interior)
        tp = t(lambda: Stft.power_spectrum(c, x))
        z = Stft.transform(c, x)
        tz = t(lambda: Stft.transform(c, x))
        ti = t(lambda: Stft.invert(c, z), reps=3) if Stft.nola(c) else float("nan")
        row.append("%s: power %.2f complex %.2f invert %.2f ms" % (interior, tp, tz, ti))
    S.set_interior("float32")
    print("fft %4d hop %3d (%d clips x %d): %s" % (fft, hop, clips, n, " | ".join(row)), flush=True)

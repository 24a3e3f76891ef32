"""Device time of power_spectrum / transform / fused mel at the other transform sizes (C ABI, outputs preallocated).
SOUNDML_AMD_LIB selects another build of the library for an A/B."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundml_amd import Stft
from soundml_amd._lib import lib, check
vp = ctypes.c_void_p
def t(fn, reps=15):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
line = []
SIZES = ((1024, 256, 441000), (512, 128, 441000), (400, 160, 160000), (4096, 1024, 480000), (2048, 512, 480000))
if len(sys.argv) > 1:   # fft:hop:n ...
    SIZES = tuple(tuple(int(v) for v in a.split(":")) for a in sys.argv[1:])
for fft, hop, n in SIZES:
    c = Stft.Config.create(fft_size=fft, hop=hop)
    frames = Stft.frames(c, n)
    x = torch.rand(256, n, device="cuda") * 2 - 1
    p = torch.empty(256, fft // 2 + 1, frames, device="cuda")
    z = torch.empty(256, fft // 2 + 1, frames, 2, device="cuda")
    tp = t(lambda: check(lib.smx_stft_power_range_f32_dev(c._h, vp(x.data_ptr()), 256, n, n, 0, frames, 2.0, vp(p.data_ptr()), None)))
    tz = t(lambda: check(lib.smx_stft_transform_range_f32_dev(c._h, vp(x.data_ptr()), 256, n, n, 0, frames, vp(z.data_ptr()), None)))
    line.append("fft %4d: power %.3f complex %.3f" % (fft, tp, tz))
print(os.path.basename(os.path.dirname(os.environ.get("SOUNDML_AMD_LIB", "lib/x"))), " | ".join(line))

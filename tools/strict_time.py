"""Diagnostic: C2-shaped power spectrogram with the float64 interior (the reference's own arithmetic)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundml_amd as S
from soundml_amd import Stft
from soundml_amd._lib import lib, check
vp = ctypes.c_void_p
clips, n = 256, 480000
c = Stft.Config.create(fft_size=2048, hop=512)
frames = Stft.frames(c, n)
x = torch.rand(clips, n, device="cuda") * 2 - 1
out = torch.empty(clips, 1025, frames, device="cuda")
for name in ("float32", "float64"):
    S.set_interior(name)
    def run():
        check(lib.smx_stft_power_range_f32_dev(c._h, vp(x.data_ptr()), clips, n, n, 0, frames, 2.0, vp(out.data_ptr()), None))
    for _ in range(2): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = sorted(ts)[2]
    print("interior %s: %.3f ms  %.1f Mframes/s" % (name, ms, clips * frames / ms / 1e3))

# float64 audio (float64 spectrogram out): always the float64 interior
x64 = x.double()
out64 = torch.empty(clips, 1025, frames, device="cuda", dtype=torch.float64)
def run64():
    check(lib.smx_stft_power_range_f64_dev(c._h, vp(x64.data_ptr()), clips, n, n, 0, frames, 2.0, vp(out64.data_ptr()), None))
for _ in range(2): run64()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run64(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ms = sorted(ts)[2]
print("float64 audio: %.3f ms  %.1f Mframes/s" % (ms, clips * frames / ms / 1e3))

# complex spectra with the float64 interior (float32 audio -> complex64)
outc = torch.empty(clips, 1025, frames, 2, device="cuda")
S.set_interior("float64")
def runc():
    check(lib.smx_stft_transform_range_f32_dev(c._h, vp(x.data_ptr()), clips, n, n, 0, frames, vp(outc.data_ptr()), None))
for _ in range(2): runc()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); runc(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
S.set_interior("float32")
ms = sorted(ts)[2]
print("transform, interior float64: %.3f ms  %.1f Mframes/s" % (ms, clips * frames / ms / 1e3))

"""Stft.invert of a C2-shaped spectrogram under the library named by SOUNDML_AMD_LIB: median time and a checksum of the output
(two builds whose kernels must agree bit for bit print the same checksum):  SOUNDML_AMD_LIB=... python tools/invert_check.py"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soundml_amd import Stft
torch.manual_seed(0)
clips, n = int(os.environ.get("CLIPS", "256")), int(os.environ.get("N", "480000"))
x = torch.rand(clips, n, device="cuda") * 2 - 1
fft = int(os.environ.get("FFT", "2048"))
c = Stft.Config.create(fft_size=fft, hop=int(os.environ.get("HOP", str(fft // 4))))
z = Stft.transform(c, x)
for _ in range(3): y = Stft.invert(c, z, length=n)
torch.cuda.synchronize()
ts = []
for _ in range(20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): y = Stft.invert(c, z, length=n)
    b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 3)
ts.sort()
h = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16]
print("%s: invert min %.4f median %.4f ms  max |y - x| %.3g  sha256 %s" % (os.environ.get("SOUNDML_AMD_LIB", "default"), ts[0], ts[len(ts) // 2],
                                                                           float((y - x).abs().max()), h))

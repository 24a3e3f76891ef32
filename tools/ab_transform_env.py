"""Interleaved timing of Stft.transform (C2: 256 x 480000, fft 2048 / hop 512, complex64 out) under several environments:
  python tools/ab_transform_env.py "" "SMX_COMPLEX_SKEW=0"   (AB_N: other clip lengths, AB_FFT: 1024 / 512)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soundml_amd import Stft
envs = sys.argv[1:] or [""]
n = int(os.environ.get("AB_N", "480000"))
x = torch.rand(256, n, device="cuda") * 2 - 1
fft = int(os.environ.get("AB_FFT", "2048"))
c = Stft.Config.create(fft_size=fft, hop=fft // 4)
def setenv(e, on):
    for kv in filter(None, e.split(",")):
        k, v = kv.split("=")
        if on: os.environ[k] = v
        else: os.environ.pop(k, None)
res = {}
for e in envs:
    setenv(e, True)
    for _ in range(3): z = Stft.transform(c, x)
    torch.cuda.synchronize(); res[e] = z.clone(); setenv(e, False)
frames = res[envs[0]].shape[-1]
ts = {e: [] for e in envs}
for rnd in range(20):
    for e in envs:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        setenv(e, True); a.record()
        for _ in range(3): z = Stft.transform(c, x)
        b.record(); torch.cuda.synchronize(); setenv(e, False)
        ts[e].append(a.elapsed_time(b) / 3)
for e in envs:
    v = sorted(ts[e])
    d = float((res[e] - res[envs[0]]).abs().max() / res[envs[0]].abs().max())
    print("%-24s min %.4f  median %.4f ms  (%.1f Mframes/s, allocation of the output included)  max |diff| vs first / peak %.2e" % (
        e or "(default)", v[0], v[len(v) // 2], 256 * frames / v[len(v) // 2] / 1e3, d))

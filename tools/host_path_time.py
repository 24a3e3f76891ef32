"""The drop-in's host entry points (numpy in, numpy out: upload, kernels, download inside the C ABI) at C2, beside the
device-resident call: the PCIe-inclusive rate DESIGN section 5 quotes.   python tools/host_path_time.py [clips]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from soundml_amd import Stft
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = np.random.default_rng(0).uniform(-1, 1, size=(clips, 480000)).astype(np.float32)
c = Stft.Config.create(fft_size=2048, hop=512)
frames = clips * Stft.frames(c, 480000)
def t(fn, reps=5):
    keep = fn(); ts = []
    for _ in range(reps):
        del keep      # releasing a GB of touched pages costs more than the call (53 ms for the C2 spectrogram): not timed
        a = time.perf_counter(); keep = fn(); ts.append(time.perf_counter() - a)
    return sorted(ts)[len(ts) // 2]
for name, fn in (("power_spectrum", lambda: Stft.power_spectrum(c, x)), ("transform", lambda: Stft.transform(c, x))):
    s = t(fn)
    print("host %-15s %8.1f ms  %6.1f Mframes/s" % (name, s * 1e3, frames / s / 1e6))
xd = torch.from_numpy(x).cuda()
def dev():
    Stft.power_spectrum(c, xd); torch.cuda.synchronize()
s = t(dev)
print("device power_spectrum %6.2f ms  %6.1f Mframes/s (allocating the output each call)" % (s * 1e3, frames / s / 1e6))
# where the host call's time goes on the Python side
import ctypes as C
from soundml_amd import _lib
a = time.perf_counter(); out = np.zeros((clips, 1025, Stft.frames(c, 480000)), dtype=np.float32); b = time.perf_counter()
r = Stft.power_spectrum(c, x); d = time.perf_counter(); del r; e = time.perf_counter(); del out; f = time.perf_counter()
print("np.zeros %.2f ms, call %.2f ms, dropping the result %.2f ms, dropping an untouched array %.2f ms" % ((b - a) * 1e3, (d - b) * 1e3, (e - d) * 1e3, (f - e) * 1e3))

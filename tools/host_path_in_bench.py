"""The host-pointer power spectrogram at C2 in bench.py's situation (torch imported, a device-resident batch and its spectrogram alive,
the host batch made by x.cpu().numpy()), under several environments, interleaved.  python tools/host_path_in_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from soundml_amd import Stft
c = Stft.Config.create(fft_size=2048, hop=512)
x = (torch.rand(256, 480000, device="cuda") * 2 - 1).float()
out = Stft.power_spectrum(c, x)
xh = x.cpu().numpy()
print("torch threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads())
envs = [{}, {"SMX_HOST_PIPELINE": "0"}]
res = {i: [] for i in range(len(envs))}
for rnd in range(8):
    for i, e in enumerate(envs):
        for k in ("SMX_HOST_PIPELINE", "SMX_COPY_HUGEPAGE"):
            os.environ.pop(k, None)
        os.environ.update(e)
        a = time.perf_counter(); y = Stft.power_spectrum(c, xh); b = time.perf_counter()
        res[i].append((b - a) * 1e3)
        del y
for i, e in enumerate(envs):
    print("%-60s %s" % (e or "(default)", " ".join("%.0f" % t for t in res[i])))
for k in ("SMX_HOST_PIPELINE", "SMX_COPY_HUGEPAGE"):
    os.environ.pop(k, None)
def series(name, arr, n=6):
    ts = []
    for _ in range(n):
        a = time.perf_counter(); y = Stft.power_spectrum(c, arr); b = time.perf_counter(); ts.append((b - a) * 1e3); del y
    print("%-60s %s" % (name, " ".join("%.0f" % t for t in ts)))
series("x.cpu().numpy() again", xh)
xh2 = np.array(xh)
series("np.array copy of it (first touched by this thread)", xh2)
xr = np.random.default_rng(0).uniform(-1, 1, size=xh.shape).astype(np.float32)
series("numpy-generated batch", xr)
del out
torch.cuda.empty_cache()
series("numpy batch, device spectrogram freed", xr)
del x
torch.cuda.empty_cache()
series("numpy batch, every device tensor freed", xr)

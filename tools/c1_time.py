import sys, os, torch
sys.path.insert(0, os.getcwd())
from soundml_amd import Stft
x = torch.rand(1, 441000, device="cuda") * 2 - 1
for fft, hop in ((1024, 256), (2048, 512), (512, 128)):
    c = Stft.Config.create(fft_size=fft, hop=hop)
    for _ in range(5): Stft.power_spectrum(c, x)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): Stft.power_spectrum(c, x)
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 10)
    ts.sort()
    print("fft %d: 1 x 441000 device call %.4f ms (median of 30 x 10 back-to-back calls, allocation included)" % (fft, ts[len(ts)//2]))

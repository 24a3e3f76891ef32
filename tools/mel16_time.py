import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from soundml_amd import Stft, Mel
from soundml_amd._lib import lib, check
vp = ctypes.c_void_p
def t(fn, reps=9):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
clips, n = 256, 480000
x = torch.rand(clips, n, device="cuda") * 2 - 1
for fft, hop, sr in ((400, 160, 16000), (512, 160, 16000), (1024, 256, 22050)):
    c = Stft.Config.create(fft_size=fft, hop=hop)
    frames = Stft.frames(c, n)
    p = torch.empty(clips, fft // 2 + 1, frames, device="cuda")
    tp = t(lambda: check(lib.smx_stft_power_range_f32_dev(c._h, vp(x.data_ptr()), clips, n, n, 0, frames, 2.0, vp(p.data_ptr()), None)))
    row = ["fft %d hop %d: power %.3f" % (fft, hop, tp)]
    for nm in (16, 40, 80, 128):
        mc = Mel.Config.create(n_mels=nm, sample_rate=sr, fft_size=fft)
        m = torch.empty(clips, nm, frames, device="cuda")
        row.append("mel%d %.3f" % (nm, t(lambda: check(lib.smx_mel_spectrogram_f32_dev(c._h, mc._h, vp(x.data_ptr()), clips, n, n, 2.0, vp(m.data_ptr()), None)))))
    print(" | ".join(row), flush=True)

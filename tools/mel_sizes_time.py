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
for hop in (512, 480):
    c = Stft.Config.create(fft_size=2048, hop=hop)
    frames = Stft.frames(c, n)
    for sr, nm in ((48000, 128), (48000, 80), (48000, 64), (48000, 40), (48000, 256), (22050, 80), (16000, 80)):
        mc = Mel.Config.create(n_mels=nm, sample_rate=sr, fft_size=2048)
        m = torch.empty(clips, nm, frames, device="cuda")
        tm = t(lambda: check(lib.smx_mel_spectrogram_f32_dev(c._h, mc._h, vp(x.data_ptr()), clips, n, n, 2.0, vp(m.data_ptr()), None)))
        print("hop %d sr %d n_mels %3d: %.3f ms" % (hop, sr, nm, tm), flush=True)

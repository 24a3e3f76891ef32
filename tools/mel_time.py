import sys, os, torch
sys.path.insert(0, os.getcwd())
import soundml_amd as S
from soundml_amd import Stft, Mel
x = torch.rand(256, 480000, device="cuda") * 2 - 1
sc = Stft.Config.create(fft_size=2048, hop=512); mc = Mel.Config.create(n_mels=128, sample_rate=48000, fft_size=2048)
def t(fn, reps=15):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
print("NOMFMA" if os.environ.get("SMX_MEL_NOMFMA") else "full  ", "mel_spectrogram %.3f ms   power_spectrum %.3f ms" % (t(lambda: S.mel_spectrogram(sc, mc, x)), t(lambda: Stft.power_spectrum(sc, x))))

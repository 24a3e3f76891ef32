"""Small-call latency: one short clip through the host entry points (numpy in, numpy out) and the device ones."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundml_amd as S
from soundml_amd import Stft, Mel
def t(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for n in (16000, 44100, 480000):
    x = np.random.default_rng(0).uniform(-1, 1, n).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    c = Stft.Config.create(fft_size=2048, hop=512)
    mc = Mel.Config.create(n_mels=128, sample_rate=22050, fft_size=2048)
    c4 = Stft.Config.create(fft_size=400, hop=160)
    m4 = Mel.Config.create(n_mels=80, sample_rate=16000, fft_size=400)
    print("n %6d | host: power %.3f ms  mel %.3f  transform %.3f  whisper-mel %.3f | device: power %.3f  mel %.3f  whisper-mel %.3f" % (
        n, t(lambda: Stft.power_spectrum(c, x)), t(lambda: S.mel_spectrogram(c, mc, x)), t(lambda: Stft.transform(c, x)), t(lambda: S.mel_spectrogram(c4, m4, x)),
        t(lambda: Stft.power_spectrum(c, xd)), t(lambda: S.mel_spectrogram(c, mc, xd)), t(lambda: S.mel_spectrogram(c4, m4, xd))), flush=True)

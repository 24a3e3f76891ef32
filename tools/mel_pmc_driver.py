"""Driver for tools/pmc_mel.sh: a few launches of the two mel kernels at C3 (fused audio -> 128 mels, and the unfused
Mel.apply on a resident power spectrogram), nothing else on the device."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import soundml_amd as S
from soundml_amd import Stft, Mel
x = torch.rand(256, 480000, device="cuda") * 2 - 1
sc = Stft.Config.create(fft_size=2048, hop=512)
mc = Mel.Config.create(n_mels=128, sample_rate=48000, fft_size=2048)
p = Stft.power_spectrum(sc, x)
for _ in range(4):
    S.mel_spectrogram(sc, mc, x)
    Mel.apply(mc, p)
torch.cuda.synchronize()

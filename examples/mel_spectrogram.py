#!/usr/bin/env python3
"""The reference's first example (examples/01-mel-spectrogram/main.ml) on the MI355X path: a 10 s 440 Hz tone at
44.1 kHz, STFT with fft 1024 / hop 256, 128 mel bands, decibels under the loudest cell -- and the same front end on
a device-resident batch.  Needs a HIP device (there is no CPU fallback)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundml_amd import Mel, Stft, mel_spectrogram, mfcc, power_to_db, spectral_centroid   # noqa: E402

sr, n = 44100, 441000
x = np.sin(2 * np.pi * 440.0 * np.arange(n) / sr).astype(np.float32)
stft = Stft.Config.create(fft_size=1024, hop=256)              # Hann, centered, reflect: librosa's defaults
mel = Mel.Config.create(n_mels=128, sample_rate=sr, fft_size=1024)
m = mel_spectrogram(stft, mel, x)                              # [128; 1723] float32, one fused launch
db = power_to_db(m, top_db=80.0)                               # Convert.power_to_db: 80 dB under the loudest cell
print("mel spectrogram", m.shape, m.dtype, "loudest band", int(np.argmax(m[:, 800])), "range %.1f dB" % (db.max() - db.min()))
print("spectral centroid of the tone: %.1f Hz" % float(spectral_centroid(np.sqrt(Stft.power_spectrum(stft, x)), sample_rate=sr)[0, 800]))
print("mfcc", mfcc(stft, mel, x, n_mfcc=13).shape)

try:
    import torch
    if torch.cuda.is_available():
        batch = torch.rand(64, n, device="cuda") * 2 - 1       # device-resident in, device-resident out
        out = mel_spectrogram(stft, mel, batch)
        torch.cuda.synchronize()
        print("batch on", out.device, tuple(out.shape))
except ImportError:
    pass

"""Mel.apply alone on the resident C2 power spectrogram (and two other shapes).  SMX_MEL_APPLY_BY_TILE=1 selects the
64-frame kernel.   python tools/mel_apply_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soundml_amd import Mel
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
for clips, fft, frames, n_mels, sr in ((256, 2048, 938, 128, 48000), (256, 2048, 938, 80, 48000), (256, 1024, 1876, 80, 16000), (64, 512, 3000, 40, 16000)):
    p = torch.rand(clips, fft // 2 + 1, frames, device="cuda")
    mc = Mel.Config.create(n_mels=n_mels, sample_rate=sr, fft_size=fft)
    ms = t(lambda: Mel.apply(mc, p))
    gb = clips * frames * 4 * (fft // 2 + 1 + n_mels) / 1e9
    print("%s Mel.apply %4d clips fft %4d frames %4d mels %3d: %.3f ms  %.2f TB/s" % (
        "tile " if os.environ.get("SMX_MEL_APPLY_BY_TILE") else "block", clips, fft, frames, n_mels, ms, gb / ms))

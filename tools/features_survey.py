"""Callers of the spectrogram at several shapes (device resident): Mel.apply, mfcc, power_to_db, spectral features, Chroma.apply."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundml_amd as S
from soundml_amd import Stft, Mel
def t(fn, reps=7):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
for clips, fft, frames, sr, nm in ((256, 2048, 938, 48000, 128), (256, 2048, 938, 48000, 80), (256, 1024, 1876, 22050, 80), (256, 512, 3751, 16000, 40),
                                   (256, 400, 3001, 16000, 80), (4096, 400, 101, 16000, 80), (16, 4096, 938, 48000, 128)):
    bins = fft // 2 + 1
    s = torch.rand(clips, bins, frames, device="cuda")
    gb = s.numel() * 4 / 1e9
    mc = Mel.Config.create(n_mels=nm, sample_rate=sr, fft_size=fft)
    row = ["%4d x %4d x %4d (%.2f GB)" % (clips, bins, frames, gb)]
    row.append("Mel.apply%-3d %.3f ms %4.0f GB/s" % (nm, (tm := t(lambda: Mel.apply(mc, s))), gb / tm * 1e3))
    row.append("to_db %.3f" % t(lambda: S.power_to_db(s)))
    row.append("centroid %.3f" % t(lambda: S.spectral_centroid(s, sample_rate=sr)))
    row.append("bandwidth %.3f" % t(lambda: S.spectral_bandwidth(s, sample_rate=sr)))
    row.append("rolloff %.3f" % t(lambda: S.spectral_rolloff(s, sample_rate=sr)))
    row.append("flatness %.3f" % t(lambda: S.spectral_flatness(s)))
    try:
        cc = S.Chroma.Config.create(sr, fft)
        row.append("chroma %.3f" % t(lambda: S.Chroma.apply(cc, s)))
    except Exception as e:
        row.append("chroma n/a")
    print(" | ".join(row), flush=True)
    del s

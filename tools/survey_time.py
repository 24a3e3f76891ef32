"""A survey of geometries (device resident, C ABI): power / complex / mel / invert times, to look for outliers.
    python tools/survey_time.py            fixed list below;   rows: clips x samples, fft / hop -> Mframes/s and GB/s algorithmic"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundml_amd as S
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
CASES = [  # clips, n, fft, hop, sr, n_mels
    (256, 480000, 2048, 512, 48000, 128), (256, 480000, 2048, 256, 48000, 128), (256, 480000, 2048, 1024, 48000, 128),
    (256, 480000, 2048, 480, 48000, 80), (4096, 48000, 2048, 512, 48000, 128), (16384, 16000, 2048, 512, 48000, 128),
    (256, 480000, 1024, 256, 22050, 80), (256, 480000, 512, 128, 16000, 40), (256, 480000, 256, 64, 16000, 40),
    (256, 480000, 4096, 1024, 48000, 128), (256, 480000, 400, 160, 16000, 80), (8192, 16000, 400, 160, 16000, 80),
    (256, 480000, 960, 480, 48000, 64), (256, 480000, 441, 220, 22050, 40), (1, 4800000, 2048, 512, 48000, 128),
]
for clips, n, fft, hop, sr, n_mels in CASES:
    c = Stft.Config.create(fft_size=fft, hop=hop)
    frames = Stft.frames(c, n)
    bins = fft // 2 + 1
    x = torch.rand(clips, n, device="cuda") * 2 - 1
    p = torch.empty(clips, bins, frames, device="cuda")
    z = torch.empty(clips, bins, frames, 2, device="cuda")
    tp = t(lambda: check(lib.smx_stft_power_range_f32_dev(c._h, vp(x.data_ptr()), clips, n, n, 0, frames, 2.0, vp(p.data_ptr()), None)))
    tz = t(lambda: check(lib.smx_stft_transform_range_f32_dev(c._h, vp(x.data_ptr()), clips, n, n, 0, frames, vp(z.data_ptr()), None)))
    try:
        mc = Mel.Config.create(n_mels=n_mels, sample_rate=sr, fft_size=fft)
        m = torch.empty(clips, n_mels, frames, device="cuda")
        tm = t(lambda: check(lib.smx_mel_spectrogram_f32_dev(c._h, mc._h, vp(x.data_ptr()), clips, n, n, 2.0, vp(m.data_ptr()), None)))
    except Exception:
        tm = float("nan")
    try:
        zi = torch.view_as_complex(z)
        ti = t(lambda: Stft.invert(c, zi), reps=5) if Stft.nola(c) else float("nan")
    except Exception:
        ti = float("nan")
    fr = clips * frames
    gb = lambda ms, out_b: fr * (hop * 4 + out_b) / ms / 1e6
    print("%6d x %7d  fft %4d hop %4d  frames %8d | power %.3f ms %6.0f Mf/s %5.0f GB/s | complex %.3f ms %5.0f GB/s | mel%-3d %.3f ms | invert %.3f ms"
          % (clips, n, fft, hop, fr, tp, fr / tp / 1e3, gb(tp, bins * 4), tz, gb(tz, bins * 8), n_mels, tm, ti), flush=True)
    del x, p, z

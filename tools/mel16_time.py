"""Diagnostic: Soundml.mel_spectrogram at fft 1024 / hop 256 (80 mels, 22.05 kHz) and fft 512 / hop 128, fused kernel
against the power + Mel.apply composition (SMX_MEL16_OFF=1)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundml_amd import Stft, Mel
from soundml_amd._lib import lib, check
vp = ctypes.c_void_p
for fft, hop, n_mels, sr, clips, n in ((1024, 256, 80, 22050, 256, 441000), (512, 128, 80, 16000, 256, 441000), (400, 160, 80, 16000, 256, 480000)):
    c = Stft.Config.create(fft_size=fft, hop=hop)
    m = Mel.Config.create(n_mels=n_mels, sample_rate=sr, fft_size=fft)
    frames = Stft.frames(c, n)
    x = torch.rand(clips, n, device="cuda") * 2 - 1
    out = torch.empty(clips, n_mels, frames, device="cuda")
    for mode in ("0", "1", "0", "notail"):
        os.environ["SMX_MEL16_OFF"] = "0" if mode == "notail" else mode
        os.environ["SMX_MEL16_NOTAIL"] = "1" if mode == "notail" else "0"
        def run():
            check(lib.smx_mel_spectrogram_f32_dev(c._h, m._h, vp(x.data_ptr()), clips, n, n, 2.0, vp(out.data_ptr()), None))
        for _ in range(2): run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        ms = sorted(ts)[3]
        print("fft %4d hop %3d %3d mels, %s: %d clips x %d frames in %.3f ms  (%.1f Mframes/s)"
              % (fft, hop, n_mels, "composition" if mode == "1" else ("no tail    " if mode == "notail" else "fused      "), clips, frames, ms, clips * frames / ms / 1e3))
    os.environ["SMX_MEL16_OFF"] = "0"
    del x, out

"""Interleaved timing of the fused mel spectrogram (C3: 256 x 480000, 128 mels) under several environments:
  python tools/ab_mel_env.py ""                  (AB_MELS, AB_SR: other filterbanks; AB_FFT, AB_N: other sizes; switches as the build reads them)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes, torch
from soundml_amd import Mel, Stft
from soundml_amd._lib import check, lib
vp = ctypes.c_void_p
envs = sys.argv[1:] or [""]
fft = int(os.environ.get("AB_FFT", "2048"))
n = int(os.environ.get("AB_N", "480000"))
x = torch.rand(256, n, device="cuda") * 2 - 1
sc = Stft.Config.create(fft_size=fft, hop=fft // 4)
n_mels, sr = int(os.environ.get("AB_MELS", "128")), int(os.environ.get("AB_SR", "48000"))
mc = Mel.Config.create(n_mels=n_mels, sample_rate=sr, fft_size=fft)
frames = Stft.frames(sc, n)
out = torch.empty(256, n_mels, frames, device="cuda")
def run():
    check(lib.smx_mel_spectrogram_f32_dev(sc._h, mc._h, vp(x.data_ptr()), 256, n, n, 2.0, vp(out.data_ptr()), None))
def setenv(e, on):
    for kv in filter(None, e.split(",")):
        k, v = kv.split("=")
        if on: os.environ[k] = v
        else: os.environ.pop(k, None)
res = {}
for e in envs:
    setenv(e, True)
    for _ in range(3): run()
    torch.cuda.synchronize(); res[e] = out.clone()
    setenv(e, False)
ts = {e: [] for e in envs}
for rnd in range(30):
    for e in envs:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        setenv(e, True); a.record()
        for _ in range(4): run()
        b.record(); torch.cuda.synchronize(); setenv(e, False)
        ts[e].append(a.elapsed_time(b) / 4)
for e in envs:
    v = sorted(ts[e])
    d = float((res[e] - res[envs[0]]).abs().max() / res[envs[0]].abs().max())
    print("%-24s min %.4f  median %.4f ms  (%.1f Mframes/s)  max |diff| vs first / peak %.2e" % (e or "(default)", v[0], v[len(v) // 2], 256 * frames / v[len(v) // 2] / 1e3, d))

"""fft 400 / hop 160: four power launches and four fused mel80 launches, for a rocprofv3 --kernel-trace --stats pass"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundml_amd import Stft, Mel
from soundml_amd._lib import lib, check
vp = ctypes.c_void_p
clips, n = 256, 480000
x = torch.rand(clips, n, device="cuda") * 2 - 1
c = Stft.Config.create(fft_size=400, hop=160)
frames = Stft.frames(c, n)
p = torch.empty(clips, 201, frames, device="cuda")
mc = Mel.Config.create(n_mels=80, sample_rate=16000, fft_size=400)
m = torch.empty(clips, 80, frames, device="cuda")
for _ in range(4):
    check(lib.smx_stft_power_range_f32_dev(c._h, vp(x.data_ptr()), clips, n, n, 0, frames, 2.0, vp(p.data_ptr()), None))
torch.cuda.synchronize()
for _ in range(4):
    check(lib.smx_mel_spectrogram_f32_dev(c._h, mc._h, vp(x.data_ptr()), clips, n, n, 2.0, vp(m.data_ptr()), None))
torch.cuda.synchronize()

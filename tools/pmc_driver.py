"""Driver for tools/profile_round.sh: a few launches of every hot kernel of the BASELINE configurations, nothing else on
the device (C2 power spectrogram, C2 with the ring-form kernel, C2 complex, C3 fused mel, Mel.apply, C4 FIR)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import soundml_amd as S
from soundml_amd import Fir, Mel, Stft
from soundml_amd._lib import check, lib
vp = ctypes.c_void_p
x = torch.rand(256, 480000, device="cuda") * 2 - 1
sc = Stft.Config.create(fft_size=2048, hop=512)
mc = Mel.Config.create(n_mels=128, sample_rate=48000, fft_size=2048)
frames = Stft.frames(sc, 480000)
out = torch.empty(256, 1025, frames, device="cuda")
mout = torch.empty(256, 128, frames, device="cuda")
h = Fir.design_lowpass(8192, 0.25, 100.0)
plan = Fir.Plan.create(h)
xs = torch.rand(8, 2880000, device="cuda") * 2 - 1
ys = torch.empty_like(xs)
reps = int(os.environ.get("REPS", "4"))
for _ in range(reps):
    os.environ.pop("SMX_POWER_RING", None)
    check(lib.smx_stft_power_range_f32_dev(sc._h, vp(x.data_ptr()), 256, 480000, 480000, 0, frames, 2.0, vp(out.data_ptr()), None))
    os.environ["SMX_POWER_V1"] = "1"      # the 64-lane kernel, for the record beside the 32-lane one
    check(lib.smx_stft_power_range_f32_dev(sc._h, vp(x.data_ptr()), 256, 480000, 480000, 0, frames, 2.0, vp(out.data_ptr()), None))
    os.environ.pop("SMX_POWER_V1", None)
    check(lib.smx_mel_spectrogram_f32_dev(sc._h, mc._h, vp(x.data_ptr()), 256, 480000, 480000, 2.0, vp(mout.data_ptr()), None))
    check(lib.smx_mel_apply_f32_dev(mc._h, vp(out.data_ptr()), 256, 1025, frames, vp(mout.data_ptr()), None))
    check(lib.smx_fir_apply_f32_dev(plan._h, vp(xs.data_ptr()), 8, 2880000, 2880000, vp(ys.data_ptr()), 2880000, None))
torch.cuda.synchronize()

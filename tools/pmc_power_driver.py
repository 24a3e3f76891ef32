"""Driver for tools/pmc_power.sh: a few launches of the two fft-2048 power kernels on C2 (the 32-lane pipeline and, under
SMX_POWER_V1=1, the 64-lane one), nothing else on the device."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soundml_amd import Stft
from soundml_amd._lib import check, lib
vp = ctypes.c_void_p
x = torch.rand(256, 480000, device="cuda") * 2 - 1
sc = Stft.Config.create(fft_size=2048, hop=512)
frames = Stft.frames(sc, 480000)
out = torch.empty(256, 1025, frames, device="cuda")
for _ in range(int(os.environ.get("REPS", "4"))):
    os.environ.pop("SMX_POWER_V1", None)
    check(lib.smx_stft_power_range_f32_dev(sc._h, vp(x.data_ptr()), 256, 480000, 480000, 0, frames, 2.0, vp(out.data_ptr()), None))
    os.environ["SMX_POWER_V1"] = "1"
    check(lib.smx_stft_power_range_f32_dev(sc._h, vp(x.data_ptr()), 256, 480000, 480000, 0, frames, 2.0, vp(out.data_ptr()), None))
torch.cuda.synchronize()

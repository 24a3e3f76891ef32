"""four launches of the complex kernel at each of two clip lengths, for a rocprofv3 --pmc pass (tools/README.md)"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundml_amd import Stft
from soundml_amd._lib import check, lib
vp = ctypes.c_void_p
c = Stft.Config.create(fft_size=2048, hop=512)
for n in (480000, 482816):
    frames = Stft.frames(c, n)
    x = torch.empty(256, n, device="cuda").uniform_(-1, 1)
    oc = torch.empty(256, 1025, frames, 2, device="cuda")
    for _ in range(4):
        check(lib.smx_stft_transform_range_f32_dev(c._h, vp(x.data_ptr()), 256, n, n, 0, frames, vp(oc.data_ptr()), None))
    torch.cuda.synchronize()

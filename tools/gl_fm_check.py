"""Griffin-Lim's internal frame-major spectra (round 5, late): stft2048_complex_fm_kernel against Stft.transform on the same audio
(bit for bit, every clip / frame / bin, with centred, left and right alignment and an odd hop) and its time at C2."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundml_amd import Stft
from soundml_amd._lib import lib, check
vp = ctypes.c_void_p
PITCH = 2 * 1032
def fm(c, x):
    lead, n = x.shape
    frames = Stft.frames(c, n)
    rows = (frames + 15) // 16 * 16
    out = torch.full((lead, rows, PITCH // 2, 2), float("nan"), device="cuda", dtype=torch.float32)
    check(lib.smx_debug_stft_transform_frame_major_f32_dev(c._h, vp(x.data_ptr()), lead, n, vp(out.data_ptr()), PITCH, rows, None))
    return out, frames
for kw, lead, n in ((dict(hop=512), 3, 60000), (dict(hop=512, alignment="left", pad="edge"), 2, 20011), (dict(hop=512, alignment="right", pad=("constant", 0.25)), 5, 33000),
                    (dict(hop=511), 2, 30000), (dict(hop=512), 300, 9000), (dict(hop=512), 1, 2048)):
    c = Stft.Config.create(fft_size=2048, **kw)
    x = (torch.rand(lead, n, device="cuda") * 2 - 1).float()
    got, frames = fm(c, x)
    want = torch.view_as_real(Stft.transform(c, x))            # [lead, 1025, frames, 2]
    g = got[:, :frames, :1025, :].permute(0, 2, 1, 3)
    same = torch.equal(g.contiguous(), want.contiguous())
    print(kw, lead, n, "frames", frames, "bit-equal" if same else "DIFFERENT max %.3g" % float((g - want).abs().max()))
    assert same
c = Stft.Config.create(fft_size=2048, hop=512)
x = (torch.rand(256, 480000, device="cuda") * 2 - 1).float()
frames = Stft.frames(c, 480000)
rows = (frames + 15) // 16 * 16
out = torch.empty(256, rows, PITCH, device="cuda")
z = torch.empty(256, 1025, frames, 2, device="cuda")
def t(fn, reps=40):
    for _ in range(200): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
t_fm = t(lambda: check(lib.smx_debug_stft_transform_frame_major_f32_dev(c._h, vp(x.data_ptr()), 256, 480000, vp(out.data_ptr()), PITCH, rows, None)))
t_ref = t(lambda: check(lib.smx_stft_transform_range_f32_dev(c._h, vp(x.data_ptr()), 256, 480000, 480000, 0, frames, vp(z.data_ptr()), None)))
print("C2: frame-major %.4f ms, Stft.transform %.4f ms" % (t_fm, t_ref))

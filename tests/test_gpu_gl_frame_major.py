"""Griffin-Lim's rebuilt spectra frame-major (round 5, late): at fft 2048 / hop 512 the loop of Stft.griffin_lim (stft.ml:961-1017)
keeps c_k as [clip][frame][bin] -- the analysis stores a frame's bins straight from its registers (stft2048_complex_fm_kernel), the
synthesis stages whole rows (istft2048_pipe_kernel<.., true>).  c_k never leaves the library, and every value is computed by the
arithmetic of the reference-layout kernels, so: (1) the frame-major analysis equals Stft.transform bit for bit; (2) the loop's output
equals the reference-layout loop's (SMX_GL_FRAME_MAJOR=0) bit for bit, for every shape of call."""
import ctypes
import os

import numpy as np
import pytest

from soundml_amd import Stft
from soundml_amd._lib import lib, check

pytestmark = pytest.mark.gpu
PITCH = 2 * 1032


@pytest.mark.parametrize("kw,lead,n", [
    (dict(hop=512), 3, 60000),
    (dict(hop=512, alignment="left", pad="edge"), 2, 20011),
    (dict(hop=512, alignment="right", pad=("constant", 0.25)), 5, 33000),
    (dict(hop=511), 2, 30000),                      # odd hop: the unaligned loads
    (dict(hop=512), 300, 9000),                     # more clips than workgroups, every tile a border tile
    (dict(hop=512), 1, 2048),
    (dict(hop=512, win_length=1200), 2, 50000),
])
def test_frame_major_analysis_equals_transform(kw, lead, n):
    import torch
    vp = ctypes.c_void_p
    torch.manual_seed(n)
    c = Stft.Config.create(fft_size=2048, **kw)
    x = (torch.rand(lead, n, device="cuda") * 2 - 1).float()
    frames = Stft.frames(c, n)
    rows = (frames + 15) // 16 * 16
    out = torch.full((lead, rows, PITCH // 2, 2), float("nan"), device="cuda", dtype=torch.float32)
    check(lib.smx_debug_stft_transform_frame_major_f32_dev(c._h, vp(x.data_ptr()), lead, n, vp(out.data_ptr()), PITCH, rows, None))
    want = torch.view_as_real(Stft.transform(c, x))            # [lead, 1025, frames, 2]
    got = out[:, :frames, :1025, :].permute(0, 2, 1, 3).contiguous()
    assert torch.equal(got, want.contiguous())
    assert torch.isnan(out[:, :frames, 1025:, :]).all()        # the row's padding is not written


def test_frame_major_analysis_refuses_what_it_cannot_take():
    import torch
    vp = ctypes.c_void_p
    x = torch.zeros(2, 30000, device="cuda")
    out = torch.empty(2, 64, PITCH, device="cuda")
    c1 = Stft.Config.create(fft_size=1024, hop=256)
    assert lib.smx_debug_stft_transform_frame_major_f32_dev(c1._h, vp(x.data_ptr()), 2, 30000, vp(out.data_ptr()), PITCH, 64, None) != 0
    c2 = Stft.Config.create(fft_size=2048, hop=512)
    frames = Stft.frames(c2, 30000)
    assert lib.smx_debug_stft_transform_frame_major_f32_dev(c2._h, vp(x.data_ptr()), 2, 30000, vp(out.data_ptr()), PITCH, frames, None) != 0   # rows not a multiple of 16
    assert lib.smx_debug_stft_transform_frame_major_f32_dev(c2._h, vp(x.data_ptr()), 2, 30000, vp(out.data_ptr()), 2048, 64, None) != 0     # rows too short for 1025 bins


@pytest.mark.parametrize("n_iter,momentum,length,with_phase,kw,lead,n", [
    (32, 0.99, None, False, dict(), 3, 40000),
    (1, 0.99, None, False, dict(), 2, 20000),
    (2, 0.99, 17000, False, dict(), 2, 20000),
    (3, 0.0, None, True, dict(), 2, 20000),
    (5, 0.5, 23456, True, dict(alignment="left", pad="edge"), 2, 25000),
    (4, 0.99, None, False, dict(alignment="right"), 1, 30000),
    (6, 0.99, 9000, True, dict(), 40, 9000),
    (3, 0.99, None, False, dict(win_length=1024), 2, 20000),
])
def test_frame_major_loop_equals_the_reference_layout_loop(n_iter, momentum, length, with_phase, kw, lead, n):
    import torch
    torch.manual_seed(n_iter + n)
    c = Stft.Config.create(fft_size=2048, hop=512, **kw)
    x = (torch.rand(lead, n, device="cuda") * 2 - 1).float()
    mag = Stft.transform(c, x).abs().contiguous()
    init = (torch.rand_like(mag) * 6.28 - 3.14) if with_phase else None
    outs = []
    for fm in ("1", "0"):
        os.environ["SMX_GL_FRAME_MAJOR"] = fm
        try:
            outs.append(Stft.griffin_lim(c, mag, n_iter=n_iter, momentum=momentum, init=init, length=length))
        finally:
            os.environ.pop("SMX_GL_FRAME_MAJOR", None)
    assert outs[0].shape == outs[1].shape and torch.equal(outs[0], outs[1])
    # ... and the host face (numpy in, numpy out) the same values
    host = Stft.griffin_lim(c, mag.cpu().numpy(), n_iter=n_iter, momentum=momentum, init=None if init is None else init.cpu().numpy(), length=length)
    assert np.array_equal(host, outs[0].cpu().numpy())


def test_the_loop_is_frame_major_by_default():
    """One launch of the transposition for the magnitudes, then two kernels per iteration: the launch counter tells the two
    loops apart (the reference-layout loop has no transposition)."""
    import torch
    c = Stft.Config.create(fft_size=2048, hop=512)
    mag = torch.rand(2, 1025, 40, device="cuda")
    counts = []
    for fm in (None, "0"):
        if fm is not None:
            os.environ["SMX_GL_FRAME_MAJOR"] = fm
        try:
            Stft.griffin_lim(c, mag, n_iter=3)
            torch.cuda.synchronize()
            l0 = lib.smx_debug_kernel_launches()
            Stft.griffin_lim(c, mag, n_iter=3)
            torch.cuda.synchronize()
            counts.append(lib.smx_debug_kernel_launches() - l0)
        finally:
            os.environ.pop("SMX_GL_FRAME_MAJOR", None)
    assert counts[0] == counts[1] + 1, counts

"""The float64 interior at fft 2048 (the reference's own numerics: stft.ml:345-346, 356-364, 670-674) has two kernels with the
same arithmetic: stft2048_power_wide_kernel (persistent workgroups, stft_wide_p64.hpp; the default) and the
one-tile-per-workgroup kernel it replaced (`SMX_WIDE_PIPELINE=0`).  The range / slice / partition laws (stft_grid.ml:58-73,
180-205) rest on every frame getting the same bits from either one, whatever the tile it sits in: checked here bit for bit over
ragged tiles, ranges that start mid-clip, every padding rule, odd hops, unaligned origins, leading axes and the three power
forms; and the new kernel against the oracle at the reference's float32 tolerance (rtol 1e-6, atol 1e-7)."""
import os

import numpy as np
import pytest

from oracle import soundml_oracle as O

import soundml_amd as S
from soundml_amd import Stft

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _float64_interior():
    S.set_interior("float64")
    os.environ.pop("SMX_WIDE_PIPELINE", None)
    yield
    os.environ.pop("SMX_WIDE_PIPELINE", None)
    S.set_interior("float32")


def both(fn):
    """fn() under the persistent kernel and under the older one"""
    os.environ.pop("SMX_WIDE_PIPELINE", None)
    new = fn()
    os.environ["SMX_WIDE_PIPELINE"] = "0"
    try:
        old = fn()
    finally:
        os.environ.pop("SMX_WIDE_PIPELINE", None)
    return new, old


GEOMETRIES = [
    # hop, win, alignment, pad, lead, n
    (512, None, "centered", "reflect", (3,), 16 * 512 * 3 + 100),
    (512, None, "centered", "reflect", (5,), 40000),
    (512, None, "left", "edge", (2, 3), 30001),
    (512, None, "right", ("constant", -0.75), (4,), 9000),
    (333, 1500, "centered", "reflect", (2,), 25000),        # odd hop: unaligned frame origins
    (1024, None, "centered", "edge", (300,), 2048 * 2 + 7),  # many short clips: ragged tiles of 5 frames
    (2048, 2048, "left", ("constant", 0.0), (1,), 2048 * 40),
    (100, 2000, "centered", "reflect", (1,), 5000),
    (512, None, "centered", "reflect", (1,), 1000),          # shorter than a frame: every frame touches both borders
]


@pytest.mark.parametrize("hop,win,alignment,pad,lead,n", GEOMETRIES)
def test_both_kernels_give_the_same_bits(hop, win, alignment, pad, lead, n):
    import torch
    torch.manual_seed(11)
    c = Stft.Config.create(fft_size=2048, hop=hop, win_length=win, alignment=alignment, pad=pad)
    x = (torch.rand(*lead, n, device="cuda") * 2 - 1).float()
    frames = Stft.frames(c, n)
    assert frames > 0
    ranges = {(0, frames), (min(3, frames - 1), frames), (0, max(1, frames - 2)), (frames // 2, frames // 2 + 1), (1 % frames, min(frames, 70))}
    for a, b in sorted(r for r in ranges if r[0] < r[1]):
        for p in (2.0, 1.0, 0.7):
            new, old = both(lambda: Stft.power_range(c, x, a, b, p))
            assert new.shape == old.shape == lead + (1025, b - a)
            assert torch.equal(new, old), "frames [%d, %d), power %g: the two kernels differ" % (a, b, p)
    # a range is the slice of the whole (stft_grid.ml:58-73), under the new kernel alone
    whole = Stft.power_spectrum(c, x, 2.0)
    a, b = frames // 3, max(frames // 3 + 1, 2 * frames // 3)
    assert torch.equal(Stft.power_range(c, x, a, b, 2.0), whole[..., a:b])
    # and a clip's values do not depend on its batch (stft_grid.ml:180-205)
    first = x.reshape(-1, n)[:1].contiguous()
    assert torch.equal(Stft.power_spectrum(c, first, 2.0)[0], whole.reshape(-1, 1025, frames)[0])


def test_against_the_oracle_at_the_reference_tolerance():
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, size=(3, 50000)).astype(np.float32)
    for kw in (dict(hop=512), dict(hop=700, win_length=1800, alignment="right", pad="edge")):
        c = Stft.Config.create(fft_size=2048, **kw)
        o = O.stft_config(2048, **kw)
        for p in (2.0, 1.0):
            got, want = Stft.power_spectrum(c, x, p), O.power_spectrum(o, x, p)
            assert got.shape == want.shape and got.dtype == np.float32
            np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-7 * float(np.max(np.abs(want))))


def test_streaming_chunks_total_the_offline_result():
    """Stft.power_stage under the float64 interior: chunks whose frames go through either kernel (a chunk's frames that touch
    its carry are few and ragged) reproduce the offline spectrogram bit for bit (stft_kernel.ml partition law)."""
    rng = np.random.default_rng(9)
    n = 60000
    x = rng.uniform(-1, 1, size=(2, n)).astype(np.float32)
    c = Stft.Config.create(fft_size=2048, hop=512)
    want = Stft.power_spectrum(c, x, 2.0)
    st = Stft.power_stage(c, 2.0).prepare(max_items=20000)
    parts, pos = [], 0
    for m in (1, 5000, 20000, 17, 9000, 20000, n):
        m = min(m, n - pos)
        if m <= 0:
            break
        out = st.step(x[..., pos:pos + m])
        pos += m
        if out is not None:
            parts.append(out)
    got = st.concat(parts + st.flush())
    assert got.shape == want.shape and np.array_equal(got, want)


def test_single_frames_odd_origins_and_many_clips():
    """one frame, a clip that starts on an odd float of its allocation (4-byte aligned samples only), a batch with more tiles than
    workgroups several times over: both kernels, same bits"""
    import torch
    torch.manual_seed(2)
    c = Stft.Config.create(fft_size=2048, hop=512, alignment="left", pad=("constant", 0.25))
    big = (torch.rand(3 * 2048 + 7, device="cuda") * 2 - 1).float()
    for off, n in ((1, 2048), (3, 2049), (0, 100), (5, 2048 * 3)):
        x = big[off:off + n]
        new, old = both(lambda: Stft.power_spectrum(c, x, 2.0))
        assert new.shape == old.shape and torch.equal(new, old), (off, n)
    c2 = Stft.Config.create(fft_size=2048, hop=512)
    x = (torch.rand(1200, 9000, device="cuda") * 2 - 1).float()      # 18 frames per clip: a full and a ragged tile, 2400 tiles
    new, old = both(lambda: Stft.power_spectrum(c2, x, 1.0))
    assert torch.equal(new, old)

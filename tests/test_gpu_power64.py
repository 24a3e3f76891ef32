"""The power spectrogram at fft 4096 on the register pipeline with a frame in a whole wave (stft4096_power64_kernel,
stft_fast_p64.hpp; round 6).  The reference reaches `analyse` at this size from cqt.ml:648 and hpss.ml:490-492.  Against the
float64 oracle, and the reference's structural laws bit for bit -- frame-range tiling across tile boundaries
(stft_grid.ml:32-73), batch == stack of slices (:180-205), the streaming partition law through the power stage
(stft_law.ml:79-164) --, border frames from gathered strips, ragged tiles, clips shorter than a frame, unaligned samples and
rows, every exponent of magnitude_pow, NaN propagation; and that the pipeline is the kernel that ran."""
import numpy as np
import pytest

import soundml_amd as S
from soundml_amd import Stft
from oracle import soundml_oracle as O

pytestmark = pytest.mark.gpu

REGRESSION = 2e-6     # of the spectrogram's peak, in amplitude terms (the gate of test_gpu_baseline_configs.py)
FFT = 4096


def _check(got, want, msg):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (msg, got.shape, want.shape)
    peak = float(np.max(np.abs(want))) if want.size else 0.0
    tol = 1e-5 * peak + 1e-5 * np.abs(want)                 # north-star contract
    assert np.all(np.abs(got - want) <= tol), (msg, float(np.max(np.abs(got - want))), peak)


@pytest.mark.parametrize("kw,n,lead", [
    (dict(hop=1024), 441000, 1),                             # 431 frames: odd rows
    (dict(hop=1024), 480000, 2),                             # 469 frames
    (dict(hop=1024), 8 * 1024 * 5 + 17, 2),                  # ragged last tile
    (dict(hop=1024), 8 * 1024 * 6, 3),                       # 49 frames: one past a whole tile
    (dict(hop=1023), 60000, 2),                              # odd hop: the unaligned load variant
    (dict(hop=800), 60000, 2),
    (dict(hop=512), 50000, 2),
    (dict(hop=1024, win_length=3000), 40000, 2),
    (dict(hop=1024, alignment="left"), 50000, 2),
    (dict(hop=1024, alignment="right", pad="edge"), 50000, 2),
    (dict(hop=1024, pad=("constant", 0.25)), 50000, 2),
    (dict(hop=4096), 100000, 2),                             # no overlap
    (dict(hop=5000), 100000, 2),                             # gaps between frames
    (dict(hop=1024), 3000, 3),                               # shorter than a frame: every frame touches both borders
    (dict(hop=1024), 1, 2),
])
@pytest.mark.parametrize("power", [2.0, 1.0])
def test_against_the_oracle(kw, n, lead, power):
    rng = np.random.default_rng(n + int(power))
    x = rng.uniform(-1, 1, size=(lead, n)).astype(np.float32)
    c = Stft.Config.create(fft_size=FFT, **kw)
    okw = dict(kw)
    if isinstance(okw.get("pad"), tuple):
        okw["pad"], okw["pad_value"] = okw["pad"]
    o = O.stft_config(FFT, **okw)
    got = Stft.power_spectrum(c, x, power)
    want = O.power_spectrum(o, x, power)
    assert got.dtype == np.float32
    for i in range(lead):
        _check(got[i], want[i], (kw, n, i))


def test_regression_gate_and_it_is_the_pipeline_that_ran():
    """16 clips of C1's length at fft 4096 / hop 1024: every value within 2e-6 of the peak in amplitude; ONE launch for the
    interior + one gather of the border strips; and the generic kernels (SMX_DISABLE_FAST) agree within the same gate -- two
    implementations of one contract."""
    import os
    import torch
    from soundml_amd._lib import lib
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, size=(16, 441000)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    c = Stft.Config.create(fft_size=FFT, hop=1024)
    Stft.power_spectrum(c, xd)
    torch.cuda.synchronize()
    l0 = lib.smx_debug_kernel_launches()
    got = Stft.power_spectrum(c, xd)
    torch.cuda.synchronize()
    assert lib.smx_debug_kernel_launches() - l0 == 2          # the strips' gather + the pipeline
    got = got.cpu().numpy()
    assert got.shape == (16, 2049, 431)
    o = O.stft_config(FFT, hop=1024)
    for i in (0, 15):
        want = O.power_spectrum(o, x[i])
        assert np.max(np.abs(got[i] - want)) <= 2 * REGRESSION * float(np.max(want)), i
    os.environ["SMX_DISABLE_FAST"] = "1"
    try:
        generic = Stft.power_spectrum(c, xd).cpu().numpy()
    finally:
        os.environ.pop("SMX_DISABLE_FAST", None)
    assert np.max(np.abs(generic - got)) <= 4 * REGRESSION * float(np.max(got))
    assert not np.array_equal(generic, got)                   # (different kernels: equal to rounding, not to the bit)


def test_ranges_tile_exactly_across_tile_boundaries():
    import torch
    x = torch.rand(3, 200000, device="cuda") * 2 - 1
    c = Stft.Config.create(fft_size=FFT, hop=1024)
    full = Stft.power_spectrum(c, x)
    total = Stft.frames(c, x.shape[-1])
    cuts = sorted({v for v in (0, 1, 2, 3, 7, 8, 9, 15, 16, 17, 24, 31, 33, 64, 65, 100, total - 17, total - 9, total - 8, total - 2, total) if 0 <= v <= total})
    parts = [Stft.power_range(c, x, a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    assert torch.equal(torch.cat(parts, dim=-1), full)
    assert np.array_equal(full.cpu().numpy(), Stft.power_spectrum(c, x.cpu().numpy()))      # device path == host path
    for power in (1.0, 0.5):
        f = Stft.power_spectrum(c, x, power)
        assert torch.equal(torch.cat([Stft.power_range(c, x, a, b, power) for a, b in zip(cuts[:-1], cuts[1:])], dim=-1), f)


def test_batch_is_the_stack_of_its_slices():
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, size=(2, 3, 12 * 4096 + 5)).astype(np.float32)
    for hop in (1024, 1023):
        c = Stft.Config.create(fft_size=FFT, hop=hop)
        full = Stft.power_spectrum(c, x)
        for i in range(2):
            for j in range(3):
                assert np.array_equal(full[i, j], Stft.power_spectrum(c, x[i, j]))


def test_many_short_clips_agree_with_small_batches():
    """3000 clips of 8 frames each, 4 touching a border: the border frames come from gathered strips; the same clips in a batch of
    50 run the identical frame code."""
    import torch
    rng = np.random.default_rng(11)
    x = torch.from_numpy(rng.uniform(-1, 1, size=(3000, 8000)).astype(np.float32)).cuda()
    c = Stft.Config.create(fft_size=FFT, hop=1024)
    p = Stft.power_spectrum(c, x)
    assert tuple(p.shape) == (3000, 2049, 8)
    for lo in (0, 1475, 2950):
        assert torch.equal(p[lo:lo + 50], Stft.power_spectrum(c, x[lo:lo + 50])), lo
    o = O.stft_config(FFT, hop=1024)
    for clip in (0, 1321, 2999):
        want = O.power_spectrum(o, x[clip].cpu().numpy())
        assert np.max(np.abs(p[clip].cpu().numpy() - want)) <= 2 * REGRESSION * float(np.max(want)), clip


@pytest.mark.parametrize("alignment,pad", [("centered", "reflect"), ("left", "edge"), ("right", ("constant", 0.5))])
def test_power_stage_streams_the_offline_result(alignment, pad):
    rng = np.random.default_rng(5)
    n = 40 * 4096 + 333
    x = rng.standard_normal((2, n)).astype(np.float32)
    c = Stft.Config.create(fft_size=FFT, hop=1024, alignment=alignment, pad=pad)
    for power in (2.0, 1.0):
        offline = Stft.power_spectrum(c, x, power)
        for block in (n, 30000, 5000, 1031):
            st = Stft.power_stage(c, power).prepare(max_items=block)
            parts = []
            for i in range(0, n, block):
                out = st.step(x[:, i:i + block])
                if out is not None:
                    parts.append(out)
            got = st.concat(parts + st.flush())
            assert np.array_equal(got, offline), (alignment, power, block)


@pytest.mark.parametrize("power", [0.5, 3.0, 0.0])
def test_general_powers(power):
    rng = np.random.default_rng(int(power * 10) + 3)
    x = rng.uniform(-1, 1, size=(2, 60000)).astype(np.float32)
    x[1] = 0.0
    c = Stft.Config.create(fft_size=FFT, hop=1024)
    got = Stft.power_spectrum(c, x, power)
    want = O.power_spectrum(O.stft_config(FFT, hop=1024), x, power)
    _check(got[0], want[0], power)
    assert np.all(got[1] == (0.0 if power > 0 else 1.0))


def test_nan_sample_propagates():
    """A NaN sample reaches every bin of the frames that cover it and no other frame (odd-length rows -- the 4-byte-aligned load
    variant -- are in test_against_the_oracle's matrix)."""
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, size=(2, 80000)).astype(np.float32)
    x[0, 40000] = np.nan
    c = Stft.Config.create(fft_size=FFT, hop=1024)
    got = Stft.power_spectrum(c, x)
    frames = Stft.frames(c, x.shape[-1])
    covered = [p for p in range(frames) if p * 1024 - 2048 <= 40000 < p * 1024 - 2048 + 4096]
    for p in range(frames):
        assert np.isnan(got[0, :, p]).all() == (p in covered), p
        assert not np.isnan(got[0, :, p]).any() or p in covered
    assert np.isfinite(got[1]).all()


@pytest.mark.parametrize("n,kw", [
    (1024 * 37 + 500, dict(hop=1024)),                          # 38 frames a clip: even rows, a ragged last tile (6 frames)
    (1024 * 38 + 500, dict(hop=1024)),                          # 39 frames: odd rows
    (61000, dict(hop=1024, alignment="left", pad="edge")),      # 56 frames
    (512 * 61, dict(hop=512)),                                  # 62 frames at hop 512
])
def test_a_big_batch_walked_side_by_side_equals_its_clips(n, kw):
    """600 clips: the workgroups walk the flat (clip, tile) sequence side by side (several tiles and clips per workgroup, neighbouring
    tiles on neighbouring workgroups).  Every clip of the batch equals the same clip computed alone, bit for bit, for whole batches and
    ranges that begin and end inside clips and tiles; three clips against the oracle."""
    import torch
    torch.manual_seed(n)
    x = (torch.rand(600, n, device="cuda") * 2 - 1).float()
    c = Stft.Config.create(fft_size=FFT, **kw)
    frames = Stft.frames(c, n)
    calls = [(0, frames, 2.0), (0, frames, 0.7), (2, frames - 2, 2.0), (7, frames, 1.0), (1, frames - 2, 2.0)]
    for a, b, p in calls:
        full = Stft.power_range(c, x, a, b, p)
        for clip in (0, 1, 255, 256, 333, 599):
            assert torch.equal(full[clip], Stft.power_range(c, x[clip], a, b, p)), (a, b, p, clip)
    okw = dict(kw)
    if isinstance(okw.get("pad"), tuple):
        okw["pad"], okw["pad_value"] = okw["pad"]
    o = O.stft_config(FFT, **okw)
    full = Stft.power_spectrum(c, x).cpu().numpy()
    for clip in (0, 299, 599):
        want = O.power_spectrum(o, x[clip].cpu().numpy())
        assert np.max(np.abs(full[clip] - want)) <= 2 * REGRESSION * float(np.max(want)), clip

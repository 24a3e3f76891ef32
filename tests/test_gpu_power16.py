"""The power spectrogram at fft 1024, fft 512 and fft 256 on the register pipeline with a frame in 16 / 8 / 4 lanes
(stft_power_lanes_kernel, stft_fast_p16.hpp): BASELINE C1's geometry (fft 1024 / hop 256), fft 512 / hop 128 and their
neighbours against the float64 oracle, and the reference's structural
laws bit for bit -- frame-range tiling across tile boundaries (stft_grid.ml:32-73), batch == stack of slices (:180-205),
the streaming partition law through the power stage (stft_law.ml:79-164) --, border frames by the kernel's epilogue and by
gathered strips, ragged tiles, clips shorter than a frame, unaligned samples, every exponent of magnitude_pow."""
import numpy as np
import pytest

import soundml_amd as S
from soundml_amd import Stft
from oracle import soundml_oracle as O

pytestmark = pytest.mark.gpu

REGRESSION = 2e-6     # of the spectrogram's peak, in amplitude terms (the gate of test_gpu_baseline_configs.py)


def _check(got, want, power, msg):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (msg, got.shape, want.shape)
    peak = float(np.max(np.abs(want))) if want.size else 0.0
    tol = 1e-5 * peak + 1e-5 * np.abs(want)                 # north-star contract
    assert np.all(np.abs(got - want) <= tol), (msg, float(np.max(np.abs(got - want))), peak)


@pytest.mark.parametrize("kw,n,lead", [
    (dict(hop=256), 441000, 1),                              # C1: 1 x 441 000 samples -> 513 x 1723
    (dict(hop=256), 44100, 3),
    (dict(hop=256), 32 * 256 * 2 + 17, 2),                   # ragged last tile
    (dict(hop=255), 30000, 2),                               # odd hop: the unaligned load variant
    (dict(hop=200), 30000, 2),
    (dict(hop=256, win_length=800), 20000, 2),
    (dict(hop=256, alignment="left"), 20000, 2),
    (dict(hop=256, alignment="right", pad="edge"), 20000, 2),
    (dict(hop=256, pad=("constant", 0.25)), 20000, 2),
    (dict(hop=1024), 50000, 2),                              # no overlap
    (dict(hop=256), 700, 3),                                 # shorter than a frame: every frame touches both borders
    (dict(hop=256), 1, 2),
])
@pytest.mark.parametrize("power", [2.0, 1.0])
@pytest.mark.parametrize("fft", [1024, 512, 256])
def test_against_the_oracle(fft, kw, n, lead, power):
    rng = np.random.default_rng(n + int(power))
    kw = dict(kw)
    for _ in range({1024: 0, 512: 1, 256: 2}[fft]):         # the same shapes at half / a quarter of the size
        kw["hop"] = max(1, kw["hop"] // 2) if kw["hop"] % 2 == 0 else kw["hop"] // 2 | 1
        if "win_length" in kw:
            kw["win_length"] //= 2
        n = n if n < 1000 else n // 2
    if fft == 256 and n >= 1000:
        n *= 2                                              # (128 frames a tile: keep several tiles per clip)
    x = rng.uniform(-1, 1, size=(lead, n)).astype(np.float32)
    c = Stft.Config.create(fft_size=fft, **kw)
    okw = dict(kw)
    if isinstance(okw.get("pad"), tuple):
        okw["pad"], okw["pad_value"] = okw["pad"]
    o = O.stft_config(fft, **okw)
    got = Stft.power_spectrum(c, x, power)
    want = O.power_spectrum(o, x, power)
    assert got.dtype == np.float32
    for i in range(lead):
        _check(got[i], want[i], power, (kw, n, i))


def test_regression_gate_on_a_c1_batch():
    """8 clips of C1's length: every value within 2e-6 of the peak in amplitude (power: twice that of the power peak)."""
    import torch
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, size=(8, 441000)).astype(np.float32)
    c = Stft.Config.create(fft_size=1024, hop=256)
    got = Stft.power_spectrum(c, torch.from_numpy(x).cuda()).cpu().numpy()
    assert got.shape == (8, 513, 1723)
    o = O.stft_config(1024, hop=256)
    for i in (0, 7):
        want = O.power_spectrum(o, x[i])
        assert np.max(np.abs(got[i] - want)) <= 2 * REGRESSION * float(np.max(want)), i


@pytest.mark.parametrize("fft", [1024, 512, 256])
def test_ranges_tile_exactly_across_tile_boundaries(fft):
    import torch
    x = torch.rand(3, 70000, device="cuda") * 2 - 1
    c = Stft.Config.create(fft_size=fft, hop=fft // 4)
    full = Stft.power_spectrum(c, x)
    total = Stft.frames(c, x.shape[-1])
    cuts = sorted({v for v in (0, 1, 2, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, total - 129, total - 65, total - 33, total - 2, total) if 0 <= v <= total})
    parts = [Stft.power_range(c, x, a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    assert torch.equal(torch.cat(parts, dim=-1), full)
    assert np.array_equal(full.cpu().numpy(), Stft.power_spectrum(c, x.cpu().numpy()))      # device path == host path
    for power in (1.0, 0.5):
        f = Stft.power_spectrum(c, x, power)
        assert torch.equal(torch.cat([Stft.power_range(c, x, a, b, power) for a, b in zip(cuts[:-1], cuts[1:])], dim=-1), f)


@pytest.mark.parametrize("fft", [1024, 512, 256])
def test_batch_is_the_stack_of_its_slices(fft):
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, size=(2, 3, 4 * 1024 + 5)).astype(np.float32)
    for hop in (fft // 4, fft // 4 - 1):
        c = Stft.Config.create(fft_size=fft, hop=hop)
        full = Stft.power_spectrum(c, x)
        for i in range(2):
            for j in range(3):
                assert np.array_equal(full[i, j], Stft.power_spectrum(c, x[i, j]))


@pytest.mark.parametrize("fft", [1024, 512, 256])
def test_many_short_clips_take_the_strip_path_and_agree(fft):
    """8000 clips of 8 frames each, 4 touching a border -- above the launcher's epilogue threshold, so the kernel reads the
    border frames from gathered strips; the same clips in a batch of 50 take the epilogue: identical frame code."""
    import torch
    rng = np.random.default_rng(11)
    x = torch.from_numpy(rng.uniform(-1, 1, size=(8000, 2000 * fft // 1024)).astype(np.float32)).cuda()
    c = Stft.Config.create(fft_size=fft, hop=fft // 4)
    p = Stft.power_spectrum(c, x)
    assert tuple(p.shape) == (8000, fft // 2 + 1, 8)
    for lo in (0, 3950, 7950):
        assert torch.equal(p[lo:lo + 50], Stft.power_spectrum(c, x[lo:lo + 50])), lo
    o = O.stft_config(fft, hop=fft // 4)
    for clip in (0, 4321, 7999):
        want = O.power_spectrum(o, x[clip].cpu().numpy())
        assert np.max(np.abs(p[clip].cpu().numpy() - want)) <= 2 * REGRESSION * float(np.max(want)), clip


@pytest.mark.parametrize("alignment,pad", [("centered", "reflect"), ("left", "edge"), ("right", ("constant", 0.5))])
@pytest.mark.parametrize("fft", [1024, 512, 256])
def test_power_stage_streams_the_offline_result(fft, alignment, pad):
    rng = np.random.default_rng(5)
    n = 40 * 1024 + 333
    x = rng.standard_normal((2, n)).astype(np.float32)
    c = Stft.Config.create(fft_size=fft, hop=fft // 4, alignment=alignment, pad=pad)
    for power in (2.0, 1.0):
        offline = Stft.power_spectrum(c, x, power)
        for block in (n, 9000, 1000, 257):
            st = Stft.power_stage(c, power).prepare(max_items=block)
            parts = []
            for i in range(0, n, block):
                out = st.step(x[:, i:i + block])
                if out is not None:
                    parts.append(out)
            got = st.concat(parts + st.flush())
            assert np.array_equal(got, offline), (alignment, power, block)


@pytest.mark.parametrize("power", [0.5, 3.0, 0.0])
@pytest.mark.parametrize("fft", [1024, 512, 256])
def test_general_powers(fft, power):
    rng = np.random.default_rng(int(power * 10) + 3)
    x = rng.uniform(-1, 1, size=(2, 20000)).astype(np.float32)
    x[1] = 0.0
    c = Stft.Config.create(fft_size=fft, hop=fft // 4)
    got = Stft.power_spectrum(c, x, power)
    want = O.power_spectrum(O.stft_config(fft, hop=fft // 4), x, power)
    _check(got[0], want[0], power, power)
    assert np.all(got[1] == (0.0 if power > 0 else 1.0))


# ---- the fused mel spectrogram on the same pipeline (fft 1024: stft_mel_lanes_kernel) ---------------------------------------

@pytest.mark.parametrize("n_mels,sr,n,lead,power", [
    (80, 22050, 441000 // 4, 3, 2.0),            # the vocoder front end: fft 1024 / hop 256 / 80 mels / 22.05 kHz
    (128, 44100, 60000, 2, 2.0),
    (80, 22050, 32 * 256 * 2 + 17, 2, 1.0),      # ragged last tile, magnitude
    (64, 16000, 700, 3, 2.0),                    # shorter than a frame: the strip path only
    (20, 16000, 30000, 2, 2.0),                  # a plan with a long wave: stays on the 16-frame kernels of stft_generic.hip
    (96, 32000, 50000, 1, 0.7),
])
def test_fused_mel_at_fft_1024(n_mels, sr, n, lead, power):
    """Soundml.mel_spectrogram = Mel.apply . power_spectrum (soundml.ml:12-24) at fft 1024 / hop 256 against the oracle, batch ==
    stack of slices bit for bit (mel_props.ml:136-155), device path == host path."""
    import torch
    from soundml_amd import Mel
    rng = np.random.default_rng(n_mels + n)
    x = rng.uniform(-1, 1, size=(lead, n)).astype(np.float32)
    sc = Stft.Config.create(fft_size=1024, hop=256)
    mc = Mel.Config.create(n_mels=n_mels, sample_rate=sr, fft_size=1024)
    got = S.mel_spectrogram(sc, mc, x, power)
    want = O.mel_spectrogram(O.stft_config(1024, hop=256), O.mel_config(n_mels, sr, 1024), x, power)
    assert got.shape == want.shape and got.dtype == np.float32
    for i in range(lead):
        _check(got[i], want[i], power, (n_mels, sr, n, i))
        assert np.array_equal(got[i], S.mel_spectrogram(sc, mc, x[i], power))
    assert np.array_equal(S.mel_spectrogram(sc, mc, torch.from_numpy(x).cuda(), power).cpu().numpy(), got)


def test_fused_mel_at_fft_1024_many_short_clips():
    """4000 clips of 8 frames: border frames from gathered strips in a big batch; the same clips in a batch of 50 agree bit for bit."""
    import torch
    from soundml_amd import Mel
    rng = np.random.default_rng(12)
    x = torch.from_numpy(rng.uniform(-1, 1, size=(4000, 2000)).astype(np.float32)).cuda()
    sc = Stft.Config.create(fft_size=1024, hop=256)
    mc = Mel.Config.create(n_mels=80, sample_rate=22050, fft_size=1024)
    m = S.mel_spectrogram(sc, mc, x)
    assert tuple(m.shape) == (4000, 80, 8)
    for lo in (0, 1950, 3950):
        assert torch.equal(m[lo:lo + 50], S.mel_spectrogram(sc, mc, x[lo:lo + 50])), lo
    want = O.mel_spectrogram(O.stft_config(1024, hop=256), O.mel_config(80, 22050, 1024), x[1234].cpu().numpy())
    _check(m[1234].cpu().numpy(), want, 2.0, "clip 1234")


# ---- Stft.transform on the same pipeline (fft 2048: stft2048_complex32_kernel; fft 1024 / 512: stft_complex_lanes_kernel) -----

def _check_c(got, want, msg):
    got, want = np.asarray(got, dtype=np.complex128), np.asarray(want, dtype=np.complex128)
    assert got.shape == want.shape, (msg, got.shape, want.shape)
    peak = float(np.max(np.abs(want))) if want.size else 0.0
    assert np.all(np.abs(got - want) <= 1e-5 * peak + 1e-5 * np.abs(want)), (msg, float(np.max(np.abs(got - want))), peak)


@pytest.mark.parametrize("fft", [2048, 1024, 512, 256])
@pytest.mark.parametrize("kw,n,lead", [
    (dict(), 60000, 2),
    (dict(hop_odd=True), 30000, 2),                          # odd hop: the unaligned load variant
    (dict(alignment="left", pad="edge"), 20011, 2),
    (dict(alignment="right", pad=("constant", 0.25)), 20000, 1),
    (dict(), 300, 3),                                        # shorter than a frame
])
def test_transform_against_the_oracle(fft, kw, n, lead):
    rng = np.random.default_rng(fft + n)
    kw = dict(kw)
    hop = fft // 4 - 1 if kw.pop("hop_odd", False) else fft // 4
    x = rng.uniform(-1, 1, size=(lead, n)).astype(np.float32)
    c = Stft.Config.create(fft_size=fft, hop=hop, **kw)
    okw = dict(kw)
    if isinstance(okw.get("pad"), tuple):
        okw["pad"], okw["pad_value"] = okw["pad"]
    want = O.transform(O.stft_config(fft, hop=hop, **okw), x)
    got = Stft.transform(c, x)
    assert got.dtype == np.complex64
    for i in range(lead):
        _check_c(got[i], want[i], (fft, kw, n, i))
        assert np.array_equal(got[i], Stft.transform(c, x[i]))                               # batch == stack of slices
    assert np.all(got[..., 0, :].imag == 0) and np.all(got[..., -1, :].imag == 0)            # DC and Nyquist bins are real


@pytest.mark.parametrize("fft", [2048, 1024, 512, 256])
def test_transform_ranges_streaming_and_device(fft):
    import torch
    rng = np.random.default_rng(fft)
    n = (30 if fft > 256 else 80) * fft + 77
    x = rng.standard_normal((2, n)).astype(np.float32)
    c = Stft.Config.create(fft_size=fft, hop=fft // 4)
    full = Stft.transform(c, x)
    total = Stft.frames(c, n)
    cuts = [0, 1, 15, 16, 17, 31, 32, 33, 64, 65, total - 3, total] if fft > 256 else [0, 1, 64, 127, 128, 129, 200, 256, 257, total - 3, total]
    parts = [Stft.transform_range(c, x, a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate(parts, axis=-1), full)                              # stft_grid.ml:32-73
    assert np.array_equal(Stft.transform(c, torch.from_numpy(x).cuda()).cpu().numpy(), full)  # device path == host path
    for block in (n, 5000, 777):                                                             # stft_law.ml:79-164 through Stft.stage
        st = Stft.stage(c).prepare(max_items=block)
        got = []
        for i in range(0, n, block):
            out = st.step(x[:, i:i + block])
            if out is not None:
                got.append(out)
        assert np.array_equal(st.concat(got + st.flush()), full), block


def test_transform_many_short_clips():
    import torch
    rng = np.random.default_rng(13)
    for fft in (2048, 1024, 512, 256):
        x = torch.from_numpy(rng.uniform(-1, 1, size=(3000, 2 * fft - 48)).astype(np.float32)).cuda()
        c = Stft.Config.create(fft_size=fft, hop=fft // 4)
        z = Stft.transform(c, x)
        for lo in (0, 1475, 2950):
            assert torch.equal(z[lo:lo + 50], Stft.transform(c, x[lo:lo + 50])), (fft, lo)
        _check_c(z[777].cpu().numpy(), O.transform(O.stft_config(fft, hop=fft // 4), x[777].cpu().numpy()), (fft, 777))


def test_fft_256_is_one_launch_of_the_four_lane_pipeline():
    """fft 256 / hop 64 on a device-resident batch: one kernel launch for interior and border frames alike (the border frames ride
    in the tile sequence), every value within the regression gate of the float64 oracle."""
    import torch
    from soundml_amd._lib import lib
    rng = np.random.default_rng(256)
    x = rng.uniform(-1, 1, size=(6, 100000)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    c = Stft.Config.create(fft_size=256, hop=64)
    Stft.power_spectrum(c, xd)
    torch.cuda.synchronize()
    l0 = lib.smx_debug_kernel_launches()
    got = Stft.power_spectrum(c, xd)
    torch.cuda.synchronize()
    assert lib.smx_debug_kernel_launches() - l0 == 1
    assert tuple(got.shape) == (6, 129, 1563)
    o = O.stft_config(256, hop=64)
    for i in (0, 5):
        want = O.power_spectrum(o, x[i])
        assert np.max(np.abs(got[i].cpu().numpy() - want)) <= 2 * REGRESSION * float(np.max(want)), i


@pytest.mark.parametrize("n,kw", [
    (256 * 32 * 3 + 5, dict(hop=256)),                          # 97 frames a clip: odd rows, a ragged last tile
    (256 * 32 * 4, dict(hop=256)),                              # 129 frames
    (256 * 32 * 3 + 200, dict(hop=255)),                        # odd hop
    (30000, dict(hop=256, alignment="left", pad="edge")),
    (30000, dict(hop=256, alignment="right", pad=("constant", 0.5))),
    (256 * 32 * 2 + 3 * 256, dict(hop=256)),                    # 76 frames: even rows
])
def test_fft_1024_aligned_lines_equal_the_plain_flush(n, kw):
    """fft 1024: the flush in whole aligned 128-byte lines (stft_fast_p16.hpp SkL: a frame per lane, a row's trailing frames carried to
    the next tile) is taken when a launch's workgroups hold several tiles each -- 520 clips here.  Bit for bit the plain flush's values
    (SMX_POWER_SKEW=0), whole batches and ranges that begin and end inside clips and tiles, every exponent form; three clips against the oracle."""
    import os
    import torch
    torch.manual_seed(n)
    x = (torch.rand(520, n, device="cuda") * 2 - 1).float()
    c = Stft.Config.create(fft_size=1024, **kw)
    frames = Stft.frames(c, n)
    calls = [(0, frames, 2.0), (0, frames, 1.0), (0, frames, 0.7), (3, frames - 2, 2.0), (31, 66, 2.0), (33, frames, 1.0)]
    got = {}
    for mode in ("1", "0"):
        os.environ["SMX_POWER_SKEW"] = mode
        try:
            got[mode] = [Stft.power_range(c, x, a, b, p) for a, b, p in calls]
        finally:
            os.environ.pop("SMX_POWER_SKEW", None)
    for g1, g0, call in zip(got["1"], got["0"], calls):
        assert torch.equal(g1, g0), call
    okw = dict(kw)
    if isinstance(okw.get("pad"), tuple):
        okw["pad"], okw["pad_value"] = okw["pad"]
    o = O.stft_config(1024, **okw)
    full = got["1"][0].cpu().numpy()
    for clip in (0, 259, 519):
        want = O.power_spectrum(o, x[clip].cpu().numpy())
        assert np.max(np.abs(full[clip] - want)) <= 2 * REGRESSION * float(np.max(want)), clip

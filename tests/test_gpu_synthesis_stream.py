"""Stft.Synthesis on the device under the laws of soundml/test/istft/sistft_law.ml: partition invariance bit for bit, the
one-batch instance is Stft.invert at its default length bit for bit, the stream totals invert's default length, the
lookahead is Config.synthesis_latency (brute-forced), and the prepare / step / drain contracts with the reference's
messages -- on the reference's configuration grid and chunkings (sistft_law.ml:79-131), both dtype pairings, a batched
leading axis, plus the geometries the fused device kernels serve (fft 2048 / hop 512, fft 1024 / hop 256 and 512, fft 512 /
hop 128) and device-resident chunks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import soundml_oracle as O

import soundml_amd as S
from soundml_amd import Stft

GRID = [   # sistft_law.ml:79-112
    ("centered fft8 hop2", dict(fft_size=8, hop=2)),
    ("centered fft8 hop3 non-divisible", dict(fft_size=8, hop=3)),
    ("centered fft9 hop2 odd", dict(fft_size=9, hop=2)),
    ("centered fft9 hop4 odd", dict(fft_size=9, hop=4)),
    ("left fft8 hop3", dict(alignment="left", fft_size=8, hop=3)),
    ("left fft16 hop4 win10", dict(alignment="left", fft_size=16, hop=4, win_length=10)),
    ("right fft8 hop3", dict(alignment="right", fft_size=8, hop=3)),
    ("right fft9 hop4 win7", dict(alignment="right", fft_size=9, hop=4, win_length=7)),
    ("centered fft16 hop5 win12", dict(fft_size=16, hop=5, win_length=12)),
    ("centered fft32 hop7 win21", dict(fft_size=32, hop=7, win_length=21)),
    ("centered fft8 hop5 tail-held", dict(fft_size=8, hop=5)),
    ("centered fft8 hop6 tail-held", dict(fft_size=8, hop=6)),
    ("centered fft16 hop11 tail-held", dict(fft_size=16, hop=11)),
    ("centered fft16 hop4 magnitude", dict(fft_size=16, hop=4, scale="magnitude")),
    ("centered fft16 hop4 hamming", dict(fft_size=16, hop=4, window="hamming")),
]
BIG = [
    ("fused fft2048 hop512", dict(fft_size=2048, hop=512)),
    ("fused fft1024 hop256 left", dict(fft_size=1024, hop=256, alignment="left")),
    ("fused fft1024 hop512", dict(fft_size=1024, hop=512)),
    ("fused fft512 hop128 right", dict(fft_size=512, hop=128, alignment="right")),
    ("generic fft2048 hop500 win1200", dict(fft_size=2048, hop=500, win_length=1200)),
    ("mixed fft400 hop160", dict(fft_size=400, hop=160)),
]


def chunkings(total):   # sistft_law.ml:116-131
    def pieces(sizes, left):
        out = []
        for s in sizes:
            if left <= 0:
                break
            s = min(s, left)
            out.append(s)
            left -= s
        if left > 0:
            out.append(left)
        return out
    return [("whole", [total]), ("ones", [1] * total), ("primes", pieces([2, 3, 5, 7, 11, 13, 2, 3, 5, 7], total)),
            ("giant-then-ones", pieces([max(1, total - 3)] + [1] * 8, total)),
            ("irregular", pieces([1, 4, 1, 1, 9, 2, 1, 6, 3, 1, 1, 1], total)), ("straddle", pieces([3, 1, 4, 1, 5, 9, 2, 6], total))]


def drive(c, z, sizes, device=False):
    """everything a fresh kernel emits for the frame batches `sizes`, drain included"""
    lead = int(np.prod(z.shape[:-2])) if z.ndim > 2 else 1
    k = Stft.Synthesis.prepare(c, z.dtype, channels=lead, max_block=max([1] + list(sizes)))
    zz = z.reshape(lead, z.shape[-2], z.shape[-1])
    if device:
        import torch
        zz = torch.from_numpy(zz).cuda()
    outs, pos = [], 0
    for s in sizes:
        out = k.step(zz[:, :, pos:pos + s])
        pos += s
        if out is not None:
            outs.append(out.cpu().numpy() if device else out)
    out = k.flush()   # (on the device when the stream was fed from the device, as Kernel.flush)
    if out is not None:
        assert hasattr(out, "is_cuda") == device
        outs.append(out.cpu().numpy() if device else out)
    real = np.float64 if z.dtype == np.complex128 else np.float32
    got = np.concatenate(outs, axis=-1) if outs else np.zeros((lead, 0), dtype=real)
    return got.reshape(z.shape[:-2] + (got.shape[-1],))


def check_spectrum(c, label, z, device_too=False):   # sistft_law.ml:166-189
    total = z.shape[-1]
    offline = Stft.invert(c, z)
    one_shot = drive(c, z, [total])
    assert one_shot.shape == offline.shape, (label, one_shot.shape, offline.shape)
    assert np.array_equal(one_shot, offline), label + ": one batch is invert"
    for name, sizes in chunkings(total):
        assert np.array_equal(drive(c, z, sizes), one_shot), "%s: chunking %s" % (label, name)
    if device_too:
        assert np.array_equal(drive(c, z, chunkings(total)[2][1], device=True), one_shot), label + ": device-resident chunks"


def synthetic(c, frames):   # sistft_law.ml:194-200
    bins = c.bins
    v = O.lcg_signal(2 * bins * frames)
    re = v[:bins * frames].reshape(bins, frames)
    im = v[bins * frames:].reshape(bins, frames)
    return (re + 1j * im).astype(np.complex128)


@pytest.mark.parametrize("name,kw", GRID, ids=[g[0] for g in GRID])
def test_partition_law_on_the_reference_grid(name, kw):
    c = Stft.Config.create(**kw)
    for n in (1, 5, 17, 61):
        sig = O.lcg_signal(n)
        check_spectrum(c, "f64/n=%d" % n, Stft.transform(c, sig[None, :])[0])
        check_spectrum(c, "f32/n=%d" % n, Stft.transform(c, sig.astype(np.float32)[None, :])[0])
    for frames in (1, 5, 17, 61):
        z = synthetic(c, frames)
        check_spectrum(c, "synthetic/f64/frames=%d" % frames, z)
        check_spectrum(c, "synthetic/f32/frames=%d" % frames, z.astype(np.complex64))
    check_spectrum(c, "batch [2;40]", Stft.transform(c, O.lcg_signal(80).reshape(2, 40)))


@pytest.mark.parametrize("name,kw", BIG, ids=[g[0] for g in BIG])
def test_partition_law_on_the_device_kernels_geometries(name, kw):
    c = Stft.Config.create(**kw)
    rng = np.random.default_rng(len(name))
    x = rng.uniform(-1, 1, size=(2, 20000)).astype(np.float32)
    check_spectrum(c, name + " f32 transform", Stft.transform(c, x), device_too=True)
    z = (rng.standard_normal((c.bins, 21)) + 1j * rng.standard_normal((c.bins, 21)))
    check_spectrum(c, name + " f32 synthetic", z.astype(np.complex64))
    check_spectrum(c, name + " f64 synthetic", z)


def test_totals_latency_and_contracts():
    """sistft_law.ml:229-300 (totals, empty spectrum, the lookahead brute-forced) and :320-375 (contracts)."""
    for name, kw in GRID:
        c = Stft.Config.create(**kw)
        o = O.stft_config(kw["fft_size"], hop=kw["hop"], win_length=kw.get("win_length"), alignment=kw.get("alignment", "centered"))
        for frames in (1, 2, 3, 7, 40):
            z = synthetic(c, frames)
            assert drive(c, z, [1] * frames).shape[-1] == Stft.invert(c, z).shape[-1] == O.output_length(o, frames), (name, frames)
        k = Stft.Synthesis.prepare(c, np.complex128, channels=1, max_block=4)
        assert k.step(np.zeros((c.bins, 0), np.complex128)) is None and k.flush() is None
        # one frame per step against the naive rate map: the largest deficit is the latency, the first sample lands at latency / hop
        z = synthetic(c, 40)
        k = Stft.Synthesis.prepare(c, np.complex128, channels=1, max_block=1)
        emitted, worst, first = 0, 0, -1
        for p in range(40):
            out = k.step(z[:, p:p + 1])
            if out is not None:
                emitted += out.shape[-1]
            if emitted > 0 and first < 0:
                first = p
            worst = max(worst, (p + 1) * c.hop - emitted)
        lat = Stft.Synthesis.latency(c)
        assert lat == Stft.left_width(c) + max(0, c.hop + Stft.right_width(c) - c.fft_size)
        assert (worst, first) == (lat, lat // c.hop), (name, worst, first, lat)
    c8 = Stft.Config.create(fft_size=8, hop=2)
    with pytest.raises(S.InvalidArgument, match=r"prepare: cannot synthesise 0 channels \(channels must be at least 1\)"):
        Stft.Synthesis.prepare(c8, np.complex128, channels=0, max_block=8)
    with pytest.raises(S.InvalidArgument, match=r"prepare: cannot accept blocks of 0 frames \(max_block must be at least 1\)"):
        Stft.Synthesis.prepare(c8, np.complex128, channels=1, max_block=0)
    gap = Stft.Config.create(fft_size=8, hop=8, win_length=4)
    with pytest.raises(S.InvalidArgument, match="prepare: cannot invert a 4-point window advanced by 8 samples inside a 8-point frame"):
        Stft.Synthesis.prepare(gap, np.complex128, channels=1, max_block=8)
    k = Stft.Synthesis.prepare(c8, np.complex128, channels=1, max_block=8)
    with pytest.raises(S.InvalidArgument, match="step: cannot invert 4 frequency bins of a 8-point transform"):
        k.step(np.zeros((4, 3), np.complex128))
    with pytest.raises(S.InvalidArgument, match="step: cannot invert a rank-1 tensor"):
        k.step(np.zeros((5,), np.complex128))
    z = synthetic(c8, 6)
    assert k.step(z) is not None and k.flush() is not None and k.flush() is None
    with pytest.raises(S.InvalidArgument, match="step: cannot feed a drained kernel"):
        k.step(z)
    k.reset()
    assert np.array_equal(np.concatenate([k.step(z)[0], k.flush()[0]]), Stft.invert(c8, z))


def test_step_checks_the_channels_and_flush_stays_on_the_device():
    """A chunk must hold exactly the prepared channels (the library reads channels x bins rows of k frames: fewer would be
    read past their end, more silently dropped) on both the host and the device path; a stream fed from the device gets
    its tail on the device, equal to the host path's."""
    import torch
    c = Stft.Config.create(fft_size=64, hop=16)
    rng = np.random.default_rng(5)
    z = (rng.standard_normal((2, c.bins, 9)) + 1j * rng.standard_normal((2, c.bins, 9))).astype(np.complex64)
    k2 = Stft.Synthesis.prepare(c, np.complex64, channels=2, max_block=16)
    for bad in (z[0], z[:1], np.concatenate([z, z[:1]])):
        with pytest.raises(S.InvalidArgument, match="step: the chunk holds"):
            k2.step(bad)
        with pytest.raises(S.InvalidArgument, match="step: the chunk holds"):
            k2.step(torch.from_numpy(np.ascontiguousarray(bad)).cuda())
    host = [k2.step(z), k2.flush()]
    k2.reset()
    dev = [k2.step(torch.from_numpy(z).cuda()), k2.flush()]
    assert all(isinstance(d, torch.Tensor) and d.is_cuda for d in dev if d is not None)
    cat = lambda parts: np.concatenate([np.asarray(p.cpu() if hasattr(p, "cpu") else p) for p in parts if p is not None], axis=-1)
    assert np.array_equal(cat(host), cat(dev))
    assert np.array_equal(cat(host), Stft.invert(c, z))

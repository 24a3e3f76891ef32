"""`Resample.Kernel` of one overlap-save stage on the device (smx_resample_kernel_*; resample.mli:270-319, executor
`ols_run` resample.ml:1456-1599): the partition law -- the concatenation of every step plus flush equals `Stage.apply`
on the concatenated input BIT FOR BIT under any chunking --, burst emission, the drain, reset, the reference's error
behaviour, host and device-resident chunks.  The stage itself (the polyphase blocks) is checked against the oracle's
definition in test_gpu_parity.py::test_resample_stage_vs_oracle and against the reference's decibel ruler in
test_gpu_resample_quality.py."""
import numpy as np
import pytest

from oracle import soundml_oracle as O

pytestmark = pytest.mark.gpu


def _stage(l, m, k):
    from soundml_amd import Resample
    proto = Resample.prototype(l, k, 0.45 / max(l, m), O.kaiser_beta(100.0))
    return proto, Resample.Stage.create(proto, l, m, k)


def _run(kern, x, cuts, flush_device=None):
    parts, sizes = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        y = kern.step(x[..., a:b])
        sizes.append(0 if y is None else int(y.shape[-1]))
        if y is not None:
            parts.append(y)
    y = kern.flush(flush_device) if flush_device is not None else kern.flush()
    sizes.append(0 if y is None else int(y.shape[-1]))
    if y is not None:
        parts.append(y)
    return parts, sizes


CHUNKINGS = {
    "one": lambda n, rng: [0, n],
    "halves": lambda n, rng: [0, n // 2, n],
    "small": lambda n, rng: list(range(0, n, 97)) + [n],
    "random": lambda n, rng: [0] + sorted(set(int(v) for v in rng.integers(1, max(2, n), size=12))) + [n],
    "with_empty": lambda n, rng: [0, 0, n // 3, n // 3, n, n],
}


@pytest.mark.parametrize("l,m,k", [(2, 1, 160), (3, 1, 50), (4, 1, 37), (1, 2, 161), (1, 3, 100), (1, 4, 101), (2, 1, 8)])
@pytest.mark.parametrize("chunking", sorted(CHUNKINGS))
def test_partition_law(l, m, k, chunking):
    from soundml_amd import Resample
    rng = np.random.default_rng(100 * l + 10 * m + k)
    n = 23017
    x = rng.uniform(-1, 1, size=(3, n)).astype(np.float32)
    proto, st = _stage(l, m, k)
    whole = Resample.Stage.apply(st, x)
    cuts = CHUNKINGS[chunking](n, rng)
    kern = Resample.Kernel.prepare(st, channels=3, max_block=n)
    parts, sizes = _run(kern, x, cuts)
    got = np.concatenate(parts, axis=-1)
    assert got.shape == whole.shape == (3, -(-n * l // m))
    assert np.array_equal(got, whole), (chunking, sizes)
    assert kern.flush() is None                       # draining consumed the tail (resample.mli:313-317)
    # and against the definition (the tolerance of test_resample_stage_vs_oracle)
    want = O.resample_stage_direct(proto, l, m, k, x.astype(np.float64))
    assert np.max(np.abs(got.astype(np.float64) - want)) <= 1e-5 * np.sum(np.abs(proto))


def test_burst_emission_and_reset():
    """Steps inside a block pair return None, boundary crossings emit the whole run; reset restores the fresh state."""
    from soundml_amd import Resample
    rng = np.random.default_rng(5)
    l, m, k = 2, 1, 160
    _, st = _stage(l, m, k)
    x = rng.uniform(-1, 1, size=(1, 20000)).astype(np.float32)
    kern = Resample.Kernel.prepare(st, channels=1, max_block=4096)
    assert kern.step(x[:, :100]) is None               # no block completes: the samples only extend the carry
    y = kern.step(x[:, 100:4196])
    assert y is not None and y.shape[-1] % (2 * l) == 0
    cuts = [0, 100, 4196] + list(range(8196, 20001, 4000))
    end = cuts[-1]
    rest, sizes = _run(kern, x, cuts[2:])
    run1 = np.concatenate([y] + rest, axis=-1)
    assert np.array_equal(run1, Resample.Stage.apply(st, x[:, :end]))
    kern.reset()
    parts, _ = _run(kern, x, cuts)
    assert np.array_equal(np.concatenate(parts, axis=-1), run1)


def test_device_resident_chunks_equal_host_chunks():
    import torch
    from soundml_amd import Resample
    rng = np.random.default_rng(9)
    for l, m, k in ((2, 1, 160), (1, 3, 100)):
        _, st = _stage(l, m, k)
        n = 30011
        x = rng.uniform(-1, 1, size=(2, n)).astype(np.float32)
        cuts = [0, 5, 7000, 7001, 19000, n]
        host, _ = _run(Resample.Kernel.prepare(st, 2, n), x, cuts)
        dev, _ = _run(Resample.Kernel.prepare(st, 2, n), torch.from_numpy(x).cuda(), cuts, flush_device="cuda")
        assert all(p.is_cuda for p in dev)
        assert np.array_equal(np.concatenate(host, axis=-1), torch.cat(dev, dim=-1).cpu().numpy())
        assert np.array_equal(np.concatenate(host, axis=-1), Resample.Stage.apply(st, x))


def test_short_and_empty_streams():
    from soundml_amd import Resample
    for l, m, k, n in ((2, 1, 40, 1), (1, 2, 41, 1), (3, 1, 20, 17), (1, 4, 33, 5), (2, 1, 40, 0)):
        _, st = _stage(l, m, k)
        x = np.linspace(-1, 1, max(n, 1), dtype=np.float32)[None, :n]
        kern = Resample.Kernel.prepare(st, 1, 64)
        parts, _ = _run(kern, x, [0, n])
        whole = Resample.Stage.apply(st, x)
        if whole.shape[-1] == 0:
            assert parts == []
        else:
            assert np.array_equal(np.concatenate(parts, axis=-1), whole)


def test_errors_follow_the_reference():
    import soundml_amd as S
    from soundml_amd import Resample
    _, st = _stage(2, 1, 40)
    with pytest.raises(S.InvalidArgument):
        Resample.Kernel.prepare(st, channels=0, max_block=16)         # resample.mli:291-294
    with pytest.raises(S.InvalidArgument):
        Resample.Kernel.prepare(st, channels=1, max_block=0)
    proto = Resample.prototype(3, 20, 0.45 / 3, O.kaiser_beta(100.0))
    with pytest.raises(S.InvalidArgument):
        Resample.Kernel.prepare(Resample.Stage.create(proto, 3, 2, 20), 1, 16)   # not overlap-save eligible (resample.ml:279-300)
    kern = Resample.Kernel.prepare(st, channels=2, max_block=16)
    with pytest.raises(S.InvalidArgument):
        kern.step(np.zeros((2, 17), dtype=np.float32))                 # longer than max_block (resample.mli:309)
    with pytest.raises(S.InvalidArgument):
        kern.step(np.zeros((3, 4), dtype=np.float32))                  # leading axis disagrees with channels
    kern.step(np.zeros((2, 4), dtype=np.float32))
    kern.flush()
    with pytest.raises(S.InvalidArgument):
        kern.step(np.zeros((2, 4), dtype=np.float32))                  # drained by flush: reset first (resample.mli:310-311)
    kern.reset()
    assert kern.step(np.zeros((2, 4), dtype=np.float32)) is None

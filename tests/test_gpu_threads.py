"""Concurrency at the boundary (SURVEY 8(b) "Threading": a Config is shareable across OCaml domains, and the stubs release the
runtime lock around every call -- ocaml/soundml_amd_stubs.c; here ctypes releases the GIL the same way).  INTEGRATION section 4
states what the library promises: a handle's lazily built device tables and plans are built once under the handle's mutex, and
the staged host <-> device transfers serialise per direction, process-wide.  These tests drive exactly those two places from
several host threads at once and require the bits of the serial call."""
import threading

import numpy as np
import pytest

import soundml_amd as S
from soundml_amd import Mel, Stft

pytestmark = pytest.mark.gpu


def run_threads(n, fn):
    out, err = [None] * n, []
    gate = threading.Barrier(n)

    def work(i):
        try:
            gate.wait()
            out[i] = fn(i)
        except BaseException as e:   # noqa: BLE001 -- reported by the asserting thread
            err.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not err, err
    return out


@pytest.mark.parametrize("fft,hop", [(2048, 512), (1024, 256), (400, 160)])
def test_one_fresh_config_from_four_threads(fft, hop):
    """Four threads meet at a barrier and make the FIRST calls on a fresh Stft.Config and a fresh Mel.Config (nothing built yet:
    window / twiddle tables, the fused mel plan and its MFMA operand image are all created under contention), each on its own
    device-resident batch: power spectrogram, complex spectrogram, fused mel spectrogram, invert.  Every result must equal the
    one a single thread computes afterwards on the same handles, bit for bit."""
    import torch
    torch.manual_seed(fft)
    xs = [(torch.rand(3 + i, 30000 + 1000 * i, device="cuda") * 2 - 1).float() for i in range(4)]
    c = Stft.Config.create(fft_size=fft, hop=hop)
    m = Mel.Config.create(n_mels=40, sample_rate=16000, fft_size=fft)

    def call(i):
        p = Stft.power_spectrum(c, xs[i])
        z = Stft.transform(c, xs[i])
        mel = S.mel_spectrogram(c, m, xs[i])
        y = Stft.invert(c, z, length=xs[i].shape[-1])
        torch.cuda.synchronize()
        return p, z, mel, y

    got = run_threads(4, call)
    for i in range(4):
        want = call(i)
        for g, w, name in zip(got[i], want, ("power", "transform", "mel", "invert")):
            assert torch.equal(g, w), (name, i)


@pytest.mark.parametrize("pinned", [False, True])
def test_two_threads_on_the_host_pointer_path(pinned):
    """Two host threads call the host-pointer power spectrogram at once on batches large enough for the pipelined transfer
    (clip units whose upload, kernels and download overlap: transfer.cpp): the staging rings serialise the two calls per
    direction, and each result equals the device-resident call's.  pinned: the results come from the page-locked pool (the
    download then needs no ring), else they are ordinary arrays (both directions staged)."""
    import torch
    S.set_pinned_results(pinned)
    rng = np.random.default_rng(5)
    c = Stft.Config.create(fft_size=2048, hop=512)
    xs = [rng.uniform(-1, 1, size=(64, 200000 + 4096 * i)).astype(np.float32) for i in range(2)]   # ~51 MB in, ~103 MB out each: above the 128 MB threshold of the pipelined path
    got = run_threads(2, lambda i: Stft.power_spectrum(c, xs[i]))
    for i in range(2):
        want = Stft.power_spectrum(c, torch.from_numpy(xs[i]).cuda()).cpu().numpy()
        assert np.array_equal(got[i], want), i
    S.set_pinned_results(True)


@pytest.mark.parametrize("pinned", [False, True])
def test_pipelined_host_calls_equal_the_device_resident_ones(pinned):
    """The host-pointer transform / power spectrogram / invert of a batch above the pipelining threshold (units of clips whose
    upload, kernels and download overlap) against the device-resident entry points on the same data: bit for bit, including a
    batch whose last unit is ragged (clips not a multiple of the unit)."""
    import torch
    rng = np.random.default_rng(8)
    c = Stft.Config.create(fft_size=2048, hop=512)
    x = rng.uniform(-1, 1, size=(37, 300000)).astype(np.float32)      # 44 MB in, 89 MB / 178 MB out
    xd = torch.from_numpy(x).cuda()
    S.set_pinned_results(pinned)   # ordinary result arrays (both directions through the staging rings) / blocks of the page-locked pool (transform, power and invert alike)
    p = Stft.power_spectrum(c, x)
    assert np.array_equal(p, Stft.power_spectrum(c, xd).cpu().numpy())
    z = Stft.transform(c, x)
    zd = Stft.transform(c, xd)
    assert np.array_equal(z, zd.cpu().numpy())
    y = Stft.invert(c, z, length=x.shape[-1])
    assert np.array_equal(y, Stft.invert(c, zd, length=x.shape[-1]).cpu().numpy())
    assert np.max(np.abs(y - x)) < 2e-6
    S.set_pinned_results(True)

"""Result arrays of the host faces from the library's page-locked pool (smx_host_alloc / smx_host_free, include/soundml_amd.h):
the reference's faces return a fresh host tensor per call (stft.ml:356-364, 670-691); a large one here is a block the DMA engine
writes directly.  The values must not depend on where the result lives: page-locked and ordinary arrays agree bit for bit, the
pipelined call (clip units: transfer.cpp) and the serial one alike; the pool hands a released block to the next result; the two
entry points fail loudly on a pointer they do not own."""
import ctypes
import gc
import threading

import numpy as np
import pytest

import soundml_amd as S
from soundml_amd import Stft
from soundml_amd._lib import lib, SMX_OK, PINNED_RESULT_MIN_BYTES

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _pinned_on_again():
    yield
    S.set_pinned_results(True)


def _is_pinned_result(a):
    return not a.flags["OWNDATA"] and a.nbytes >= PINNED_RESULT_MIN_BYTES


@pytest.mark.parametrize("lead,n", [(37, 300000), (12, 400000)])   # above / below the 128 MB threshold of the pipelined call
def test_pinned_and_ordinary_results_agree(lead, n):
    import torch
    rng = np.random.default_rng(lead)
    x = rng.uniform(-1, 1, size=(lead, n)).astype(np.float32)
    c = Stft.Config.create(fft_size=2048, hop=512)
    got = {}
    for pinned in (True, False):
        S.set_pinned_results(pinned)
        p, z = Stft.power_spectrum(c, x), Stft.transform(c, x)
        assert _is_pinned_result(p) == pinned and _is_pinned_result(z) == pinned
        got[pinned] = (p, z)
    assert np.array_equal(got[True][0], got[False][0]) and np.array_equal(got[True][1], got[False][1])
    assert np.array_equal(got[True][0], Stft.power_spectrum(c, torch.from_numpy(x).cuda()).cpu().numpy())
    p = got[True][0]
    p[0, 0, 0] = -1.0                      # an ordinary writable array for the caller
    assert p[0, 0, 0] == -1.0 and p.flags["WRITEABLE"] and p.flags["C_CONTIGUOUS"]
    t = torch.from_numpy(p)                # ... and for torch
    assert t.shape == p.shape


def test_a_released_block_serves_the_next_result():
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, size=(16, 300000)).astype(np.float32)
    c = Stft.Config.create(fft_size=2048, hop=512)
    a = Stft.power_spectrum(c, x)
    assert _is_pinned_result(a)
    addr, keep = a.ctypes.data, a.copy()
    b = Stft.power_spectrum(c, x)          # `a` is alive: another block
    assert b.ctypes.data != addr
    del a
    gc.collect()
    d = Stft.power_spectrum(c, x)          # the released block comes back
    assert d.ctypes.data == addr and np.array_equal(d, keep) and np.array_equal(b, keep)
    view = d[3:5]                          # a view keeps the block alive after the array's name is gone
    del d
    gc.collect()
    e = Stft.power_spectrum(c, x)
    assert e.ctypes.data != addr and np.array_equal(view, keep[3:5])


def test_small_results_stay_ordinary_arrays():
    c = Stft.Config.create(fft_size=2048, hop=512)
    p = Stft.power_spectrum(c, np.zeros((2, 50000), dtype=np.float32))
    assert p.flags["OWNDATA"]


def test_host_alloc_and_free_at_the_c_abi():
    vp = ctypes.c_void_p
    p, q = vp(), vp()
    assert lib.smx_host_alloc(1 << 20, ctypes.byref(p)) == SMX_OK and p.value
    assert lib.smx_host_alloc(0, ctypes.byref(q)) == SMX_OK and q.value and q.value != p.value
    buf = (ctypes.c_ubyte * (1 << 20)).from_address(p.value)
    buf[0], buf[(1 << 20) - 1] = 7, 9      # usable host memory
    assert lib.smx_host_free(p) == SMX_OK and lib.smx_host_free(q) == SMX_OK
    assert lib.smx_host_free(p) != SMX_OK and b"smx_host_alloc" in lib.smx_last_error()       # released already
    own = np.zeros(16, dtype=np.uint8)
    assert lib.smx_host_free(vp(own.ctypes.data)) != SMX_OK                                    # somebody else's memory
    assert lib.smx_host_free(None) == SMX_OK                                                   # free(NULL)
    assert lib.smx_host_alloc(64, None) != SMX_OK


def test_a_caller_supplied_block_takes_the_direct_path_in_both_directions():
    """x AND out inside blocks of smx_host_alloc (what a binding that allocates its tensors there hands over): upload and download
    by the DMA engine alone; same bits as ordinary arrays."""
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, size=(40, 300000)).astype(np.float32)
    xp = S.pinned_empty(x.shape, np.float32)
    assert _is_pinned_result(xp)
    small = S.pinned_empty((3, 5), np.complex64)          # any size
    assert small.shape == (3, 5) and small.dtype == np.complex64 and not small.flags["OWNDATA"]
    xp[...] = x
    c = Stft.Config.create(fft_size=2048, hop=512)
    S.set_pinned_results(False)
    want = Stft.power_spectrum(c, x)
    S.set_pinned_results(True)
    assert np.array_equal(Stft.power_spectrum(c, xp), want)


def test_four_threads_with_pinned_results():
    rng = np.random.default_rng(4)
    c = Stft.Config.create(fft_size=1024, hop=256)
    xs = [rng.uniform(-1, 1, size=(48 + i, 200000)).astype(np.float32) for i in range(4)]
    S.set_pinned_results(False)
    want = [Stft.power_spectrum(c, x) for x in xs]
    S.set_pinned_results(True)
    out, err = [None] * 4, []
    gate = threading.Barrier(4)

    def work(i):
        try:
            gate.wait()
            for _ in range(2):
                out[i] = Stft.power_spectrum(c, xs[i])
        except BaseException as e:   # noqa: BLE001
            err.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not err, err
    for i in range(4):
        assert _is_pinned_result(out[i]) and np.array_equal(out[i], want[i]), i

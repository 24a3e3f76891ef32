"""Clip sharding under the C ABI (smx_set_devices; SURVEY 8(e), 7 step 7 "one host thread/stream per GPU").  The reference's
caller is ONE process handing over host tensors, a batch of clips in one call (stft.mli:211-250); with a device list the
host-pointer batch entry points cut `lead` into contiguous clip ranges (soundml_amd/shard.py clip_range) and run them side by
side, one host thread + one staging ring pair per listed device.  What must hold is the reference's per-slice law
(stft_grid.ml:180-205, mel_props.ml:136-155): the sharded call's result is the single-device call's, bit for bit.  A 1-GPU box
lists device 0 several times (virtual shards: same code path, two uploads of one device in flight); where a second device
is visible the same tests run across real devices."""
import ctypes

import numpy as np
import pytest

import soundml_amd as S
from soundml_amd import Mel, Stft, _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def single_device_after():
    yield
    S.set_devices([])
    S.set_pinned_results(True)


def device_lists():
    lists = [[0, 0], [0, 0, 0]]
    if S.device_count() >= 2:
        lists += [[0, 1], list(range(S.device_count()))]
    return lists


def staging_peak(reset=False):
    up, down = ctypes.c_int(), ctypes.c_int()
    _lib.check(_lib.lib.smx_debug_staging_peak(ctypes.byref(up), ctypes.byref(down), 1 if reset else 0))
    return up.value, down.value


def test_device_list_round_trips_and_validates():
    assert S.get_devices() == []
    S.set_devices([0, 0])
    assert S.get_devices() == [0, 0]
    S.set_devices(None)
    assert S.get_devices() == []
    with pytest.raises(S.Failure) as e:
        S.set_devices([0, S.device_count()])
    assert "is not one of the %d visible devices" % S.device_count() in str(e.value)
    assert S.get_devices() == []   # a rejected list changes nothing
    with pytest.raises(S.Failure):
        S.set_devices([-1])


@pytest.mark.parametrize("lead", [1, 2, 5, 7])
def test_small_batches_every_face_equals_the_single_device_call(lead):
    """Below the pipelining threshold (serial upload / kernels / download per shard), clips not divisible by the shard count,
    fewer clips than shards: transform, transform_range, power_spectrum (float32 and float64 audio), mel_spectrogram, invert."""
    rng = np.random.default_rng(lead)
    x = rng.uniform(-1, 1, size=(lead, 30000 + 37 * lead)).astype(np.float32)
    x64 = x[:, :9000].astype(np.float64)
    c = Stft.Config.create(fft_size=2048, hop=512)
    c4 = Stft.Config.create(fft_size=400, hop=160)
    m = Mel.Config.create(n_mels=64, sample_rate=48000, fft_size=2048)
    S.set_pinned_results(False)

    def faces():
        z = Stft.transform(c, x)
        return {"power": Stft.power_spectrum(c, x), "transform": z, "range": Stft.transform_range(c, x, 3, 41),
                "power_f64": Stft.power_spectrum(c4, x64), "mel": S.mel_spectrogram(c, m, x),
                "invert": Stft.invert(c, z, length=x.shape[-1]), "power_1.5": Stft.power_spectrum(c, x, power=1.5)}

    want = faces()
    for devices in device_lists():
        S.set_devices(devices)
        got = faces()
        S.set_devices([])
        for k in want:
            assert got[k].shape == want[k].shape and np.array_equal(got[k], want[k]), (devices, k)


@pytest.mark.parametrize("pinned", [False, True])
def test_pipelined_shards_equal_the_single_device_call_and_their_uploads_overlap(pinned):
    """Every shard above the pipelining threshold (clip units whose upload, kernels and download overlap, per shard): 67 clips
    over 2 and 3 shards (34 + 33; 23 + 22 + 22), ordinary and page-locked results.  The staging rings are per device and a device
    serves several transfers at once: with ordinary arrays at least two staged uploads were in flight at the same time."""
    rng = np.random.default_rng(17)
    x = rng.uniform(-1, 1, size=(67, 400000)).astype(np.float32)   # 107 MB in, 214 MB / 429 MB out
    c = Stft.Config.create(fft_size=2048, hop=512)
    S.set_pinned_results(pinned)
    want_p = Stft.power_spectrum(c, x)
    want_z = Stft.transform(c, x)
    want_y = Stft.invert(c, want_z, length=x.shape[-1])
    for devices in device_lists():
        S.set_devices(devices)
        Stft.power_spectrum(c, x)   # (first call with this list: every shard's device arrays come out of the driver, milliseconds apart)
        up = down = 0
        for _ in range(3):          # the shards start together (a rendezvous inside the call); their transfers take milliseconds
            staging_peak(reset=True)
            got_p = Stft.power_spectrum(c, x)
            u, d = staging_peak()
            up, down = max(up, u), max(down, d)
        got_z = Stft.transform(c, x)
        got_y = Stft.invert(c, got_z, length=x.shape[-1])
        S.set_devices([])
        assert np.array_equal(got_p, want_p), devices
        assert np.array_equal(got_z, want_z), devices
        assert np.array_equal(got_y, want_y), devices
        assert up >= 2, (devices, up, down)   # the shards' uploads were staged side by side (rounds 2-5: one ring per process, one at a time)
        if not pinned:
            assert down >= 2, (devices, up, down)


def test_two_threads_on_one_device_no_longer_take_turns():
    """transfer.cpp round 6: a device's staging is a pool of rings, so two host threads' staged uploads to ONE device overlap
    (test_gpu_threads.py checks their results; here: the peak)."""
    import threading
    rng = np.random.default_rng(3)
    c = Stft.Config.create(fft_size=2048, hop=512)
    xs = [rng.uniform(-1, 1, size=(64, 200000)).astype(np.float32) for _ in range(2)]
    S.set_pinned_results(False)
    Stft.power_spectrum(c, xs[0][:8])   # tables built
    out = [None, None]
    up = down = 0
    for attempt in range(4):   # (the first round also warms the device-side pool for two callers; the transfers take milliseconds and the threads start microseconds apart)
        staging_peak(reset=True)
        gate = threading.Barrier(2)

        def work(i):
            gate.wait()
            out[i] = Stft.power_spectrum(c, xs[i])

        ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in ts: t.start()
        for t in ts: t.join()
        u, d = staging_peak()
        up, down = max(up, u), max(down, d)
    assert up >= 2 and down >= 2, (up, down)
    import torch
    for i in range(2):
        assert np.array_equal(out[i], Stft.power_spectrum(c, torch.from_numpy(xs[i]).cuda()).cpu().numpy())


def test_errors_keep_their_kind_through_the_shards():
    """Precondition failures are raised before any shard starts, with the reference's messages and as the kind they are
    (Invalid_argument stays Invalid_argument, a null result stays a Failure)."""
    c = Stft.Config.create(fft_size=2048, hop=512)
    x = np.zeros((4, 8000), np.float32)
    S.set_devices([0, 0])
    with pytest.raises(S.InvalidArgument) as e:
        Stft.transform_range(c, x, 5, 2)
    assert str(e.value).startswith("transform_range: cannot take frames [5, 2)")
    m = Mel.Config.create(n_mels=40, sample_rate=16000, fft_size=1024)
    with pytest.raises(S.InvalidArgument):
        S.mel_spectrogram(c, m, x)
    rc = _lib.lib.smx_stft_power_spectrum_f32(c._h, x.ctypes.data_as(ctypes.c_void_p), 4, 8000, 2.0, None)
    assert rc == 2 and b"null pointer" in _lib.lib.smx_last_error()

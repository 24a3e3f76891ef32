"""Pins the CPU oracle (oracle/soundml_oracle.py) against every golden vector the
reference's own tests hold for the hot path, at the reference's own tolerances
(soundml/test/stft/stft_goldens.ml, soundml/test/mel/mel_goldens.ml,
soundml/test/window/test_window.ml)."""
import numpy as np
import pytest

from conftest import (F32_ATOL, F32_RTOL, F64_ATOL, F64_RTOL, check_close,
                      golden_cases, load_golden)
from oracle import soundml_oracle as O

# every synthesis vector file of the reference (soundml/test/istft/vectors; Griffin-Lim files apart)
ISTFT_FILES = ["inverse_fft2048_hop512", "inverse_fft64_hop16", "inverse_fft16_hop4", "inverse_fft2048_hop500_win1200", "inverse_fft31_hop5", "inverse_fft32_hop7", "inverse_fft32_hop8_win20", "inverse_fft64_hop17_win40", "lengths"]

STFT_FILES = ["fft16_hop4", "fft32_hop7", "fft64_hop16", "fft32_hop8_win20"]


def _stft_config(p):
    return O.stft_config(p["fft_size"], win_length=p["win_length"], hop=p["hop"],
                         alignment=p["alignment"])


@pytest.mark.parametrize("fname", STFT_FILES)
def test_stft_spectra(fname):
    for case in load_golden("stft", fname)["cases"]:
        p = case["params"]
        c = _stft_config(p)
        sig = O.lcg_signal(p["length"])
        power = {"magnitude": 1.0, "power": 2.0}[p["kind"]]
        if p["dtype"] == "float64":
            got = O.power_spectrum(c, sig, power)
            check_close(got, case["values"], case["shape"], F64_RTOL, F64_ATOL, case["name"])
        else:
            x = sig.astype(np.float32)
            got = O.power_spectrum(c, x, power)
            assert got.dtype == np.float32
            check_close(got, case["values"], case["shape"], F32_RTOL, F32_ATOL, case["name"])
            if power == 1.0:   # the complex64-witness leg of stft_goldens.ml:88-95
                z = O.transform(c, x, np.complex64)
                check_close(np.abs(z).astype(np.float32), case["values"], case["shape"],
                            F32_RTOL, F32_ATOL, case["name"] + "/complex64-witness")


def test_stft_sign_convention():
    for case in load_golden("stft", "complex_fft16_hop4")["cases"]:
        p = case["params"]
        z = O.transform(_stft_config(p), O.lcg_signal(p["length"]), np.complex128)
        part = z.real if p["kind"] == "real" else z.imag
        check_close(part, case["values"], case["shape"], F64_RTOL, F64_ATOL, case["name"])


def test_coordinates():
    for case in load_golden("stft", "coordinates")["cases"]:
        p = case["params"]
        if p["kind"] == "frequencies":
            c = O.stft_config(p["fft_size"], hop=p["hop"])
            got = O.frequencies(c, p["sample_rate"])
        else:
            c = O.stft_config(p["fft_size"], hop=p["hop"], alignment=p["alignment"])
            got = O.times(c, p["sample_rate"], p["length"])
            assert got.shape[0] == O.frames(c, p["length"])
        check_close(got, case["values"], case["shape"], msg=case["name"])
        check_close(got.astype(np.float32), case["values"], case["shape"], F32_RTOL, F32_ATOL,
                    case["name"] + "/float32")


WINDOW_FAMILIES = ["hann", "hamming", "blackman", "rectangular", "blackman_harris", "nuttall", "flat_top", "bartlett", "gaussian",
                   "kaiser", "tukey"]


def window_param(p):
    """The shape parameter of the parametric families as the vector files name it."""
    return p.get("beta", p.get("std", p.get("taper")))


@pytest.mark.parametrize("family", WINDOW_FAMILIES)
def test_windows(family):
    """All eleven families of Window.t against scipy's (test_window.ml:66-80)."""
    for case in load_golden("window", family)["cases"]:
        p = case["params"]
        got = O.window(p["window"], p["n"], periodic=p["periodic"], param=window_param(p))
        check_close(got, case["values"], case["shape"], msg=case["name"])


def _mel_config(p):
    return O.mel_config(p["n_mels"], p["sample_rate"], p["fft_size"], f_min=p["f_min"],
                        f_max=p["f_max"], scale=p["scale"], norm=p["norm"])


def test_mel_filterbank():
    for case in load_golden("mel", "filterbank")["cases"]:
        w = _mel_config(case["params"]).weights
        check_close(w, case["values"], case["shape"], F64_RTOL, F64_ATOL, case["name"])
        check_close(w.astype(np.float32), case["values"], case["shape"], F32_RTOL, F32_ATOL,
                    case["name"] + "/float32")


def test_mel_spectrogram():
    for case in load_golden("mel", "mel_spectrogram")["cases"]:
        p = case["params"]
        sc = O.stft_config(p["fft_size"], hop=p["hop"], alignment=p["alignment"])
        mc = _mel_config(p)
        sig = O.lcg_signal(p["length"], seed=20260803, envelope=p["envelope"])
        if p["dtype"] == "float64":
            got = O.mel_spectrogram(sc, mc, sig, p["power"])
            check_close(got, case["values"], case["shape"], F64_RTOL, F64_ATOL, case["name"])
        else:
            got = O.mel_spectrogram(sc, mc, sig.astype(np.float32), p["power"])
            assert got.dtype == np.float32
            check_close(got, case["values"], case["shape"], F32_RTOL, F32_ATOL, case["name"])


# --- structural laws of the reference restated on the oracle -------------------------------

def _mfcc_case(case, make_stft, make_mel, lcg):
    p = case["params"]
    sc = make_stft(p["fft_size"], hop=p["hop"], alignment=p["alignment"])
    mc = make_mel(p["n_mels"], p["sample_rate"], p["fft_size"], f_min=p["f_min"], f_max=p["f_max"],
                  scale=p["scale"], norm=p["norm"])
    x = lcg(p["length"], 20260803, envelope=p["envelope"])     # mel_goldens.ml:34-42
    if p["dtype"] == "float32":
        x = x.astype(np.float32)
    return sc, mc, x, p["n_mfcc"], (None if p["lifter"] == 0.0 else p["lifter"])


def test_mfcc_goldens():
    """Soundml.mfcc restated (oracle.mfcc) against librosa.feature.mfcc vectors (mel_goldens.ml:131-158):
    float64 rtol 1e-9 / atol 1e-9, float32 1e-4 / 1e-4 as the reference sets them."""
    for case in load_golden("mel", "mfcc")["cases"]:
        sc, mc, x, n_mfcc, lifter = _mfcc_case(case, lambda fft, **kw: O.stft_config(fft, **kw), O.mel_config, O.lcg_signal)
        got = O.mfcc(sc, mc, x, n_mfcc, lifter)
        f32 = case["params"]["dtype"] == "float32"
        check_close(got, case["values"], shape=case["shape"], rtol=1e-4 if f32 else 1e-9, atol=1e-4 if f32 else 1e-9,
                    msg=case["name"])


@pytest.mark.parametrize("vectors", ISTFT_FILES)
def test_istft_goldens(vectors):
    """Stft.invert restated (oracle.invert) against the reference's librosa-0.11 synthesis vectors
    (soundml/test/istft/vectors, replayed by istft_goldens.ml with these tolerances)."""
    from conftest import istft_golden_config, istft_golden_spectrum
    for case in load_golden("istft", vectors)["cases"]:
        p = case["params"]
        cfg = istft_golden_config(lambda fft, **kw: O.stft_config(fft, **kw), p)
        z = istft_golden_spectrum(p["fft_size"], p["frames"])
        f32 = p["dtype"] == "float32"
        got = O.invert(cfg, z.astype(np.complex64) if f32 else z, p.get("length"))
        assert got.dtype == (np.float32 if f32 else np.float64)
        check_close(got, case["values"], shape=case["shape"], rtol=F32_RTOL if f32 else F64_RTOL,
                    atol=F32_ATOL if f32 else F64_ATOL, msg=case["name"])


GL_FILES = ["griffinlim_fft64_hop16", "griffinlim_fft64_hop16_win40", "griffinlim_fft512_hop128"]


def gl_golden_magnitudes(fft_size, frames):
    """gl_goldens.ml:36-38: one 31-bit LCG stream shifted into [0, 2)."""
    bins = fft_size // 2 + 1
    return (O.lcg_signal(bins * frames, 20250803) + 1.0).reshape(bins, frames)


@pytest.mark.parametrize("vectors", GL_FILES)
def test_griffin_lim_goldens(vectors):
    """Stft.griffin_lim restated (oracle.griffin_lim) against librosa.griffinlim vectors at the deterministic
    settings (all-ones initial phase, zero padding; gl_goldens.ml), the reference's float64 tolerance."""
    from conftest import istft_golden_config
    for case in load_golden("istft", vectors)["cases"]:
        p = case["params"]
        cfg = istft_golden_config(lambda fft, **kw: O.stft_config(fft, **kw), p)
        mag = gl_golden_magnitudes(p["fft_size"], p["frames"])
        f32 = p["dtype"] == "float32"
        got = O.griffin_lim(cfg, mag.astype(np.float32) if f32 else mag, p["n_iter"], p["momentum"], None, p.get("length"))
        check_close(got, case["values"], shape=case["shape"], rtol=F32_RTOL if f32 else F64_RTOL,
                    atol=F32_ATOL if f32 else F64_ATOL, msg=case["name"])


def test_istft_oracle_laws():
    """istft_law.ml: the default length is the frame-count fixed point; a round trip restores the interior;
    the criterion rejects a hop the squared window cannot cover."""
    rng = np.random.default_rng(3)
    for fft, hop, align in ((64, 16, "centered"), (64, 16, "left"), (64, 16, "right"), (48, 12, "centered"), (31, 5, "centered")):
        cfg = O.stft_config(fft, hop=hop, alignment=align)
        x = rng.standard_normal(700)
        z = O.transform(cfg, x)
        assert O.frames(cfg, O.output_length(cfg, z.shape[-1])) == z.shape[-1]
        y = O.invert(cfg, z, length=x.size)
        lo, hi = fft, x.size - fft
        np.testing.assert_allclose(y[lo:hi], x[lo:hi], rtol=0, atol=1e-10)
    assert not O.nola(O.stft_config(64, hop=64))
    with pytest.raises(ValueError, match="cannot invert a 64-point window advanced by 64 samples"):
        O.invert(O.stft_config(64, hop=64), np.zeros((33, 4), np.complex128))
    with pytest.raises(ValueError, match="cannot invert 30 frequency bins"):
        O.invert(O.stft_config(64, hop=16), np.zeros((30, 4), np.complex128))


@pytest.mark.parametrize("alignment", ["centered", "left", "right"])
@pytest.mark.parametrize("pad", ["reflect", "edge", "constant"])
@pytest.mark.parametrize("fft,hop", [(16, 4), (32, 7), (16, 20)])
def test_partition_law(alignment, pad, fft, hop):
    """stft_law.ml:79-164: every chunking of the stream == the offline transform, exactly."""
    rng = np.random.default_rng(fft * 131 + hop)
    c = O.stft_config(fft, hop=hop, alignment=alignment, pad=pad, pad_value=0.25)
    for n in (1, 5, fft // 2, fft // 2 + 1, fft, 97, 200):
        x = rng.standard_normal((2, n))
        want = O.transform(c, x, np.complex128)
        for trial in range(3):
            k = O.StreamKernel(c, np.complex128)
            parts, pos = [], 0
            while pos < n:
                m = int(rng.integers(0, 9)) if trial else 1
                m = min(m, n - pos)
                out = k.step(x[:, pos:pos + m])
                pos += m
                if out is not None:
                    parts.append(out)
            out = k.flush()
            if out is not None:
                parts.append(out)
            got = np.concatenate(parts, axis=-1) if parts else np.zeros((2, c.bins, 0), complex)
            assert got.shape == want.shape, (n, got.shape, want.shape)
            assert np.array_equal(got, want)


def test_transform_range_tiles():
    """stft_grid.ml:32-73: adjacent frame ranges reassemble the full transform exactly."""
    c = O.stft_config(64, hop=16)
    x = O.lcg_signal(2000)
    full = O.transform(c, x)
    total = O.frames(c, 2000)
    cuts = [0, 1, 7, 8, 60, total]
    parts = [O.transform_range(c, x, a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate(parts, axis=-1), full)


def test_fir_fft_form_matches_direct():
    rng = np.random.default_rng(7)
    h = O.design_lowpass(63, 0.25, O.kaiser_beta(80.0))
    assert abs(h.sum() - 1.0) < 1e-12 and np.allclose(h, h[::-1], rtol=0, atol=1e-18)
    x = rng.standard_normal((3, 500))
    np.testing.assert_allclose(O.fir_filter(h, x), O.fir_filter_direct(h, x), rtol=0, atol=1e-12)


# --- the C restatement (oracle/oracle_stft.c) is pinned by the same goldens ---------------

@pytest.mark.parametrize("fname", STFT_FILES)
def test_c_oracle_stft_spectra(fname):
    from oracle import c_oracle
    for case in load_golden("stft", fname)["cases"]:
        p = case["params"]
        c = _stft_config(p)
        sig = O.lcg_signal(p["length"])
        power = {"magnitude": 1.0, "power": 2.0}[p["kind"]]
        if p["dtype"] == "float64":
            check_close(c_oracle.stft(c, sig, power), case["values"], case["shape"], F64_RTOL, F64_ATOL,
                        case["name"])
        else:
            check_close(c_oracle.stft(c, sig.astype(np.float32), power), case["values"], case["shape"],
                        F32_RTOL, F32_ATOL, case["name"])


def test_c_oracle_sign_and_numpy_agreement():
    from oracle import c_oracle
    for case in load_golden("stft", "complex_fft16_hop4")["cases"]:
        p = case["params"]
        z = c_oracle.stft(_stft_config(p), O.lcg_signal(p["length"]), complex_out=True)
        check_close(z.real if p["kind"] == "real" else z.imag, case["values"], case["shape"], F64_RTOL,
                    F64_ATOL, case["name"])
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, size=(3, 20000)).astype(np.float32)
    for pad in ("reflect", "edge", "constant"):
        c = O.stft_config(2048, hop=512, pad=pad, pad_value=0.5)
        a, b = c_oracle.stft(c, x, 2.0, threads=3), O.power_spectrum(c, x, 2.0)
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-6 * np.max(b))
    mc = O.mel_config(128, 48000, 2048)
    s = rng.uniform(0, 2, size=(2, 1025, 7)).astype(np.float32)
    np.testing.assert_allclose(c_oracle.mel_apply(mc, s), O.mel_apply(mc, s), rtol=1e-6, atol=1e-7)


# ---- spectral-shape features (soundml/test/spectral/spectral_goldens.ml) ----------------------------------------

SPECTRAL_FILES = ["spectral_centroid", "spectral_bandwidth", "spectral_rolloff", "spectral_flatness"]


def spectral_golden_input(p):
    """spectral_goldens.ml:27-40,84-132: |LCG| (optionally squared) reshaped to shape_s, or the magnitude STFT of
    the LCG signal (seed 20261024)."""
    f32 = p["dtype"] == "float32"
    if p["source"] == "spectrogram":
        shape = p["shape_s"]
        v = np.abs(O.lcg_signal(int(np.prod(shape)), 20261024))
        if p["squared"]:
            v = v * v
        s = v.reshape(shape)
        return s.astype(np.float32) if f32 else s
    x = O.lcg_signal(p["length"], 20261024)
    return O.power_spectrum(O.stft_config(p["fft_size"], hop=p["hop"]), x.astype(np.float32) if f32 else x, 1.0)


def spectral_golden_freqs(p, s):
    if p["freqs"] == "fft":
        return None
    bins = s.shape[-2]                       # spectral_goldens.ml:51-53: 10 (k+1)(k+2)/2, exact integers
    return np.array([10.0 * float((k + 1) * (k + 2)) / 2.0 for k in range(bins)], dtype=s.dtype)


def run_spectral(mod, stem, p, s, freqs):
    if stem == "spectral_centroid":
        return mod.spectral_centroid(s, sample_rate=p["sample_rate"], freqs=freqs)
    if stem == "spectral_bandwidth":
        return mod.spectral_bandwidth(s, sample_rate=p["sample_rate"], p=p["p"], freqs=freqs)
    if stem == "spectral_rolloff":
        return mod.spectral_rolloff(s, sample_rate=p["sample_rate"], roll_percent=p["roll_percent"], freqs=freqs)
    return mod.spectral_flatness(s, amin=p["amin"], power=p["power"])


@pytest.mark.parametrize("stem", SPECTRAL_FILES)
def test_spectral_goldens(stem):
    """Spectral.{centroid,bandwidth,rolloff,flatness} restated against the librosa-0.11 vectors at the reference's
    tolerances (spectral_goldens.ml:21-23: float64 1e-9 / 1e-12; float32 the shared 1e-6 / 1e-7)."""
    for case in load_golden("spectral", stem)["cases"]:
        p = case["params"]
        s = spectral_golden_input(p)
        got = run_spectral(O, stem, p, s, spectral_golden_freqs(p, s))
        f32 = p["dtype"] == "float32"
        assert got.dtype == (np.float32 if f32 else np.float64)
        check_close(got, case["values"], shape=case["shape"], rtol=F32_RTOL if f32 else F64_RTOL,
                    atol=F32_ATOL if f32 else F64_ATOL, msg=case["name"])


# ---- chroma over the linear-frequency spectrum (soundml/test/chroma/chroma_goldens.ml) ---------------------------

def chroma_golden_config(make, p):
    return make(p["sample_rate"], p["fft_size"], n_chroma=p["n_chroma"], tuning=p["tuning"],
                ctroct=p.get("ctroct", 5.0), octwidth=p.get("octwidth", 2.0), base_c=p.get("base_c", True))


CHROMA_NORMS = {"inf": "inf", "l1": 1.0, "l2": 2.0, "none": None}


def test_chroma_filterbank_goldens():
    """Chroma.Config weights against librosa.filters.chroma (chroma_goldens.ml:111-136): closed form, 1e-12
    relative + 1e-13 of the peak."""
    for case in load_golden("chroma", "chroma_fb")["cases"]:
        p = case["params"]
        if p["kind"] != "filterbank":
            continue                                  # constant-Q projections: out of scope (no CQT here)
        c = chroma_golden_config(O.chroma_config, p)
        peak = float(np.max(np.abs(case["values"])))
        check_close(c.weights, case["values"], shape=case["shape"], rtol=1e-12, atol=1e-13 * peak, msg=case["name"])
        check_close(c.weights.astype(np.float32), case["values"], shape=case["shape"], rtol=F32_RTOL, atol=F32_ATOL,
                    msg=case["name"] + "/float32")


def test_chroma_stft_goldens():
    """Soundml.chroma_stft against librosa.feature.chroma_stft on the harmonic test signal
    (chroma_goldens.ml:165-196: zero padding, closed-form tolerances; float32 input is the quantised signal)."""
    for case in load_golden("chroma", "chroma_stft")["cases"]:
        p = case["params"]
        sc = O.stft_config(p["fft_size"], hop=p["hop"], pad="constant", pad_value=0.0)
        cc = chroma_golden_config(O.chroma_config, p)
        x = O.harmonic_signal(p["length"], p["sample_rate"])
        f32 = p["dtype"] == "float32"
        got = O.chroma_stft(sc, cc, x.astype(np.float32) if f32 else x, power=p["power"], norm=CHROMA_NORMS[p["norm"]])
        peak = float(np.max(np.abs(case["values"])))
        if f32:
            assert got.dtype == np.float32
            check_close(got, case["values"], shape=case["shape"], rtol=F32_RTOL, atol=F32_ATOL, msg=case["name"])
        else:
            check_close(got, case["values"], shape=case["shape"], rtol=1e-12, atol=1e-13 * peak, msg=case["name"])


# ---- decibel conversions (soundml/test/db/test_golden.ml) ---------------------------------------------------------

def db_golden_cases(which):
    for case in load_golden("db", which)["cases"]:
        p = case["params"]
        dt = np.float32 if p["dtype"] == "float32" else np.float64
        yield case, np.asarray(p["input"], dtype=dt).reshape(case["shape"]), p, dt


@pytest.mark.parametrize("which", ["power_to_db", "amplitude_to_db"])
def test_db_goldens(which):
    """Convert.power_to_db / amplitude_to_db restated against librosa's (test_golden.ml:39-57: float32 1e-4,
    float64 1e-10); the inputs ride in the vector files."""
    fn = getattr(O, which)
    for case, x, p, dt in db_golden_cases(which):
        got = fn(x, reference=p["reference"], amin=p["amin"], top_db=p["top_db"])
        assert got.dtype == dt
        tol = 1e-4 if dt == np.float32 else 1e-10
        check_close(got, case["values"], shape=case["shape"], rtol=tol, atol=tol, msg=case["name"])


def test_cola_goldens():
    """Window.cola against scipy.signal.check_COLA (test_window.ml)."""
    for case in load_golden("window", "cola")["cases"]:
        p = case["params"]
        assert O.cola(p["window"], p["length"], p["hop"], window_param(p)) == case["expected"], case["name"]

"""GPU parity tests proper (`-m gpu`): the HIP path, called through the C ABI,
against (i) the reference's committed golden vectors, (ii) the CPU oracle on
seeded inputs, (iii) the reference's structural laws (range tiling, leading-axis
broadcast, streaming partition law) and (iv) size-independent properties at the
BASELINE sizes.  Nothing here reads /root/reference.

Tolerances
  * float64 audio / float64 interior: the reference's own (rtol 1e-9, atol 1e-12;
    float32 goldens rtol 1e-6, atol 1e-7 -- stft_goldens.ml:13-17, tutils.ml:80-86).
  * float32 interior (the fast path): north_star's 1e-5 relative, evaluated as
    |a - e| <= 1e-5 * max|e| + 1e-5 * |e| per signal (BASELINE.md "Parity gate").
"""
import os

import numpy as np
import pytest

from conftest import (F32_ATOL, F32_RTOL, F64_ATOL, F64_RTOL, ROOT, check_close, load_golden)
from oracle import soundml_oracle as O

# every synthesis vector file of the reference (soundml/test/istft/vectors; Griffin-Lim files apart)
ISTFT_FILES = ["inverse_fft2048_hop512", "inverse_fft64_hop16", "inverse_fft16_hop4", "inverse_fft2048_hop500_win1200", "inverse_fft31_hop5", "inverse_fft32_hop7", "inverse_fft32_hop8_win20", "inverse_fft64_hop17_win40", "lengths"]

import soundml_amd as S
from soundml_amd import Fir, Mel, Stft

pytestmark = pytest.mark.gpu

FAST_RTOL = 1e-5


def check_fast(actual, expected, msg=""):
    """north_star tolerance: rtol 1e-5 plus atol 1e-5 * peak of the expected signal."""
    a = np.asarray(actual, dtype=np.float64)
    e = np.asarray(expected, dtype=np.float64)
    assert a.shape == e.shape, (msg, a.shape, e.shape)
    peak = np.max(np.abs(e)) if e.size else 0.0
    tol = FAST_RTOL * peak + FAST_RTOL * np.abs(e)
    bad = np.abs(a - e) > tol
    assert not bad.any(), "%s: %d/%d outside 1e-5 (max err %.3g, peak %.3g)" % (
        msg, int(bad.sum()), a.size, float(np.max(np.abs(a - e))), peak)


@pytest.fixture(autouse=True)
def _default_interior():
    S.set_interior("float32")
    yield
    S.set_interior("float32")


def _cfg(p, **kw):
    return Stft.Config.create(fft_size=p["fft_size"], win_length=p.get("win_length"), hop=p["hop"],
                              alignment=p["alignment"], **kw)


STFT_FILES = ["fft16_hop4", "fft32_hop7", "fft64_hop16", "fft32_hop8_win20"]


@pytest.mark.parametrize("fname", STFT_FILES)
def test_stft_goldens_float64(fname):
    """float64 audio: the reference's float64 interior, at the reference's tolerance."""
    for case in load_golden("stft", fname)["cases"]:
        p = case["params"]
        if p["dtype"] != "float64":
            continue
        power = {"magnitude": 1.0, "power": 2.0}[p["kind"]]
        got = Stft.power_spectrum(_cfg(p), O.lcg_signal(p["length"]), power)
        assert got.dtype == np.float64
        check_close(got, case["values"], case["shape"], F64_RTOL, F64_ATOL, case["name"])


@pytest.mark.parametrize("fname", STFT_FILES)
def test_stft_goldens_float32_strict_interior(fname):
    """float32 audio with the float64 interior: the reference's float32 tolerance."""
    S.set_interior("float64")
    for case in load_golden("stft", fname)["cases"]:
        p = case["params"]
        if p["dtype"] != "float32":
            continue
        power = {"magnitude": 1.0, "power": 2.0}[p["kind"]]
        x = O.lcg_signal(p["length"]).astype(np.float32)
        got = Stft.power_spectrum(_cfg(p), x, power)
        assert got.dtype == np.float32
        check_close(got, case["values"], case["shape"], F32_RTOL, F32_ATOL, case["name"])
        if power == 1.0:   # complex64 witness leg (stft_goldens.ml:88-95)
            z = Stft.transform(_cfg(p), x)
            assert z.dtype == np.complex64
            check_close(np.abs(z).astype(np.float32), case["values"], case["shape"], F32_RTOL, F32_ATOL,
                        case["name"] + "/complex64-witness")


@pytest.mark.parametrize("fname", STFT_FILES)
def test_stft_goldens_float32_fast_interior(fname):
    """float32 audio, float32 interior (documented deviation): north_star's 1e-5."""
    for case in load_golden("stft", fname)["cases"]:
        p = case["params"]
        if p["dtype"] != "float32":
            continue
        power = {"magnitude": 1.0, "power": 2.0}[p["kind"]]
        got = Stft.power_spectrum(_cfg(p), O.lcg_signal(p["length"]).astype(np.float32), power)
        check_fast(got, np.asarray(case["values"]).reshape(case["shape"]), case["name"])


def test_sign_convention():
    for case in load_golden("stft", "complex_fft16_hop4")["cases"]:
        p = case["params"]
        z = Stft.transform(_cfg(p), O.lcg_signal(p["length"]))
        assert z.dtype == np.complex128
        part = z.real if p["kind"] == "real" else z.imag
        check_close(part, case["values"], case["shape"], F64_RTOL, F64_ATOL, case["name"])


def _mel_cfg(p):
    return Mel.Config.create(n_mels=p["n_mels"], sample_rate=p["sample_rate"], fft_size=p["fft_size"],
                             f_min=p["f_min"], f_max=p["f_max"], scale=p["scale"], norm=p["norm"])


def test_mel_spectrogram_goldens():
    for case in load_golden("mel", "mel_spectrogram")["cases"]:
        p = case["params"]
        sc = Stft.Config.create(fft_size=p["fft_size"], hop=p["hop"], alignment=p["alignment"])
        mc = _mel_cfg(p)
        sig = O.lcg_signal(p["length"], seed=20260803, envelope=p["envelope"])
        want = np.asarray(case["values"]).reshape(case["shape"])
        if p["dtype"] == "float64":
            got = S.mel_spectrogram(sc, mc, sig, p["power"])
            check_close(got, want, case["shape"], F64_RTOL, F64_ATOL, case["name"])
        else:
            x = sig.astype(np.float32)
            check_fast(S.mel_spectrogram(sc, mc, x, p["power"]), want, case["name"] + "/fast")
            S.set_interior("float64")
            check_close(S.mel_spectrogram(sc, mc, x, p["power"]), want, case["shape"], F32_RTOL, F32_ATOL,
                        case["name"] + "/strict")
            S.set_interior("float32")


# ---- against the oracle at the benchmark geometries -------------------------------------------

@pytest.mark.parametrize("fft,hop,n,lead", [
    (2048, 512, 480000, 3),     # C2 geometry, 3 clips
    (2048, 512, 5000, 5),       # short clips: every frame touches a border
    (2048, 512, 16 * 512 * 3 + 17, 2),
    (1024, 256, 44100, 2),      # C1 geometry (1 s)
    (2048, 500, 30000, 2),      # hop not a multiple of 4 (8-byte aligned loads still hold)
    (2048, 511, 30000, 2),      # odd hop: unaligned load variant
    (512, 128, 20000, 2),
    (4096, 1024, 50000, 1),
])
@pytest.mark.parametrize("power", [2.0, 1.0])
def test_power_spectrum_vs_oracle(fft, hop, n, lead, power):
    rng = np.random.default_rng(fft + hop + n)
    x = rng.uniform(-1, 1, size=(lead, n)).astype(np.float32)
    c = Stft.Config.create(fft_size=fft, hop=hop)
    got = Stft.power_spectrum(c, x, power)
    want = O.power_spectrum(O.stft_config(fft, hop=hop), x, power)
    assert got.shape == want.shape and got.dtype == np.float32
    for i in range(lead):
        check_fast(got[i], want[i], "clip %d" % i)


@pytest.mark.parametrize("power", [0.5, 0.67, 1.0, 3.0, 0.0, -0.5])
def test_fused_general_power_vs_oracle(power):
    """`magnitude_pow` (stft.ml:670-674) at fft 2048 on the fused kernels for exponents other than 2: the magnitude is one
    v_sqrt_f32, a general power the split-exponent form of power_from_square (stft_fast.hip) -- each value within a few
    ulp of the float64 power (1e-6 relative, bin by bin: tighter than the north-star tolerance, which is relative to the
    peak), borders, ragged tile and the unaligned variant included; a silent clip gives 0^p as powf does (0, 1 or inf);
    slices equal the batch bit for bit in every mode."""
    rng = np.random.default_rng(int(power * 100) + 107)
    x = rng.uniform(-1, 1, size=(3, 16 * 512 * 2 + 333)).astype(np.float32)
    x[1] *= 1e-3
    x[2] = 0.0
    for hop in (512, 511):
        c = Stft.Config.create(fft_size=2048, hop=hop)
        got = Stft.power_spectrum(c, x, power)
        want = O.power_spectrum(O.stft_config(2048, hop=hop), x, power)
        for i in range(2):
            if power >= 0:   # (a negative exponent turns the smallest bins, the least accurate ones, into the peak)
                check_fast(got[i], want[i], "power %g clip %d" % (power, i))
            # bin by bin, where the float32 interior's own error (4e-7 of the peak in amplitude) is small beside the value
            big = np.abs(want[i]) ** (1.0 / power if power else 1.0) > 1e-2 * np.max(np.abs(want[i]) ** (1.0 / power if power else 1.0))
            rel = np.abs(got[i].astype(np.float64) - want[i])[big] / np.abs(want[i])[big]
            assert rel.max() < 1e-4 * max(1.0, abs(power)), (power, i, rel.max())
        silent = got[2]
        assert np.all(silent == (0.0 if power > 0 else 1.0 if power == 0 else np.inf)), (power, silent.min(), silent.max())
        assert np.array_equal(got[1], Stft.power_spectrum(c, x[1], power))
        a, b = 3, got.shape[-1] - 2
        assert np.array_equal(Stft.power_range(c, x, a, b, power), got[..., a:b])


@pytest.mark.parametrize("fft,hop", [(2048, 512), (1024, 256), (512, 128), (256, 64)])
@pytest.mark.parametrize("power", [0.5, 1.0, 2.0, 3.0, -0.5])
def test_nan_sample_propagates_through_every_power(fft, hop, power):
    """One NaN sample makes every bin of every frame that covers it NaN, whatever the exponent (the reference's float64
    arithmetic and `Float.pow` propagate it: stft.ml:670-674); the frames that do not cover it stay finite.  The general
    power of the register pipelines clamps its logarithm with v_med3_f32, which drops a NaN operand -- it must be put back."""
    rng = np.random.default_rng(fft + int(10 * power))
    n = 40 * hop + 17
    x = rng.uniform(-1, 1, size=(2, n)).astype(np.float32)
    pos = 17 * hop + 5
    x[1, pos] = np.nan
    c = Stft.Config.create(fft_size=fft, hop=hop)
    got = Stft.power_spectrum(c, x, power)
    assert np.isfinite(got[0]).all()
    frames = got.shape[-1]
    starts = np.arange(frames) * hop - fft // 2       # centered frames (librosa convention of the default configuration)
    covered = (starts <= pos) & (pos < starts + fft)
    assert covered.any() and not covered.all()
    assert np.isnan(got[1][:, covered]).all(), (fft, power)
    assert np.isfinite(got[1][:, ~covered]).all(), (fft, power)


@pytest.mark.parametrize("hop,n,lead", [(512, 480000, 2), (512, 5000, 3), (512, 16 * 512 * 3 + 17, 2), (500, 30000, 2),
                                        (511, 30000, 2)])
def test_transform_float32_vs_oracle(hop, n, lead):
    """Stft.transform on float32 audio at fft 2048 (the fused complex-spectrum kernel): real and imaginary
    parts against the float64 oracle, every frame including the reflected borders and the ragged last tile."""
    rng = np.random.default_rng(hop + n)
    x = rng.uniform(-1, 1, size=(lead, n)).astype(np.float32)
    c = Stft.Config.create(fft_size=2048, hop=hop)
    got = Stft.transform(c, x)
    want = O.transform(O.stft_config(2048, hop=hop), x)
    assert got.shape == want.shape and got.dtype == np.complex64
    for i in range(lead):
        check_fast(got[i].real, want[i].real, "re clip %d" % i)
        check_fast(got[i].imag, want[i].imag, "im clip %d" % i)
    # and it is the spectrum whose squared magnitude power_spectrum returns
    pw = Stft.power_spectrum(c, x)
    check_fast(pw, np.abs(got.astype(np.complex128)) ** 2, "power vs |transform|^2")


@pytest.mark.parametrize("fft,hop", [(256, 64), (512, 128), (1024, 256), (4096, 1024), (8192, 2048), (16384, 4096), (1024, 300),
                                     (400, 160), (441, 220), (100, 33), (1200, 300), (3000, 750), (6000, 1500), (8191, 2047)])
def test_other_sizes_float32(fft, hop):
    """Float32 audio at sizes other than 2048: powers of two 256 .. 16384 on the Stockham-pass kernel, everything
    else up to 8192 by chirp-z (Bluestein) on the same passes (stft_generic.hip): spectrum and power against the oracle, reflected borders and ragged last tile included, and
    adjacent frame ranges reassemble the whole bit for bit (stft_grid.ml:32-73)."""
    rng = np.random.default_rng(fft + hop)
    n = 7 * fft + 333
    x = rng.uniform(-1, 1, size=(3, n)).astype(np.float32)
    c = Stft.Config.create(fft_size=fft, hop=hop)
    o = O.stft_config(fft, hop=hop)
    z, want = Stft.transform(c, x), O.transform(o, x)
    assert z.shape == want.shape and z.dtype == np.complex64
    check_fast(z.real, want.real, "re")
    check_fast(z.imag, want.imag, "im")
    check_fast(Stft.power_spectrum(c, x), O.power_spectrum(o, x), "power")
    total = Stft.frames(c, n)
    cuts = [0, 1, 5, total // 2, total - 1, total]
    parts = [Stft.transform_range(c, x, a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate(parts, axis=-1), z)


@pytest.mark.parametrize("fft", [12, 20, 24, 30, 36, 40, 48, 60, 100, 120, 160, 200, 240, 320, 400, 480, 600, 640, 800, 960, 1000, 1200, 1920, 2000])
def test_mixed_radix_sizes(fft):
    """Even sizes whose half length is 2^a 3^b 5^c (not a power of two, <= 512): the power spectrogram and the fused mel
    spectrogram come from the direct mixed-radix kernel (stft_mixed_power16_kernel: radix 4 / 2 / 5 / 3 passes), every
    radix combination up to fft 1000; against the oracle, with a regression gate (the kernel's measured error is
    2-4e-7 of the peak), ragged last tile, reflected borders, a general power, and the chirp-z kernel it replaces as a
    second opinion (SMX_MIXED_OFF is read once per process, so that comparison runs in a child process)."""
    rng = np.random.default_rng(fft)
    hop = max(1, fft // 3)
    n = 37 * hop + fft // 2 + 5
    x = rng.uniform(-1, 1, size=(3, n)).astype(np.float32)
    c, o = Stft.Config.create(fft_size=fft, hop=hop), O.stft_config(fft, hop=hop)
    for power in (2.0, 1.0):
        got, want = Stft.power_spectrum(c, x, power), O.power_spectrum(o, x, power)
        check_fast(got, want, "power %g" % power)
        assert np.max(np.abs(got - want)) <= 2e-6 * np.max(np.abs(want)), "regression gate"
    total = Stft.frames(c, n)
    a, b = total // 3, total - 2
    assert np.array_equal(Stft.power_range(c, x, a, b), Stft.power_spectrum(c, x)[..., a:b])
    if fft >= 64:
        try:
            mc = Mel.Config.create(n_mels=20, sample_rate=16000, fft_size=fft)
        except S.InvalidArgument:
            mc = None
        if mc is not None:
            check_fast(S.mel_spectrogram(c, mc, x), O.mel_spectrogram(o, O.mel_config(20, 16000, fft), x), "mel")


@pytest.mark.parametrize("fft", [4, 8, 12, 16, 60, 64, 100, 128, 240, 256, 400, 480, 800, 960, 1000, 1200, 2000])
def test_mixed_radix_invert(fft):
    """Stft.invert at the same sizes: the frames come from the mixed-radix inverse kernel (istft_mixed_frames_kernel) instead
    of the O(N^2) direct inverse DFT (and, for the powers of two up to 256, instead of the radix-2 kernel); against the oracle on the spectrum of a seeded signal (so the round trip is checked
    too), complex64 spectra, float32 interior."""
    rng = np.random.default_rng(fft + 1)
    hop = fft // 4
    n = 23 * hop + 7
    x = rng.uniform(-1, 1, size=(2, n)).astype(np.float32)
    c, o = Stft.Config.create(fft_size=fft, hop=hop), O.stft_config(fft, hop=hop)
    z = O.transform(o, x).astype(np.complex64)
    got, want = Stft.invert(c, z, n), O.invert(o, z, n)
    assert got.shape == want.shape == x.shape and got.dtype == np.float32
    check_fast(got, want, "invert")
    check_fast(got, x, "round trip")


@pytest.mark.parametrize("fft", [12, 100, 400, 480, 960, 1000, 1200, 2000])
def test_mixed_radix_float64_interior(fft):
    """The same sizes with the reference's float64 interior (stft_mixed_power16_kernel / istft_mixed_frames_kernel on doubles):
    float64 audio at the reference's float64 tolerance (complex128 / float64 out), float32 audio at its float32 tolerance
    (complex64 / float32 out, rounded once); transform, power and invert."""
    rng = np.random.default_rng(fft + 2)
    hop = fft // 4
    n = 19 * hop + 3
    o = O.stft_config(fft, hop=hop)
    c = Stft.Config.create(fft_size=fft, hop=hop)
    x64 = rng.uniform(-1, 1, size=(2, n))
    z = Stft.transform(c, x64)
    assert z.dtype == np.complex128
    want = O.transform(o, x64)
    check_close(z.real, want.real, rtol=F64_RTOL, atol=1e-11, msg="re")
    check_close(z.imag, want.imag, rtol=F64_RTOL, atol=1e-11, msg="im")
    check_close(Stft.power_spectrum(c, x64), O.power_spectrum(o, x64), rtol=4 * F64_RTOL, atol=1e-10, msg="power")
    check_close(Stft.invert(c, z, n), O.invert(o, want, n), rtol=1e-8, atol=1e-11, msg="invert")
    x32 = x64.astype(np.float32)
    S.set_interior("float64")
    z32, w32 = Stft.transform(c, x32), O.transform(o, x32)
    assert z32.dtype == np.complex64
    peak = float(np.max(np.abs(w32)))
    assert np.max(np.abs(z32 - w32)) <= 2e-7 * peak                      # one rounding of a float64 result
    p32, wp = Stft.power_spectrum(c, x32), O.power_spectrum(o, x32)
    assert np.max(np.abs(p32 - wp)) <= 4e-7 * float(np.max(wp))
    xi, wi = Stft.invert(c, z32, n), O.invert(o, z32, n)
    assert xi.dtype == np.float32 and np.max(np.abs(xi - wi)) <= 4e-7 * float(np.max(np.abs(wi)))


@pytest.mark.parametrize("fft", [9, 15, 21, 45, 49, 105, 225, 315, 441, 675, 945])
def test_mixed_radix_odd_sizes(fft):
    """Odd sizes 3^b 5^c 7^d: the same kernels on the frame as a complex signal of N points (radix 5 / 3 / 7 passes, no half-size
    trick), forward (power, complex) and inverse, float32 and the float64 interior, against the oracle."""
    rng = np.random.default_rng(fft + 3)
    hop = max(1, fft // 4)
    n = 21 * hop + fft + 2
    x = rng.uniform(-1, 1, size=(2, n)).astype(np.float32)
    c, o = Stft.Config.create(fft_size=fft, hop=hop), O.stft_config(fft, hop=hop)
    z, want = Stft.transform(c, x), O.transform(o, x)
    check_fast(z.real, want.real, "re")
    check_fast(z.imag, want.imag, "im")
    got, wp = Stft.power_spectrum(c, x), O.power_spectrum(o, x)
    check_fast(got, wp, "power")
    assert np.max(np.abs(got - wp)) <= 2e-6 * np.max(np.abs(wp)), "regression gate"
    if Stft.nola(c):
        zz = want.astype(np.complex64)
        check_fast(Stft.invert(c, zz, n), O.invert(o, zz, n), "invert")
    x64 = x.astype(np.float64)
    z64, w64 = Stft.transform(c, x64), O.transform(o, x64)
    assert z64.dtype == np.complex128
    check_close(z64.real, w64.real, rtol=F64_RTOL, atol=1e-11, msg="re f64")
    check_close(z64.imag, w64.imag, rtol=F64_RTOL, atol=1e-11, msg="im f64")
    if Stft.nola(c):
        check_close(Stft.invert(c, z64, n), O.invert(o, w64, n), rtol=1e-8, atol=1e-11, msg="invert f64")


def test_mixed_radix_kernel_agrees_with_chirp_z():
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
from soundml_amd import Stft
rng = np.random.default_rng(5)
out = {}
for fft in (100, 400, 960):
    x = rng.uniform(-1, 1, size=(2, 9 * fft + 17)).astype(np.float32)
    out[str(fft)] = Stft.power_spectrum(Stft.Config.create(fft_size=fft, hop=fft // 4), x)
np.savez(sys.argv[1], **out)
""" % ROOT
    import subprocess, sys, tempfile
    res = []
    for off in ("0", "1"):
        path = tempfile.mktemp(suffix=".npz")
        subprocess.run([sys.executable, "-c", code, path], check=True, env=dict(os.environ, SMX_MIXED_OFF=off), timeout=300)
        res.append(dict(np.load(path)))
        os.remove(path)
    for k in res[0]:
        peak = float(np.max(res[0][k]))
        assert np.max(np.abs(res[0][k] - res[1][k])) <= 2e-6 * peak, k
        assert not np.array_equal(res[0][k], res[1][k]), "both runs took the same kernel"


@pytest.mark.parametrize("fft,hop,alignment,pad", [(512, 128, "centered", "reflect"), (1024, 256, "left", "edge"),
                                                   (2048, 512, "centered", "reflect"), (2048, 500, "right", ("constant", 0.5)),
                                                   (4096, 1024, "centered", "reflect"), (1024, 1500, "centered", "edge")])
def test_float64_interior_stockham_sizes(fft, hop, alignment, pad):
    """The float64 interior on the Stockham passes (fft 512 .. 4096, stft_stockham_real_kernel<.., double, ..>):
    float64 audio at the reference's float64 tolerance, float32 audio with `set_interior("float64")` at its float32
    one (stft_goldens.ml:13-17), spectrum and power, borders and ragged tiles, a scaled window; frame ranges
    reassemble the whole bit for bit."""
    rng = np.random.default_rng(fft + hop)
    n = 9 * fft + 123
    x64 = rng.standard_normal((2, n))
    kw = dict(hop=hop, alignment=alignment, scale="magnitude")
    c = Stft.Config.create(fft_size=fft, pad=pad, **kw)
    o = (O.stft_config(fft, pad=pad[0], pad_value=pad[1], **kw) if isinstance(pad, tuple) else O.stft_config(fft, pad=pad, **kw))
    z, want = Stft.transform(c, x64), O.transform(o, x64)
    assert z.dtype == np.complex128 and z.shape == want.shape
    scale = float(np.max(np.abs(want)))
    np.testing.assert_allclose(z, want, rtol=F64_RTOL, atol=1e-12 * scale)
    np.testing.assert_allclose(Stft.power_spectrum(c, x64, power=1.0), O.power_spectrum(o, x64, 1.0), rtol=F64_RTOL, atol=1e-12 * scale)
    x32 = x64.astype(np.float32)
    S.set_interior("float64")
    try:
        z32, p32 = Stft.transform(c, x32), Stft.power_spectrum(c, x32)
        total = Stft.frames(c, n)
        cuts = [0, 1, 3, total // 2, total]
        parts = [Stft.transform_range(c, x32, a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    finally:
        S.set_interior("float32")
    w32 = O.transform(o, x32)
    assert z32.dtype == np.complex64 and p32.dtype == np.float32
    np.testing.assert_allclose(z32, w32, rtol=F32_RTOL, atol=F32_ATOL * float(np.max(np.abs(w32))))
    np.testing.assert_allclose(p32, O.power_spectrum(o, x32), rtol=2 * F32_RTOL, atol=F32_ATOL * float(np.max(np.abs(w32))) ** 2)
    assert np.array_equal(np.concatenate(parts, axis=-1), z32)


@pytest.mark.parametrize("fft,hop", [(2048, 512), (1024, 256), (400, 160), (64, 16)])
def test_constant_padding_is_rounded_to_the_input_dtype(fft, hop):
    """The padded signal is built in the INPUT dtype before the float64 interior widens it (stft.ml:318-338): a pad
    value float32 cannot hold reaches the transform as its float32 rounding, also under `set_interior("float64")`
    (found by tools/fuzz_parity.py).  A short right-aligned clip makes the padding dominate every frame."""
    value = 0.9733277781322542
    x = np.random.default_rng(fft).uniform(-1, 1, size=(2, fft // 2 + 11)).astype(np.float32)
    c = Stft.Config.create(fft_size=fft, hop=hop, alignment="right", pad=("constant", value))
    o = O.stft_config(fft, hop=hop, alignment="right", pad="constant", pad_value=value)
    want = O.transform(o, x)
    S.set_interior("float64")
    try:
        z = Stft.transform(c, x)
    finally:
        S.set_interior("float32")
    np.testing.assert_allclose(z, want, rtol=2e-6, atol=2e-7 * float(np.max(np.abs(want))))
    np.testing.assert_allclose(Stft.transform(c, x), want, rtol=F32_RTOL, atol=F32_ATOL * float(np.max(np.abs(want))))


@pytest.mark.parametrize("alignment", ["centered", "left", "right"])
@pytest.mark.parametrize("pad", ["reflect", "edge", ("constant", 0.25)])
@pytest.mark.parametrize("fft,hop", [(2048, 512), (64, 16), (16, 20), (31, 5)])
def test_alignment_and_pad_modes(alignment, pad, fft, hop):
    rng = np.random.default_rng(11)
    n = 3 * fft + 77
    x64 = rng.standard_normal((2, n))
    c = Stft.Config.create(fft_size=fft, hop=hop, alignment=alignment, pad=pad)
    if isinstance(pad, tuple):
        o = O.stft_config(fft, hop=hop, alignment=alignment, pad=pad[0], pad_value=pad[1])
    else:
        o = O.stft_config(fft, hop=hop, alignment=alignment, pad=pad)
    z = Stft.transform(c, x64)
    want = O.transform(o, x64)
    assert z.shape == want.shape
    np.testing.assert_allclose(z, want, rtol=1e-9, atol=1e-10)
    x32 = x64.astype(np.float32)
    got = Stft.power_spectrum(c, x32)
    for i in range(2):
        check_fast(got[i], O.power_spectrum(o, x32)[i], "f32 clip %d" % i)


@pytest.mark.parametrize("n", [1, 2, 7, 1023, 1024, 1025, 2047, 2048, 2049])
def test_short_signals(n):
    """Multi-reflection borders are valid for any n >= 1 (stft.ml:300-305)."""
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n)
    for fft, hop in ((2048, 512), (16, 4)):
        c = Stft.Config.create(fft_size=fft, hop=hop)
        o = O.stft_config(fft, hop=hop)
        assert Stft.frames(c, n) == O.frames(o, n)
        np.testing.assert_allclose(Stft.transform(c, x), O.transform(o, x), rtol=1e-9, atol=1e-10)
        check_fast(Stft.power_spectrum(c, x.astype(np.float32)), O.power_spectrum(o, x.astype(np.float32)),
                   "n=%d fft=%d" % (n, fft))


def test_scale_and_win_length():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 9000)).astype(np.float32)
    for scale in ("magnitude", "psd"):
        c = Stft.Config.create(fft_size=2048, hop=512, win_length=1200, scale=scale)
        o = O.stft_config(2048, hop=512, win_length=1200, scale=scale)
        got, want = Stft.power_spectrum(c, x), O.power_spectrum(o, x)
        for i in range(2):
            check_fast(got[i], want[i], scale)


# ---- structural laws of the reference, now on the HIP path ---------------------------------------

@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("fft,hop,n", [(64, 16, 2000), (2048, 512, 20000)])
def test_transform_range_tiles_exactly(dtype, fft, hop, n):
    """stft_grid.ml:32-73: adjacent ranges reassemble the full transform bit for bit."""
    x = O.lcg_signal(n).astype(dtype)
    c = Stft.Config.create(fft_size=fft, hop=hop)
    full = Stft.transform(c, x)
    total = Stft.frames(c, n)
    cuts = [0, 1, 7, 8, total // 2, total - 1, total]
    parts = [Stft.transform_range(c, x, a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate(parts, axis=-1), full)


@pytest.mark.parametrize("fft,hop", [(64, 16), (2048, 512)])
def test_leading_axes_broadcast_is_per_slice(fft, hop):
    """stft_grid.ml:180-205: a batch is exactly the stack of its slices."""
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, size=(2, 3, 4 * fft + 5)).astype(np.float32)
    c = Stft.Config.create(fft_size=fft, hop=hop)
    full = Stft.power_spectrum(c, x)
    assert full.shape[:2] == (2, 3)
    for i in range(2):
        for j in range(3):
            assert np.array_equal(full[i, j], Stft.power_spectrum(c, x[i, j]))


def test_power_range_device_tiles_exactly():
    """The sharding seam: frame ranges of device-resident audio reassemble exactly."""
    import torch
    x = torch.rand(3, 100000, device="cuda") * 2 - 1
    c = Stft.Config.create(fft_size=2048, hop=512)
    full = Stft.power_spectrum(c, x)
    total = Stft.frames(c, x.shape[-1])
    cuts = [0, 16, 33, 100, total]
    parts = [Stft.power_range(c, x, a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    assert torch.equal(torch.cat(parts, dim=-1), full)
    # device path == host path, bit for bit
    assert np.array_equal(full.cpu().numpy(), Stft.power_spectrum(c, x.cpu().numpy()))


@pytest.mark.parametrize("alignment", ["centered", "left", "right"])
@pytest.mark.parametrize("pad", ["reflect", "edge", ("constant", 0.5)])
@pytest.mark.parametrize("fft,hop", [(16, 4), (32, 7), (16, 20)])
def test_streaming_partition_law(alignment, pad, fft, hop):
    """stft_law.ml:79-164: every chunking of the stream equals the offline transform, exactly."""
    rng = np.random.default_rng(fft * 7 + hop)
    c = Stft.Config.create(fft_size=fft, hop=hop, alignment=alignment, pad=pad)
    for n in (1, 5, fft // 2, fft // 2 + 1, 97, 200):
        x = rng.standard_normal((2, n))
        want = Stft.transform(c, x)
        for trial in range(3):
            k = Stft.Kernel.prepare(c, np.float64, channels=2, max_block=max(1, n))
            parts, pos = [], 0
            while pos < n:
                m = min(int(rng.integers(0, 9)) if trial else 1, n - pos)
                out = k.step(x[:, pos:pos + m])
                pos += m
                if out is not None:
                    parts.append(out)
            out = k.flush()
            if out is not None:
                parts.append(out)
            got = np.concatenate(parts, axis=-1) if parts else np.zeros((2, c.bins, 0), np.complex128)
            assert got.shape == want.shape, (n, trial, got.shape, want.shape)
            assert np.array_equal(got, want), (n, trial)
            with pytest.raises(S.InvalidArgument, match="step: cannot feed a drained kernel"):
                k.step(x[:, :1])
            k.reset()


def test_streaming_kernel_follows_the_first_chunks_leading_shape():
    """stft.ml:603-617 checks only `channels >= 1` at prepare and the state takes its leading shape from the chunks
    (stft.ml:521-559): a kernel prepared for 4 channels analyses a 1-channel stream, bit for bit the 1-channel transform,
    and a stream that already holds samples refuses another count (the library reads and writes exactly the rows of the
    count it holds: a binding must never pass buffers of another extent -- ADVICE round 2)."""
    rng = np.random.default_rng(77)
    c = Stft.Config.create(fft_size=32, hop=8)
    x = rng.standard_normal((1, 300))
    k = Stft.Kernel.prepare(c, np.float64, channels=4, max_block=300)
    parts = [o for o in (k.step(x[:, :100]), k.step(x[:, 100:]), k.flush()) if o is not None]
    got = np.concatenate(parts, axis=-1)
    want = Stft.transform(c, x)
    assert got.shape == want.shape == (1, c.bins, Stft.frames(c, 300))
    assert np.array_equal(got, want)
    k.reset()
    k.step(rng.standard_normal((3, 50)))                              # a fresh stream may take another shape ...
    with pytest.raises(S.InvalidArgument, match="step: cannot feed 2 channels to a stream of 3"):
        k.step(rng.standard_normal((2, 50)))                          # ... one that holds samples may not
    import ctypes
    from soundml_amd._lib import check, lib
    have = ctypes.c_int64()
    check(lib.smx_stft_kernel_channels(k._h, ctypes.byref(have)))
    assert have.value == 3


@pytest.mark.parametrize("fft,hop,alignment,pad", [(2048, 512, "centered", "reflect"), (64, 16, "right", "edge"), (32, 7, "left", ("constant", 0.25)),
                                                   (16, 20, "centered", "reflect"), (1024, 256, "centered", ("constant", 0.0))])
def test_streaming_on_device_resident_chunks(fft, hop, alignment, pad):
    """smx_stft_kernel_step_dev / flush_dev: the state machine fed from and emitting into device memory (prelude, tail and
    the border gathers live on the device) gives the host-chunk stream -- and therefore Stft.transform -- bit for bit,
    under the partition law's chunkings (stft_law.ml:79-164), both faces."""
    import torch
    rng = np.random.default_rng(fft + hop)
    c = Stft.Config.create(fft_size=fft, hop=hop, alignment=alignment, pad=pad)
    for n in (1, fft // 2, fft // 2 + 1, 3 * fft + 5, 20000 if fft >= 1024 else 700):
        x = rng.uniform(-1, 1, size=(2, n)).astype(np.float32)
        want = Stft.transform(c, x)
        xd = torch.from_numpy(x).cuda()
        for trial in range(3):
            k = Stft.Kernel.prepare(c, np.float32, channels=2, max_block=max(1, n))
            parts, pos = [], 0
            while pos < n:
                m = min([n, 1, int(rng.integers(1, 3000))][trial], n - pos)
                out = k.step(xd[:, pos:pos + m])
                pos += m
                if out is not None:
                    assert out.is_cuda
                    parts.append(out.cpu().numpy())
            out = k.flush()
            if out is not None:
                parts.append(out.cpu().numpy())
            got = np.concatenate(parts, axis=-1) if parts else np.zeros((2, c.bins, 0), np.complex64)
            assert got.shape == want.shape and np.array_equal(got, want), (n, trial)


def test_streaming_2048_float32():
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, size=(2, 30000)).astype(np.float32)
    c = Stft.Config.create(fft_size=2048, hop=512)
    want = Stft.transform(c, x)
    k = Stft.Kernel.prepare(c, np.float32, channels=2, max_block=4096)
    parts = [k.step(x[:, i:i + 4096]) for i in range(0, 30000, 4096)] + [k.flush()]
    got = np.concatenate([p for p in parts if p is not None], axis=-1)
    assert np.array_equal(got, want)



# ---- least-squares synthesis: Stft.invert --------------------------------------------------------------

@pytest.mark.parametrize("vectors", ISTFT_FILES)
def test_invert_goldens(vectors):
    """The reference's librosa-0.11 synthesis vectors (istft_goldens.ml) on the HIP path: float64 cases at the
    reference's float64 tolerance, float32 cases at its float32 tolerance with the float64 interior and at the
    north-star tolerance with the float32 interior."""
    from conftest import F32_ATOL, F32_RTOL, F64_ATOL, F64_RTOL, check_close, istft_golden_config, istft_golden_spectrum, load_golden
    for case in load_golden("istft", vectors)["cases"]:
        p = case["params"]
        c = istft_golden_config(lambda fft, pad, pad_value, **kw: Stft.Config.create(fft_size=fft, pad=(pad, pad_value), **kw), p)
        z = istft_golden_spectrum(p["fft_size"], p["frames"])
        want = np.array(case["values"])
        if p["dtype"] == "float64":
            got = Stft.invert(c, z, p.get("length"))
            assert got.dtype == np.float64
            check_close(got, want, shape=case["shape"], rtol=F64_RTOL, atol=F64_ATOL, msg=case["name"])
        else:
            S.set_interior("float64")
            got = Stft.invert(c, z.astype(np.complex64), p.get("length"))
            assert got.dtype == np.float32
            check_close(got, want, shape=case["shape"], rtol=F32_RTOL, atol=F32_ATOL, msg=case["name"])
            S.set_interior("float32")
            check_fast(Stft.invert(c, z.astype(np.complex64), p.get("length")), want, case["name"])


@pytest.mark.parametrize("fft,hop,win,alignment", [(2048, 512, None, "centered"), (2048, 512, 1200, "left"), (1024, 256, None, "right"),
                                                   (64, 16, None, "centered"), (48, 12, None, "centered"), (31, 5, None, "left"),
                                                   (64, 60, None, "centered"), (100, 33, 64, "right"),
                                                   (512, 128, None, "centered"), (1024, 256, 800, "left"), (2048, 500, None, "centered"),
                                                   (4096, 1024, None, "centered"), (1024, 512, None, "centered"), (512, 256, 400, "left"),
                                                   (2048, 1024, None, "right")])
@pytest.mark.parametrize("length_mode", ["default", "short", "long"])
def test_invert_vs_oracle(fft, hop, win, alignment, length_mode):
    """Random (inconsistent) spectra: the least-squares solution itself, every alignment, power-of-two and other
    sizes, default / cut / zero-extended lengths, two leading axes."""
    rng = np.random.default_rng(fft * 7 + hop)
    c = Stft.Config.create(fft_size=fft, hop=hop, win_length=win, alignment=alignment)
    o = O.stft_config(fft, hop=hop, win_length=win, alignment=alignment)
    frames = 23
    z = rng.standard_normal((2, 3, fft // 2 + 1, frames)) + 1j * rng.standard_normal((2, 3, fft // 2 + 1, frames))
    default = O.output_length(o, frames)
    length = {"default": None, "short": max(1, default // 3), "long": default + 2 * fft + 5}[length_mode]
    want = O.invert(o, z, length)
    got = Stft.invert(c, z, length)
    assert got.shape == want.shape and got.dtype == np.float64
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-11)
    z32 = z.astype(np.complex64)
    got32 = Stft.invert(c, z32, length)
    assert got32.dtype == np.float32
    check_fast(got32, O.invert(o, z32, length), "float32")


def test_invert_round_trip_and_device_path():
    """invert (transform x) restores x on the interior (stft.mli, istft_law.ml); a device-resident spectrum gives
    the host result bit for bit."""
    import torch
    rng = np.random.default_rng(21)
    x = rng.uniform(-1, 1, size=(3, 40000)).astype(np.float32)
    c = Stft.Config.create(fft_size=2048, hop=512)
    z = Stft.transform(c, x)
    y = Stft.invert(c, z, length=x.shape[-1])
    assert y.shape == x.shape and y.dtype == np.float32
    assert np.max(np.abs(y - x)) < 2e-5
    zd = torch.from_numpy(z).cuda()
    yd = Stft.invert(c, zd, length=x.shape[-1])
    assert yd.is_cuda and np.array_equal(yd.cpu().numpy(), y)
    assert np.array_equal(Stft.invert(c, zd).cpu().numpy(), Stft.invert(c, z))


GL_FILES = ["griffinlim_fft64_hop16", "griffinlim_fft64_hop16_win40", "griffinlim_fft512_hop128"]


@pytest.mark.parametrize("vectors", GL_FILES)
def test_griffin_lim_goldens(vectors):
    """librosa.griffinlim vectors of the reference (gl_goldens.ml) on the HIP path: up to 32 synthesis / analysis
    round trips, float64 at the reference's 1e-9 / 1e-12."""
    from conftest import F32_ATOL, F32_RTOL, F64_ATOL, F64_RTOL, check_close, istft_golden_config, load_golden
    from test_oracle_goldens import gl_golden_magnitudes
    for case in load_golden("istft", vectors)["cases"]:
        p = case["params"]
        c = istft_golden_config(lambda fft, pad, pad_value, **kw: Stft.Config.create(fft_size=fft, pad=(pad, pad_value), **kw), p)
        mag = gl_golden_magnitudes(p["fft_size"], p["frames"])
        f32 = p["dtype"] == "float32"
        if f32:
            S.set_interior("float64")
        got = Stft.griffin_lim(c, mag.astype(np.float32) if f32 else mag, n_iter=p["n_iter"], momentum=p["momentum"],
                               length=p.get("length"))
        S.set_interior("float32")
        check_close(got, case["values"], shape=case["shape"], rtol=F32_RTOL if f32 else F64_RTOL,
                    atol=F32_ATOL if f32 else F64_ATOL, msg=case["name"])


def test_griffin_lim_fast_path_and_messages():
    """fft 2048 / hop 512 float32 (fused synthesis + fused complex analysis kernels in the loop): the iteration is a
    non-expansive map, so float32 rounding stays at the 1e-5 level of the signal's peak after 8 round trips; initial
    phase, batch and device-resident inputs; the reference's messages."""
    import torch
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, size=(2, 20000)).astype(np.float32)
    c = Stft.Config.create(fft_size=2048, hop=512)
    o = O.stft_config(2048, hop=512)
    z = Stft.transform(c, x)
    mag = np.abs(z).astype(np.float32)
    phase = (np.angle(z) + 0.3 * rng.standard_normal(z.shape)).astype(np.float32)
    want = O.griffin_lim(o, mag, 8, 0.99, phase, None)
    got = Stft.griffin_lim(c, mag, n_iter=8, momentum=0.99, init=phase)
    assert got.shape == want.shape and got.dtype == np.float32
    # float32 round trips: 1e-4 of the peak on the interior; on the last fft_size samples of a clip the envelope
    # falls to ~1e-4 of its interior value and divides the rounding of every pass (the reference's loop is
    # float64 there -- set_interior("float64") reproduces it, see test_griffin_lim_goldens)
    peak = np.max(np.abs(want))
    assert np.max(np.abs(got - want)[:, 2048:-2048]) < 2e-4 * peak
    assert np.linalg.norm(got - want) < 1e-3 * np.linalg.norm(want)
    gd = Stft.griffin_lim(c, torch.from_numpy(mag).cuda(), n_iter=8, momentum=0.99, init=torch.from_numpy(phase).cuda())
    assert gd.is_cuda and np.array_equal(gd.cpu().numpy(), got)
    with pytest.raises(S.InvalidArgument) as e:
        Stft.griffin_lim(c, mag, n_iter=0)
    assert str(e.value) == "griffin_lim: cannot run 0 iterations (n_iter must be at least 1)"
    with pytest.raises(S.InvalidArgument) as e:
        Stft.griffin_lim(c, mag, momentum=-0.5)
    assert str(e.value) == "griffin_lim: cannot use a momentum of -0.5 (momentum must be non-negative)"
    with pytest.raises(S.InvalidArgument) as e:
        Stft.griffin_lim(c, mag, init=phase[:1])
    assert str(e.value).startswith("griffin_lim: cannot start from a [1; 1025; 40] phase for a [2; 1025; 40] spectrogram")


GL_GATE_K = 16.0   # see test_griffin_lim_defaults_statistical_gate


def test_griffin_lim_defaults_statistical_gate():
    """The reference's defaults (stft.ml:961-964: 32 iterations, momentum 0.99, random initial phase) under the float32 interior,
    gated STATISTICALLY over 48 seeds (round 5's gate was two seeds and a convergence fallback).
    The accelerated update forms c_k - 0.99 c_(k-1), which cancels to ~1 % of |c|, and unit() of a bin whose difference nearly
    vanishes turns by O(1) under a perturbation of its size: the distance between two trajectories that differ by rounding is
    heavy-tailed (chaotic dynamics), so a single draw says little and a distribution says what there is to say.
    Yardstick: O.griffin_lim_float32_storage -- the float64 ORACLE with nothing but its stored intermediates rounded to float32,
    the least any float32 interior can do.  It injects 2^-24 = 6.0e-8 relative per stored value and iteration; this library's
    float32 transforms inject their own rounding on top, measured 4.5e-7 of the peak per transform over all of C2
    (bench.py, gpu_vs_oracle_max_err_over_peak) = 7.5 x the yardstick's.  The same dynamics amplify both, so the device's
    distance distribution must be the yardstick's scaled by about that factor: GATE K = 16 (2 x 7.5) on the MEDIAN and on the
    MAXIMUM of the relative l2 distance to the float64 oracle.  Measured on 48 seeds (profiles/r08/gl_stat_f32.log): device
    median 1.18e-4 / max 1.97e-2, yardstick median 1.51e-5 / max 1.50e-2: ratios 7.8 and 1.3.
    Beside it, for every seed: the result is finite and does what Griffin-Lim is for -- its spectral convergence
    || |STFT(y)| - S || / ||S|| is the oracle's within 1 % (measured: within 0.006 %); and on the first seed the trajectory
    while it is still a rounding error (2 iterations within 1e-5) and the reference's own arithmetic (set_interior float64
    within 1e-5 of the oracle after all 32)."""
    c = Stft.Config.create(fft_size=2048, hop=512)
    o = O.stft_config(2048, hop=512)
    rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
    dev, yard = [], []
    for seed in range(1000, 1048):
        rng = np.random.default_rng(seed)
        x = rng.uniform(-1, 1, size=(1, 24000)).astype(np.float32)
        mag = np.abs(Stft.transform(c, x)).astype(np.float32)
        phase = rng.uniform(-np.pi, np.pi, size=mag.shape).astype(np.float32)   # (the default is a random phase: stft.ml:976-984)
        want = O.griffin_lim(o, mag, 32, 0.99, phase, None).astype(np.float64)
        got = Stft.griffin_lim(c, mag, n_iter=32, momentum=0.99, init=phase)
        assert got.shape == want.shape and np.isfinite(got).all(), seed
        dev.append(rel(got, want))
        yard.append(rel(O.griffin_lim_float32_storage(o, mag, 32, 0.99, phase, None), want))
        conv = lambda y: float(np.linalg.norm(np.abs(O.transform(o, y.astype(np.float64))) - mag) / np.linalg.norm(mag))
        cg, cw = conv(got), conv(want)
        assert cg <= 1.01 * cw + 1e-6, (seed, cg, cw)
        if seed == 1000:
            short = O.griffin_lim(o, mag, 2, 0.99, phase, None).astype(np.float64)
            assert rel(Stft.griffin_lim(c, mag, n_iter=2, momentum=0.99, init=phase), short) <= 1e-5
            S.set_interior("float64")
            try:
                strict = Stft.griffin_lim(c, mag, n_iter=32, momentum=0.99, init=phase)
            finally:
                S.set_interior("float32")
            assert np.linalg.norm(strict - want) <= 1e-5 * np.linalg.norm(want)
    dev, yard = np.asarray(dev), np.asarray(yard)
    assert np.median(dev) <= GL_GATE_K * np.median(yard), (np.median(dev), np.median(yard))
    assert dev.max() <= GL_GATE_K * yard.max(), (dev.max(), yard.max())


@pytest.mark.parametrize("fft,hop", [(1024, 256), (512, 100), (4096, 1024)])
def test_griffin_lim_other_sizes(fft, hop):
    """The folded loop on the Stockham frames kernel (fft 512 .. 4096, any hop): against the oracle after 6 round
    trips, with an initial phase and two leading axes."""
    rng = np.random.default_rng(fft)
    x = rng.uniform(-1, 1, size=(2, 2, 6 * fft + 321)).astype(np.float32)
    c = Stft.Config.create(fft_size=fft, hop=hop)
    o = O.stft_config(fft, hop=hop)
    z = Stft.transform(c, x)
    mag = np.abs(z).astype(np.float32)
    phase = (np.angle(z) + 0.3 * rng.standard_normal(z.shape)).astype(np.float32)
    want = O.griffin_lim(o, mag, 6, 0.99, phase, None)
    got = Stft.griffin_lim(c, mag, n_iter=6, momentum=0.99, init=phase)
    assert got.shape == want.shape and got.dtype == np.float32
    peak = np.max(np.abs(want))
    assert np.max(np.abs(got - want)[..., fft:-fft]) < 2e-4 * peak
    assert np.linalg.norm(got - want) < 1e-3 * np.linalg.norm(want)


@pytest.mark.parametrize("n_iter,momentum,length", [(1, 0.99, None), (2, 0.99, 17000), (3, 0.0, None), (5, 0.5, 23456)])
def test_griffin_lim_folded_update_short_runs(n_iter, momentum, length):
    """The fused loop never materialises S * angles nor the angles (the synthesis kernel forms S * unit(c_k - beta
    c_(k-1)) while it stages the spectrum): one iteration (no previous spectrum at the end), two (none inside), zero
    momentum, and an explicit output length, against the oracle and against the unfused loop
    (SMX_DISABLE_FAST is read at load time, so the float64 interior stands in: the same loop, materialised)."""
    rng = np.random.default_rng(n_iter)
    x = rng.uniform(-1, 1, size=(2, 20000)).astype(np.float32)
    c = Stft.Config.create(fft_size=2048, hop=512)
    o = O.stft_config(2048, hop=512)
    mag = np.abs(Stft.transform(c, x)).astype(np.float32)
    want = O.griffin_lim(o, mag, n_iter, momentum, None, length)
    got = Stft.griffin_lim(c, mag, n_iter=n_iter, momentum=momentum, length=length)
    assert got.shape == want.shape and got.dtype == np.float32
    peak = np.max(np.abs(want))
    assert np.max(np.abs(got - want)[:, 2048:-2048]) < 2e-4 * peak
    assert np.linalg.norm(got - want) < 1e-3 * np.linalg.norm(want)
    S.set_interior("float64")
    try:
        strict = Stft.griffin_lim(c, mag, n_iter=n_iter, momentum=momentum, length=length)
    finally:
        S.set_interior("float32")
    np.testing.assert_allclose(strict, want, rtol=1e-5, atol=1e-6 * peak)


# ---- Mel -------------------------------------------------------------------------------------------

@pytest.mark.parametrize("n_mels,sr,fft,frames,lead", [(128, 48000, 2048, 938, 2), (40, 22050, 512, 77, 3),
                                                        (13, 16000, 128, 5, 1), (128, 48000, 2048, 1, 1)])
def test_mel_apply_vs_oracle(n_mels, sr, fft, frames, lead):
    rng = np.random.default_rng(n_mels + frames)
    mc = Mel.Config.create(n_mels=n_mels, sample_rate=sr, fft_size=fft)
    oc = O.mel_config(n_mels, sr, fft)
    s64 = rng.uniform(0, 4, size=(lead, fft // 2 + 1, frames))
    np.testing.assert_allclose(Mel.apply(mc, s64), O.mel_apply(oc, s64), rtol=1e-12, atol=1e-14)
    s32 = s64.astype(np.float32)
    got, want = Mel.apply(mc, s32), O.mel_apply(oc, s32)
    assert got.dtype == np.float32 and got.shape == (lead, n_mels, frames)
    for i in range(lead):
        check_fast(got[i], want[i], "mel clip %d" % i)
    # broadcast == per slice, exactly (mel_props.ml:136-155)
    assert np.array_equal(got[0], Mel.apply(mc, s32[0]))


def test_mel_spectrogram_is_the_composition():
    """mel_props.ml:183-194: mel_spectrogram = apply . power_spectrum."""
    rng = np.random.default_rng(21)
    x = rng.uniform(-1, 1, size=(2, 48000)).astype(np.float32)
    sc = Stft.Config.create(fft_size=2048, hop=512)
    mc = Mel.Config.create(n_mels=128, sample_rate=48000, fft_size=2048)
    got = S.mel_spectrogram(sc, mc, x)
    comp = Mel.apply(mc, Stft.power_spectrum(sc, x))
    want = O.mel_spectrogram(O.stft_config(2048, hop=512), O.mel_config(128, 48000, 2048), x)
    for i in range(2):
        check_fast(got[i], want[i], "mel_spectrogram")
        check_fast(comp[i], want[i], "composition")


@pytest.mark.parametrize("n_mels,sr,n,lead,power", [
    (128, 48000, 48000, 2, 2.0),     # C3 geometry
    (128, 48000, 16 * 512 * 5 + 1234, 3, 2.0),   # several tiles + tail tile
    (128, 48000, 3000, 4, 2.0),      # every frame is a border frame
    (80, 16000, 20000, 2, 2.0),      # 5 mel blocks
    (40, 22050, 30000, 1, 1.0),      # magnitude mel, 3 mel blocks (one partly empty)
    (13, 48000, 9000, 2, 2.0),       # single partial block
    (200, 48000, 20000, 1, 2.0),     # 13 mel blocks
])
def test_fused_mel_spectrogram_vs_oracle(n_mels, sr, n, lead, power):
    """The fused audio -> mel kernel (fft 2048): MFMA over the LDS-resident power tile."""
    rng = np.random.default_rng(n_mels + n)
    x = rng.uniform(-1, 1, size=(lead, n)).astype(np.float32)
    sc = Stft.Config.create(fft_size=2048, hop=512)
    mc = Mel.Config.create(n_mels=n_mels, sample_rate=sr, fft_size=2048)
    got = S.mel_spectrogram(sc, mc, x, power)
    want = O.mel_spectrogram(O.stft_config(2048, hop=512), O.mel_config(n_mels, sr, 2048), x, power)
    assert got.shape == want.shape and got.dtype == np.float32
    for i in range(lead):
        check_fast(got[i], want[i], "fused mel clip %d" % i)
    # a batch is exactly the stack of its slices (deterministic reduction order)
    assert np.array_equal(got[0], S.mel_spectrogram(sc, mc, x[0], power))
    # device-resident path == host path, bit for bit
    import torch
    assert np.array_equal(S.mel_spectrogram(sc, mc, torch.from_numpy(x).cuda(), power).cpu().numpy(), got)



def test_mel_spectrogram_mixed_radix_large_batch_equals_its_slices():
    """mel_props.ml:136-155 at a size with a mixed-radix plan (fft 400 / hop 160, 80 mels: whisper's front end) on a batch of
    more than 65 536 frames, where round 2's launcher switched from the fused kernel to power + Mel.apply: a clip's values
    must not depend on what it is batched with -- slices of the large batch, computed alone and in a small
    batch, are the large batch's rows bit for bit."""
    rng = np.random.default_rng(400)
    lead, n = 700, 16000                     # 101 frames per clip -> 70 700 frames
    x = rng.uniform(-1, 1, size=(lead, n)).astype(np.float32)
    sc = Stft.Config.create(fft_size=400, hop=160)
    mc = Mel.Config.create(n_mels=80, sample_rate=16000, fft_size=400)
    big = S.mel_spectrogram(sc, mc, x)
    assert big.shape == (lead, 80, Stft.frames(sc, n)) and lead * big.shape[2] > 65536
    for i in (0, 1, 349, 699):
        assert np.array_equal(big[i], S.mel_spectrogram(sc, mc, x[i])), i
    assert np.array_equal(big[10:13], S.mel_spectrogram(sc, mc, x[10:13]))
    want = O.mel_spectrogram(O.stft_config(400, hop=160), O.mel_config(80, 16000, 400), x[:2])
    for i in range(2):
        check_fast(big[i], want[i], "clip %d" % i)


@pytest.mark.parametrize("fft,hop,n_mels,sr,n,lead,power", [
    (1024, 256, 80, 22050, 22050, 3, 2.0),      # the usual vocoder front end
    (1024, 256, 128, 44100, 16 * 256 * 3 + 777, 2, 1.0),   # eight row tiles (one per wave), ragged last tile, magnitude
    (512, 128, 40, 16000, 9000, 2, 2.0),        # four waves, three row tiles (one partly empty)
    (512, 160, 13, 16000, 4000, 1, 2.0),        # a single partial row tile, hop that does not divide the size
    (1024, 300, 136, 16000, 30000, 1, 2.0),     # nine row tiles over 8 waves
    (400, 160, 80, 16000, 16000 * 3, 2, 2.0),   # whisper's front end: chirp-z at M = 512, the same MFMA tail
    (100, 33, 10, 8000, 3000, 1, 1.0),          # M = 256
    (1000, 250, 64, 22050, 9000, 1, 2.0),       # M = 1024: one 1024-thread workgroup
])
def test_fused_mel_spectrogram_512_1024(fft, hop, n_mels, sr, n, lead, power):
    """Soundml.mel_spectrogram for fft 512 / 1024 (stft_stockham_power16_kernel<.., MEL>): the power columns stay in
    LDS and meet the banded filterbank on the fp32 MFMA; against the oracle, against the unfused composition, slices
    of a batch and the device-resident face bit for bit; dense caller-supplied weights through the same kernel."""
    import torch
    rng = np.random.default_rng(fft + n_mels + n)
    x = rng.uniform(-1, 1, size=(lead, n)).astype(np.float32)
    sc = Stft.Config.create(fft_size=fft, hop=hop)
    mc = Mel.Config.create(n_mels=n_mels, sample_rate=sr, fft_size=fft)
    got = S.mel_spectrogram(sc, mc, x, power)
    want = O.mel_spectrogram(O.stft_config(fft, hop=hop), O.mel_config(n_mels, sr, fft), x, power)
    assert got.shape == want.shape and got.dtype == np.float32
    for i in range(lead):
        check_fast(got[i], want[i], "mel clip %d" % i)
        check_fast(Mel.apply(mc, Stft.power_spectrum(sc, x[i], power)), want[i], "composition clip %d" % i)
    assert np.array_equal(got[0], S.mel_spectrogram(sc, mc, x[0], power))
    assert np.array_equal(S.mel_spectrogram(sc, mc, torch.from_numpy(x).cuda(), power).cpu().numpy(), got)
    w = np.abs(rng.standard_normal((12, fft // 2 + 1)))
    dense = S.mel_spectrogram(sc, Mel.Config.from_weights(w, fft), x, power)
    wd = np.einsum("mb,lbt->lmt", w, O.power_spectrum(O.stft_config(fft, hop=hop), x, power).astype(np.float64))
    for i in range(lead):
        check_fast(dense[i], wd[i], "dense weights clip %d" % i)


@pytest.mark.parametrize("fft", [512, 1024])
@pytest.mark.parametrize("alignment,pad", [("centered", "reflect"), ("left", "edge"), ("right", ("constant", 0.25)), ("right", "reflect")])
def test_sixteen_frame_kernels_on_short_and_odd_inputs(fft, alignment, pad):
    """The stage-free 16-frame kernels (fft 512 / 1024 power and mel) where every frame is a border frame: signals
    shorter than the window, a single sample, hops larger than the size, every alignment and pad mode."""
    rng = np.random.default_rng(fft)
    mc = Mel.Config.create(n_mels=24, sample_rate=16000, fft_size=fft)
    om = O.mel_config(24, 16000, fft)
    for n, hop in ((1, 7), (fft // 3, fft // 4), (fft + 1, 3 * fft), (5 * fft + 11, fft // 4 + 1)):
        x = rng.uniform(-1, 1, size=(2, n)).astype(np.float32)
        c = Stft.Config.create(fft_size=fft, hop=hop, alignment=alignment, pad=pad)
        o = (O.stft_config(fft, hop=hop, alignment=alignment, pad=pad[0], pad_value=pad[1]) if isinstance(pad, tuple)
             else O.stft_config(fft, hop=hop, alignment=alignment, pad=pad))
        want = O.power_spectrum(o, x)
        got = Stft.power_spectrum(c, x)
        assert got.shape == want.shape
        if want.size:
            for i in range(2):
                check_fast(got[i], want[i], "power n=%d hop=%d" % (n, hop))
                check_fast(S.mel_spectrogram(c, mc, x)[i], O.mel_spectrogram(o, om, x)[i], "mel n=%d hop=%d" % (n, hop))


@pytest.mark.parametrize("fft,hop", [(1024, 256), (512, 100), (400, 160), (4096, 1024), (2048, 512)])
@pytest.mark.parametrize("power", [0.5, 1.0, 3.0])
def test_general_powers_on_every_power_path(fft, hop, power):
    """|X|^p for exponents other than 2 (stft.ml:670-674: magnitude first, then the power) through the stage-free,
    chirp-z and fused kernels, float32 interior and the float64 one."""
    rng = np.random.default_rng(fft + int(10 * power))
    x = rng.uniform(-1, 1, size=(2, 5 * fft + 77)).astype(np.float32)
    c = Stft.Config.create(fft_size=fft, hop=hop)
    want = O.power_spectrum(O.stft_config(fft, hop=hop), x, power)
    got = Stft.power_spectrum(c, x, power)
    for i in range(2):
        check_fast(got[i], want[i], "float32 interior")
    S.set_interior("float64")
    try:
        strict = Stft.power_spectrum(c, x, power)
    finally:
        S.set_interior("float32")
    np.testing.assert_allclose(strict, want, rtol=4 * F32_RTOL, atol=F32_ATOL * float(np.max(want)))


def test_filterbank_from_weights():
    """Caller-supplied dense weights (the shape of Chroma.apply, chroma.ml:307: 12 rows over all bins) through the
    same entry points: W @ S in float64 for float64 spectrograms, the float32 MFMA kernel for float32 ones, and the
    spectrogram composition from audio (no banded structure, so the unfused path)."""
    rng = np.random.default_rng(12)
    w = np.abs(rng.standard_normal((12, 1025)))
    bank = Mel.Config.from_weights(w, 2048)
    assert bank.n_mels == 12 and bank.bins == 1025
    s64 = np.abs(rng.standard_normal((3, 1025, 41)))
    np.testing.assert_allclose(Mel.apply(bank, s64), np.einsum("mb,lbt->lmt", w, s64), rtol=1e-12, atol=1e-12)
    s32 = s64.astype(np.float32)
    check_fast(Mel.apply(bank, s32), np.einsum("mb,lbt->lmt", w, s32.astype(np.float64)), "float32 apply")
    x = rng.uniform(-1, 1, size=(2, 20000)).astype(np.float32)
    sc = Stft.Config.create(fft_size=2048, hop=512)
    want = np.einsum("mb,lbt->lmt", w, O.power_spectrum(O.stft_config(2048, hop=512), x).astype(np.float64))
    check_fast(S.mel_spectrogram(sc, bank, x), want, "chroma-like spectrogram")


# ---- Pipeline-stage faces of the streaming kernel (stft.ml:1301-1409) -----------------------------------------------

@pytest.mark.parametrize("fft,hop,alignment,pad", [(2048, 512, "centered", "reflect"), (64, 16, "right", "reflect"),
                                                   (64, 100, "left", "edge"), (1024, 256, "centered", ("constant", 0.5))])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_stage_and_power_stage_stream_the_offline_result(fft, hop, alignment, pad, dtype):
    """Stft.stage / Stft.power_stage: whatever the chunking, the concatenated steps and flush chunks are the offline
    transform / power_spectrum bit for bit (the partition law, stft_law.ml:79-164, through the stage bodies), every
    emitted chunk honours the threaded frame bound, and reset rewinds."""
    rng = np.random.default_rng(fft + hop)
    n = 5 * fft + 3 * hop + 17
    x = rng.standard_normal((2, n)).astype(dtype)
    c = Stft.Config.create(fft_size=fft, hop=hop, alignment=alignment, pad=pad)
    for factory, offline in ((Stft.stage(c), Stft.transform(c, x)), (Stft.power_stage(c, 1.0), Stft.power_spectrum(c, x, 1.0)),
                             (Stft.power_stage(c), Stft.power_spectrum(c, x))):
        for block in (n, 1000, 333):
            st = factory.prepare(max_items=block)
            assert st.latency == Stft.stage_latency(c) and st.bound == Stft.frame_bound(c, block)
            for _ in range(2):                                   # second round after reset
                parts = []
                for i in range(0, n, block):
                    out = st.step(x[:, i:i + block])
                    if out is not None:
                        assert out.shape[-1] <= st.bound
                        parts.append(out)
                tail = st.flush()
                assert all(t.shape[-1] <= st.bound for t in tail)
                got = st.concat(parts + tail)
                assert got.dtype == offline.dtype and np.array_equal(got, offline)
                st.reset()
    fresh = Stft.power_stage(c).prepare()
    assert fresh.flush() == []
    with pytest.raises(S.InvalidArgument) as e:
        fresh.concat([])
    assert str(e.value) == "power_stage: cannot concatenate zero chunks before any chunk fixed the element dtype"
    assert Stft.stage(c).prepare().concat([]).shape == (c.bins, 0)


# ---- decibel conversions (power_to_db) ----------------------------------------------------------

@pytest.mark.parametrize("which", ["power_to_db", "amplitude_to_db"])
def test_db_goldens(which):
    """librosa's power_to_db / amplitude_to_db vectors (soundml/test/db/test_golden.ml: float32 1e-4, float64 1e-10) on
    the HIP path, host and device-resident."""
    import torch
    from test_oracle_goldens import db_golden_cases
    fn = getattr(S, which)
    for case, x, p, dt in db_golden_cases(which):
        got = fn(x, reference=p["reference"], amin=p["amin"], top_db=p["top_db"])
        assert got.dtype == dt and got.shape == x.shape
        tol = 1e-4 if dt == np.float32 else 1e-10
        check_close(got, case["values"], shape=case["shape"], rtol=tol, atol=tol, msg=case["name"])
        if dt == np.float32:
            gd = fn(torch.from_numpy(x).cuda(), reference=p["reference"], amin=p["amin"], top_db=p["top_db"])
            assert gd.is_cuda and np.array_equal(gd.cpu().numpy(), got)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_db_vs_oracle_large(dtype):
    """A spectrogram-sized tensor: the clamp is under the maximum of the WHOLE tensor (one reduction over 4 M values),
    negative powers sit at the floor, amplitudes take magnitudes first."""
    rng = np.random.default_rng(9)
    x = (rng.standard_normal((3, 1025, 1300)) * np.exp(rng.uniform(-20, 5, size=(3, 1025, 1300)))).astype(dtype)
    tol = 2e-6 if dtype == np.float32 else 1e-12
    for top_db in (None, 80.0, 0.0):
        for fn, on in ((S.power_to_db, O.power_to_db), (S.amplitude_to_db, O.amplitude_to_db)):
            got, want = fn(x, reference=2.0, top_db=top_db), on(x, reference=2.0, top_db=top_db)
            assert got.dtype == dtype
            np.testing.assert_allclose(got, want, rtol=tol, atol=tol * 100)


# ---- log-mel / MFCC tail -----------------------------------------------------------------------------------

def test_mfcc_goldens():
    """librosa.feature.mfcc vectors of the reference (mel_goldens.ml:131-158) on the HIP path, the reference's
    tolerances; float32 cases with the float64 STFT interior (its contract) and with the float32 one."""
    from conftest import check_close, load_golden
    from test_oracle_goldens import _mfcc_case
    for case in load_golden("mel", "mfcc")["cases"]:
        sc, mc, x, n_mfcc, lifter = _mfcc_case(
            case, lambda fft, **kw: Stft.Config.create(fft_size=fft, **kw),
            lambda n_mels, sr, fft, **kw: Mel.Config.create(n_mels=n_mels, sample_rate=sr, fft_size=fft, **kw), O.lcg_signal)
        f32 = case["params"]["dtype"] == "float32"
        for interior in (("float64", "float32") if f32 else ("float64",)):
            S.set_interior(interior)
            got = S.mfcc(sc, mc, x, n_mfcc=n_mfcc, lifter=lifter)
            assert got.dtype == (np.float32 if f32 else np.float64)
            check_close(got, case["values"], shape=case["shape"], rtol=1e-4 if f32 else 1e-9,
                        atol=1e-4 if f32 else 1e-9, msg=case["name"] + " " + interior)
        S.set_interior("float32")


@pytest.mark.parametrize("n_mfcc,lifter", [(20, None), (13, 22.0), (128, None)])
def test_mfcc_vs_oracle_batch(n_mfcc, lifter):
    """C3 geometry (fft 2048, 128 mels, fused mel kernel underneath), a batch with very different levels so that
    the 80 dB clamp under the GLOBAL maximum acts on the quiet clips; device-resident input gives the host result."""
    import torch
    rng = np.random.default_rng(77)
    x = rng.uniform(-1, 1, size=(3, 30000)).astype(np.float32)
    x[1] *= 1e-3
    x[2, 15000:] = 0.0
    sc = Stft.Config.create(fft_size=2048, hop=512)
    mc = Mel.Config.create(n_mels=128, sample_rate=48000, fft_size=2048)
    want = O.mfcc(O.stft_config(2048, hop=512), O.mel_config(128, 48000, 2048), x, n_mfcc, lifter)
    got = S.mfcc(sc, mc, x, n_mfcc=n_mfcc, lifter=lifter)
    assert got.shape == want.shape == (3, n_mfcc, 59) and got.dtype == np.float32
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-3)   # log domain: 1e-5 relative on a power is 4e-5 dB per band
    gd = S.mfcc(sc, mc, torch.from_numpy(x).cuda(), n_mfcc=n_mfcc, lifter=lifter)
    assert gd.is_cuda and np.array_equal(gd.cpu().numpy(), got)


def test_mfcc_messages():
    sc = Stft.Config.create(fft_size=512, hop=128)
    mc = Mel.Config.create(n_mels=40, sample_rate=22050, fft_size=512)
    x = np.zeros(1000, np.float32)
    with pytest.raises(S.InvalidArgument) as e:
        S.mfcc(sc, mc, x, n_mfcc=41)
    assert str(e.value) == "mfcc: cannot keep 41 cepstral coefficients of 40 mel bands (n_mfcc must lie in [1, n_mels])"
    with pytest.raises(S.InvalidArgument) as e:
        S.mfcc(sc, mc, x, lifter=-1.0)
    assert str(e.value) == "mfcc: cannot lifter with a coefficient of -1 (lifter must be finite and non-negative)"
    with pytest.raises(S.InvalidArgument) as e:
        S.mfcc(Stft.Config.create(fft_size=256, hop=64), mc, x)
    assert str(e.value).startswith("mfcc: cannot project a 256-point STFT through a filterbank built for an FFT of size 512")
    assert S.mfcc(sc, mc, np.zeros((0, 1000), np.float32)).shape == (0, 20, Stft.frames(sc, 1000))


# ---- spectral-shape features ------------------------------------------------------------

SPECTRAL_FILES = ["spectral_centroid", "spectral_bandwidth", "spectral_rolloff", "spectral_flatness"]


@pytest.mark.parametrize("stem", SPECTRAL_FILES)
def test_spectral_goldens(stem):
    """librosa-0.11 vectors of the reference (spectral_goldens.ml) on the HIP path, the reference's tolerances:
    float64 1e-9 / 1e-12, float32 1e-6 / 1e-7.  The end-to-end "signal" cases take the magnitude spectrogram from
    this library's own STFT (float64 interior for the float32 cases: the reference's contract)."""
    from test_oracle_goldens import run_spectral, spectral_golden_freqs
    for case in load_golden("spectral", stem)["cases"]:
        p = case["params"]
        f32 = p["dtype"] == "float32"
        if p["source"] == "spectrogram":
            v = np.abs(O.lcg_signal(int(np.prod(p["shape_s"])), 20261024))
            s = (v * v if p["squared"] else v).reshape(p["shape_s"])
            s = s.astype(np.float32) if f32 else s
        else:
            x = O.lcg_signal(p["length"], 20261024)
            S.set_interior("float64")
            s = Stft.power_spectrum(Stft.Config.create(fft_size=p["fft_size"], hop=p["hop"]),
                                    x.astype(np.float32) if f32 else x, power=1.0)
            S.set_interior("float32")
        got = run_spectral(S, stem, p, s, spectral_golden_freqs(p, s))
        assert got.dtype == (np.float32 if f32 else np.float64)
        check_close(got, case["values"], shape=case["shape"], rtol=F32_RTOL if f32 else F64_RTOL,
                    atol=F32_ATOL if f32 else F64_ATOL, msg=case["name"])


@pytest.mark.parametrize("shape", [(1025, 938), (3, 513, 130), (2, 2, 40, 65), (2, 1)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spectral_vs_oracle(shape, dtype):
    """Every feature against the oracle on seeded magnitudes, at the C2 spectrogram shape among others (bins split
    over four waves, frame tiles ragged), with silent frames, the FFT grid and a custom one, a reused centroid,
    odd exponents; device-resident float32 input gives the host result bit for bit."""
    import torch
    rng = np.random.default_rng(sum(shape))
    s = np.abs(rng.standard_normal(shape)).astype(dtype)
    s[..., :, 0] = 0.0                                        # a silent frame: the underflow guards
    rtol, atol = (2e-6, 1e-7) if dtype == np.float32 else (1e-11, 1e-13)
    bins = shape[-2]
    custom = np.cumsum(rng.uniform(1.0, 50.0, size=bins)).astype(dtype)
    cases = [("centroid", lambda M, a: M.spectral_centroid(a, sample_rate=22050)),
             ("centroid/custom", lambda M, a: M.spectral_centroid(a, sample_rate=8000, freqs=custom)),
             ("bandwidth", lambda M, a: M.spectral_bandwidth(a, sample_rate=22050)),
             ("bandwidth/p3", lambda M, a: M.spectral_bandwidth(a, sample_rate=22050, p=3.0, freqs=custom)),
             ("bandwidth/p1", lambda M, a: M.spectral_bandwidth(a, sample_rate=44100, p=1.0)),
             ("rolloff", lambda M, a: M.spectral_rolloff(a, sample_rate=22050)),
             ("rolloff/10", lambda M, a: M.spectral_rolloff(a, sample_rate=22050, roll_percent=0.1, freqs=custom)),
             ("flatness", lambda M, a: M.spectral_flatness(a)),
             ("flatness/p1.5", lambda M, a: M.spectral_flatness(a, amin=1e-3, power=1.5))]
    for name, fn in cases:
        want = fn(O, s)
        got = fn(S, s)
        assert got.dtype == dtype and got.shape == want.shape == shape[:-2] + (1, shape[-1]), name
        if name.startswith("rolloff"):
            # a bin frequency: exact unless the running sum lands within rounding of the threshold
            assert np.mean(got == want) >= 0.999, name
        else:
            np.testing.assert_allclose(got, want, rtol=rtol, atol=atol * float(np.max(want)), err_msg=name)
        if dtype == np.float32:
            gd = fn(S, torch.from_numpy(s).cuda())
            assert gd.is_cuda and np.array_equal(gd.cpu().numpy(), got), name
    c = S.spectral_centroid(s, sample_rate=22050)
    got = S.spectral_bandwidth(s, sample_rate=22050, centroid=c)
    np.testing.assert_allclose(got, O.spectral_bandwidth(s, sample_rate=22050, centroid=c), rtol=rtol,
                               atol=atol * float(np.max(got)))


def test_memoryless_stages_are_the_flat_functions():
    """Mel.stage and the spectral *_stage constructors (mel.ml:233, spectral.ml:257-284): chunk by chunk they give
    what the flat function gives on the chunk."""
    rng = np.random.default_rng(3)
    s = np.abs(rng.standard_normal((2, 257, 40))).astype(np.float32)
    mc = Mel.Config.create(n_mels=20, sample_rate=16000, fft_size=512)
    for st, fn in ((Mel.stage(mc), lambda a: Mel.apply(mc, a)),
                   (S.spectral_centroid_stage(sample_rate=16000), lambda a: S.spectral_centroid(a, sample_rate=16000)),
                   (S.spectral_bandwidth_stage(sample_rate=16000, p=3.0), lambda a: S.spectral_bandwidth(a, sample_rate=16000, p=3.0)),
                   (S.spectral_rolloff_stage(sample_rate=16000, roll_percent=0.5), lambda a: S.spectral_rolloff(a, sample_rate=16000, roll_percent=0.5)),
                   (S.spectral_flatness_stage(amin=1e-6), lambda a: S.spectral_flatness(a, amin=1e-6))):
        for lo, hi in ((0, 13), (13, 40)):
            assert np.array_equal(st.step(s[..., lo:hi]), fn(s[..., lo:hi]))
        assert st.flush() == []


def test_spectral_messages_and_edges():
    s = np.abs(np.random.default_rng(1).standard_normal((9, 12))).astype(np.float32)
    for fn, op in ((lambda a: S.spectral_centroid(a, sample_rate=22050), "spectral_centroid"),
                   (lambda a: S.spectral_bandwidth(a, sample_rate=22050), "spectral_bandwidth"),
                   (lambda a: S.spectral_rolloff(a, sample_rate=22050), "spectral_rolloff"),
                   (lambda a: S.spectral_flatness(a), "spectral_flatness")):
        for poison in (-1e-3, float("nan")):
            t = s.copy()
            t[4, 7] = poison
            with pytest.raises(S.InvalidArgument) as e:
                fn(t)
            assert str(e.value) == ("%s: cannot analyse a spectrogram with negative or NaN values (a magnitude "
                                    "spectrogram is non-negative)" % op)
        with pytest.raises(S.InvalidArgument) as e:
            fn(np.zeros(5, np.float32))
        assert str(e.value) == "%s: cannot analyse a rank-1 tensor (a spectrogram is [...; bins; frames])" % op
        for shape in ((0, 9, 12), (9, 0), (2, 0, 5)):           # nothing to reduce: zeros of the feature shape
            if shape == (2, 0, 5) and op != "spectral_flatness":
                continue                                         # zero bins cannot imply a grid (checked below)
            out = fn(np.zeros(shape, np.float32))
            assert out.shape == shape[:-2] + (1, shape[-1]) and not out.any()
    with pytest.raises(S.InvalidArgument) as e:
        S.spectral_centroid(np.zeros((1, 4), np.float32), sample_rate=22050)
    assert str(e.value) == ("spectral_centroid: cannot derive bin frequencies for a 1-bin spectrogram (the implied FFT "
                            "size is 0; pass freqs explicitly)")
    assert S.spectral_centroid(np.ones((1, 4), np.float32), sample_rate=22050, freqs=np.array([7.0])).tolist() == [[7.0] * 4]
    # an all-zero frame: centroid 0, bandwidth 0, roll-off at the first bin, flatness 1 (amin / amin)
    z = np.zeros((5, 3), np.float64)
    assert not S.spectral_centroid(z, sample_rate=100).any() and not S.spectral_bandwidth(z, sample_rate=100).any()
    assert not S.spectral_rolloff(z, sample_rate=100).any()
    np.testing.assert_allclose(S.spectral_flatness(z), np.ones((1, 3)), rtol=1e-12)


# ---- chroma over the linear-frequency spectrum -------------------------------------------------------------

def test_chroma_stft_goldens():
    """librosa.feature.chroma_stft vectors (chroma_goldens.ml:165-196) through Soundml.chroma_stft on the HIP path:
    closed-form tolerance for float64, the shared float32 one with the float64 STFT interior (the reference's
    contract) and north_star's 1e-5 with the float32 interior."""
    from test_oracle_goldens import CHROMA_NORMS, chroma_golden_config
    for case in load_golden("chroma", "chroma_stft")["cases"]:
        p = case["params"]
        sc = Stft.Config.create(fft_size=p["fft_size"], hop=p["hop"], pad=("constant", 0.0))
        cc = chroma_golden_config(lambda sr, fft, **kw: S.Chroma.Config.create(sr, fft, **kw), p)
        x = O.harmonic_signal(p["length"], p["sample_rate"])
        peak = float(np.max(np.abs(case["values"])))
        if p["dtype"] == "float64":
            got = S.chroma_stft(sc, cc, x, power=p["power"], norm=CHROMA_NORMS[p["norm"]])
            assert got.dtype == np.float64
            check_close(got, case["values"], shape=case["shape"], rtol=1e-9, atol=1e-12 * peak, msg=case["name"])
        else:
            S.set_interior("float64")
            got = S.chroma_stft(sc, cc, x.astype(np.float32), power=p["power"], norm=CHROMA_NORMS[p["norm"]])
            S.set_interior("float32")
            assert got.dtype == np.float32
            check_close(got, case["values"], shape=case["shape"], rtol=F32_RTOL, atol=F32_ATOL, msg=case["name"])
            fast = S.chroma_stft(sc, cc, x.astype(np.float32), power=p["power"], norm=CHROMA_NORMS[p["norm"]])
            check_fast(fast, np.asarray(case["values"]).reshape(case["shape"]), case["name"] + " float32 interior")


@pytest.mark.parametrize("n_chroma,norm", [(12, "inf"), (12, None), (24, 2.0), (13, 1.0), (36, 3.0), (5, "inf")])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_chroma_apply_vs_oracle(n_chroma, norm, dtype):
    """Chroma.apply against the oracle: one, two and three row chunks, every norm, a silent frame (length below
    the smallest normal: left alone), leading axes; device-resident input gives the host result."""
    import torch
    rng = np.random.default_rng(n_chroma)
    cc = S.Chroma.Config.create(22050, 2048, n_chroma=n_chroma, tuning=0.1)
    oc = O.chroma_config(22050, 2048, n_chroma=n_chroma, tuning=0.1)
    s = (rng.standard_normal((2, 3, 1025, 70)) ** 2).astype(dtype)
    s[0, 1, :, 5] = 0.0
    s[1, 2, :, 9] = 1e-30 if dtype == np.float32 else 1e-300   # projects below the smallest normal
    got = S.Chroma.apply(cc, s, norm=norm)
    want = O.chroma_apply(oc, s, norm=norm)
    assert got.dtype == dtype and got.shape == want.shape == (2, 3, n_chroma, 70)
    rtol = 2e-6 if dtype == np.float32 else 1e-11
    np.testing.assert_allclose(got, want, rtol=rtol, atol=rtol * float(np.max(np.abs(want))))
    assert not got[0, 1, :, 5].any()
    if dtype == np.float32:
        gd = S.Chroma.apply(cc, torch.from_numpy(s).cuda(), norm=norm)
        assert gd.is_cuda and np.array_equal(gd.cpu().numpy(), got)


def test_chroma_stft_vs_oracle_c2_geometry():
    """Soundml.chroma_stft at the C2 geometry (fft 2048 hop 512 reflect, fused power kernel underneath) on a small
    batch, host and device-resident."""
    import torch
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, size=(3, 40000)).astype(np.float32)
    sc = Stft.Config.create(fft_size=2048, hop=512)
    cc = S.Chroma.Config.create(48000, 2048)
    want = O.chroma_stft(O.stft_config(2048, hop=512), O.chroma_config(48000, 2048), x)
    got = S.chroma_stft(sc, cc, x)
    assert got.shape == want.shape == (3, 12, 79) and got.dtype == np.float32
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)            # normalised to a peak of 1 per frame
    gd = S.chroma_stft(sc, cc, torch.from_numpy(x).cuda())
    assert gd.is_cuda and np.array_equal(gd.cpu().numpy(), got)
    assert S.chroma_stft(sc, cc, np.zeros((0, 4000), np.float32)).shape == (0, 12, Stft.frames(sc, 4000))


def test_chroma_messages():
    cc = S.Chroma.Config.create(22050, 512)
    with pytest.raises(S.InvalidArgument) as e:
        S.Chroma.apply(cc, np.zeros((100, 4), np.float32))
    assert str(e.value) == ("apply: cannot project 100 frequency bins through a matrix built for an FFT of size 512 "
                            "(257 bins)")
    with pytest.raises(S.InvalidArgument) as e:
        S.Chroma.apply(cc, np.zeros((257, 4), np.float32), norm=-2.0)
    assert str(e.value) == "apply: cannot normalise in the -2-norm (the exponent must be finite and positive)"
    with pytest.raises(S.InvalidArgument) as e:
        S.Chroma.apply(cc, np.zeros(7, np.float32))
    assert str(e.value) == "apply: cannot project a rank-1 tensor (the projection needs [...; bins; frames])"
    with pytest.raises(S.InvalidArgument) as e:
        S.chroma_stft(Stft.Config.create(fft_size=256, hop=64), cc, np.zeros(1000, np.float32))
    assert str(e.value).startswith("chroma_stft: cannot project a 256-point STFT through a filterbank built for an FFT of size 512")
    assert S.Chroma.apply(cc, np.zeros((2, 257, 0), np.float64)).shape == (2, 12, 0)


# ---- FIR ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("taps,n,ch", [(63, 5000, 2), (1, 100, 1), (8192, 60000, 2), (1000, 1, 1), (257, 16384 * 3, 3),
                                       (16384, 70001, 1), (4097, 49153, 2), (2048, 8191, 1), (600, 3001, 2)])
def test_fir_vs_oracle(taps, n, ch):
    rng = np.random.default_rng(taps + n)
    h = Fir.design_lowpass(taps, 0.25, 80.0)
    x = rng.uniform(-1, 1, size=(ch, n)).astype(np.float32)
    got = Fir.apply(Fir.Plan.create(h), x)
    want = O.fir_filter(h, x)
    assert got.shape == want.shape and got.dtype == np.float32
    # FFT convolution error scales with the filter's L1 gain times the input peak, not with the
    # local output value: |err| <= 1e-5 * sum|h| * max|x|  (FIR path is "parity unpinned")
    bound = 1e-5 * np.sum(np.abs(h)) * np.max(np.abs(x))
    assert np.max(np.abs(got.astype(np.float64) - want)) <= bound


def test_fir_known_answers():
    h = Fir.design_lowpass(255, 0.3, 70.0)
    p = Fir.Plan.create(h)
    n = 40000
    imp = np.zeros((1, n), np.float32)
    imp[0, 0] = 1.0
    y = Fir.apply(p, imp)
    np.testing.assert_allclose(y[0, :255], h, rtol=0, atol=2e-7)        # impulse -> taps
    np.testing.assert_allclose(y[0, 255:], 0, rtol=0, atol=2e-7)
    step = np.ones((1, n), np.float32)
    ys = Fir.apply(p, step)
    np.testing.assert_allclose(ys[0, 254:], 1.0, rtol=0, atol=2e-6)     # unit DC gain after the transient
    np.testing.assert_allclose(ys[0, :255], np.cumsum(h), rtol=0, atol=2e-6)


# ---- Resample: the overlap-save executor's pieces ----------------------------------------------------

@pytest.mark.parametrize("l,m", [(1, 1), (2, 1), (3, 1), (4, 1), (1, 2), (1, 3), (1, 4)])
def test_resample_shape_is_the_reference_stub_bit_for_bit(l, m):
    """smx_resample_shape_c128 against the restated `soundml_resample_shape_run` (resample_stubs.c:329-372): float64,
    same operations in the same order -> identical bits; lines are independent (a stack == its lines)."""
    from soundml_amd import Resample
    from oracle import c_oracle
    rng = np.random.default_rng(7 * l + m)
    n = 96 if m == 3 else 2048
    k = 5
    proto = rng.uniform(-1, 1, size=2 * k * l + 1)
    oh = O.ols_plan_spectrum(proto, n, l, m)
    spec = np.fft.rfft(rng.uniform(-1, 1, size=(9, n)), axis=-1)
    got = Resample.shape(spec, oh, n, l, m)
    assert np.array_equal(got, c_oracle.resample_shape(spec, oh, n, l, m))
    assert np.array_equal(got, O.ols_shape(spec, oh, n, l, m))
    # the reference's own compiled `soundml_resample_shape_run` (oracle/_ref: built from the reference's source where it lies,
    # the .so travels to the GPU box) is the checker proper
    assert c_oracle.have_ref()
    assert np.array_equal(got, c_oracle.ref_resample_shape(spec, oh, n, l, m))
    assert np.array_equal(Resample.shape(spec[3:4], oh, n, l, m), got[3:4])


@pytest.mark.parametrize("l,m,k,n", [(1, 1, 40, 50000), (2, 1, 160, 30001), (1, 2, 161, 60000), (1, 3, 100, 70001), (4, 1, 37, 15000),
                                     (1, 4, 101, 90000), (3, 1, 50, 20000), (3, 2, 60, 40000), (1, 1, 40, 10), (2, 1, 40, 1),
                                     (2, 1, 4095, 50000)])
def test_resample_stage_vs_oracle(l, m, k, n):
    """One stage on the device against its definition (the direct polyphase sum) AND against the restated overlap-save
    executor on the reference's own block grid (ols_run + drain): ceil(n L / M) outputs.  float32 interior: the error of
    an FFT convolution scales with the prototype's L1 gain (Resample is pinned by dB thresholds only in the reference,
    test/resample/resample_quality.ml -- "parity unpinned")."""
    from soundml_amd import Resample
    rng = np.random.default_rng(k + n)
    proto = Resample.prototype(l, k, 0.45 / max(l, m), O.kaiser_beta(100.0))
    x = rng.uniform(-1, 1, size=(2, n)).astype(np.float32)
    st = Resample.Stage.create(proto, l, m, k)
    got = Resample.Stage.apply(st, x)
    want = O.resample_stage_direct(proto, l, m, k, x.astype(np.float64))
    assert got.shape == want.shape == (2, -(-n * l // m)) and got.dtype == np.float32
    bound = 1e-5 * np.sum(np.abs(proto))
    assert np.max(np.abs(got.astype(np.float64) - want)) <= bound
    geom = O.ols_geom(10 ** 9, l, m, k) if (l == 1 or m == 1) and max(l, m) <= 4 else None
    if geom is not None and n > 1000:
        exe = O.ols_stage(proto, l, m, k, geom, x.astype(np.float64))
        assert np.max(np.abs(got.astype(np.float64) - exe)) <= bound
    import torch
    xd = torch.from_numpy(x).cuda()
    assert np.array_equal(Resample.Stage.apply(st, xd).cpu().numpy(), got)       # device path == host path


# ---- BASELINE sizes: size-independent properties -------------------------------------------------------

def test_c2_full_size_properties():
    """C2: 256 x 10 s x 48 kHz.  Checked without a full-size oracle: (a) Parseval per frame
    for interior frames of a rectangular window, (b) linearity of the complex transform,
    (c) oracle parity on the first/last two frames and an interior tile of 3 clips."""
    import torch
    torch.manual_seed(42)
    lead, n = 256, 480000
    x = torch.rand(lead, n, device="cuda") * 2 - 1
    c = Stft.Config.create(fft_size=2048, hop=512)
    p = Stft.power_spectrum(c, x)
    total = Stft.frames(c, n)
    assert tuple(p.shape) == (lead, 1025, total) and total == 938
    assert torch.isfinite(p).all()
    o = O.stft_config(2048, hop=512)
    for clip in (0, 100, 255):
        xc = x[clip].cpu().numpy()
        for a, b in ((0, 2), (total - 2, total), (400, 416)):
            want = np.abs(O.transform_range(o, xc, a, b, np.complex128)) ** 2
            check_fast(p[clip, :, a:b].cpu().numpy(), want, "C2 clip %d frames %d:%d" % (clip, a, b))
    # Parseval with a rectangular window: sum_k c_k |X_k|^2 = N * sum x^2 for interior frames
    cr = Stft.Config.create(fft_size=2048, hop=512, window="rectangular")
    pr = Stft.power_range(cr, x[:8], 2, 34)
    wts = torch.full((1025,), 2.0, device="cuda")
    wts[0] = wts[1024] = 1.0
    lhs = (pr * wts[None, :, None]).sum(dim=1)
    fr = x[:8].unfold(-1, 2048, 512)[:, 0:32, :]
    rhs = 2048.0 * (fr.double() ** 2).sum(dim=-1)
    assert torch.allclose(lhs.double(), rhs, rtol=2e-5)


# ---- randomized cross-check of the fused kernels against the generic ones -------------------------------

@pytest.mark.timeout(600)
def test_fused_kernels_match_generic_on_random_geometries(monkeypatch):
    """The fused fft-2048 kernels (counter-synchronised persistent workgroups, border frames folded in, XCD tile
    order) against the library's own generic kernels (SMX_DISABLE_FAST) over random batch sizes, lengths, hops,
    alignments and paddings; each fused result is also bit-for-bit reproducible."""
    import subprocess, sys, json, os, tempfile
    rng = np.random.default_rng(20261003)
    cases = []
    for _ in range(24):
        cases.append(dict(lead=int(rng.choice([1, 2, 3, 5, 17, 64, 300])), n=int(rng.integers(1, 70000)),
                          hop=int(rng.choice([512, 512, 512, 300, 77, 1024, 2048, 3000, 511])),
                          alignment=str(rng.choice(["centered", "left", "right"])),
                          pad=str(rng.choice(["reflect", "edge", "constant"])), power=float(rng.choice([2.0, 2.0, 1.0, 0.7]))))
    def run(disable_fast):
        # the switch is read once per process: run each side in its own interpreter
        code = """
import json, sys, numpy as np
sys.path.insert(0, %r)
import soundml_amd as S
from soundml_amd import Stft
cases = json.loads(sys.argv[1])
out = {}
for i, c in enumerate(cases):
    rng = np.random.default_rng(1000 + i)
    x = rng.uniform(-1, 1, size=(c['lead'], c['n'])).astype(np.float32)
    pad = (c['pad'], 0.25) if c['pad'] == 'constant' else c['pad']
    cfg = Stft.Config.create(fft_size=2048, hop=c['hop'], alignment=c['alignment'], pad=pad)
    a = Stft.power_spectrum(cfg, x, c['power'])
    b = Stft.power_spectrum(cfg, x, c['power'])
    assert np.array_equal(a, b), ('not reproducible', c)
    z = Stft.transform(cfg, x)
    out['p%%d' %% i] = a
    out['z%%d' %% i] = z
np.savez(sys.argv[2], **out)
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ)
        if disable_fast:
            env["SMX_DISABLE_FAST"] = "1"
        else:
            env.pop("SMX_DISABLE_FAST", None)
        path = tempfile.mktemp(suffix=".npz")
        subprocess.run([sys.executable, "-c", code, json.dumps(cases), path], check=True, env=env, timeout=500)
        data = dict(np.load(path))
        os.remove(path)
        return data
    fused, generic = run(False), run(True)
    for i, c in enumerate(cases):
        check_fast(fused["p%d" % i], generic["p%d" % i], "power %s" % c)
        check_fast(fused["z%d" % i].real, generic["z%d" % i].real, "re %s" % c)
        check_fast(fused["z%d" % i].imag, generic["z%d" % i].imag, "im %s" % c)


def test_c_abi_from_plain_c(tmp_path):
    """The boundary is a C ABI: examples/power_spectrum.c (no Python, no torch) compiled with gcc against the header and
    the shared library finds the 440 Hz tone in bin 10 of a fft-1024 power spectrogram and computes its mel spectrogram."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "power_spectrum")
    libdir = os.path.join(root, "soundml_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "power_spectrum.c"), "-L", libdir, "-lsoundml_amd",
                           "-Wl,-rpath," + libdir, "-lm", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "peak bin 10" in out.stdout



def test_staged_transfers_of_large_host_tensors():
    """Host tensors above 4 MB cross in 16 MB chunks through pinned staging buffers (transfer.cpp): several chunks, a
    ragged last one, both directions — the host entry point returns exactly what the device-resident one does."""
    import torch
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, size=(37, 300007)).astype(np.float32)         # 44.4 MB up, 91 MB / 182 MB down
    c = Stft.Config.create(fft_size=2048, hop=512)
    xd = torch.from_numpy(x).cuda()
    p = Stft.power_spectrum(c, x)
    assert np.array_equal(p, Stft.power_spectrum(c, xd).cpu().numpy())
    z = Stft.transform(c, x)
    assert np.array_equal(z, Stft.transform(c, xd).cpu().numpy())
    back = Stft.invert(c, z)
    assert np.array_equal(back, Stft.invert(c, torch.from_numpy(z).cuda()).cpu().numpy())
    np.testing.assert_allclose(back, x[:, :back.shape[-1]], rtol=0, atol=5e-6)

"""TEST INFRASTRUCTURE (checker only; never imported by the product): the decibel ruler the reference pins its
resampler with, restated in numpy -- soundml/test/resample/resample_quality.ml (its header: "spec assertions in
decibels, never bit comparisons"; the metric definitions there mirror dev/soxr_reference.py "line for line").
The reference holds NO sample vector for this arithmetic: these metrics and the measured soxr edge of
soundml/test/resample/vectors/soxr_reference.json (copied as data to tests/golden/resample/) are its whole pin.

Also here: the single-stage plan of `Resample.Config.create` (resample.ml:113-116, 519-526, 919-932) for the pure
xL / /M conversions, so that a test designs the stage the reference would run.
"""
import math
from functools import lru_cache

import numpy as np

from . import soundml_oracle as O

TRIM_FRACTION = 0.15            # resample_quality.ml:73
FUNDAMENTAL_HALF_WIDTH = 16     # resample_quality.ml:75
QUALITY = {"fast": (100.0, 0.913), "high": (126.0, 0.913), "best": (175.0, 0.913)}   # resample.ml:519-526


# ---- the plan of a pure xL or /M conversion (resample.ml:113-116, 919-932) ------------------------------------------

def kaiser_numtaps(att: float, width: float) -> float:
    """resample.ml:113-116: kaiserord tap count for a transition `width` (Nyquist = 1), made odd."""
    n = math.ceil((att - 7.95) / 2.285 / (math.pi * width) + 1.0)
    return n + 1.0 if math.fmod(n, 2.0) == 0.0 else n


def single_stage(l: int, m: int, quality: str = "high"):
    """resample.ml:919-932: (K, fc, beta) of the one stage of an L/M conversion: transition (1 - passband) / max(L, M),
    K = ceil((taps - 1) / (2 L)) (at least 1), cutoff mid-transition, Kaiser beta from the attenuation."""
    att, passband = QUALITY[quality]
    width = (1.0 - passband) / float(max(l, m))
    ntaps = kaiser_numtaps(att, width)
    k = max(1, int(math.ceil((ntaps - 1.0) / (2.0 * float(l)))))
    fc = (1.0 + passband) / (2.0 * float(max(l, m)))
    return k, fc, O.kaiser_beta(att)


# ---- signals (resample_quality.ml:58-67) -----------------------------------------------------------------------------

def tone(sr: int, f: float, seconds: float) -> np.ndarray:
    n = int(round(sr * seconds))
    return np.sin(2.0 * math.pi * f * np.arange(n, dtype=np.float64) / float(sr))


# ---- analysis (resample_quality.ml:69-152) ---------------------------------------------------------------------------

@lru_cache(maxsize=16)
def kaiser(beta: float, n: int) -> np.ndarray:
    """`Window.make (Kaiser beta) n` (periodic, window.ml:296-318) with numpy's I0: equal to the oracle's series form to
    1e-13 (tests/test_resample_metrics.py checks it at a small n); only decibels are read off it."""
    m = n + 1
    alpha = (m - 1) / 2.0
    r = (np.arange(n, dtype=np.float64) - alpha) / alpha
    return np.i0(beta * np.sqrt(np.maximum(0.0, 1.0 - r * r))) / np.i0(beta)


def interior(x: np.ndarray) -> np.ndarray:
    """resample_quality.ml:86-92: the 15 %-trimmed interior cropped to a power of two."""
    n = x.shape[0]
    i0 = int(float(n) * TRIM_FRACTION)
    cut = x[i0:n - i0]
    p = 1
    while p * 2 <= cut.shape[0]:
        p *= 2
    return cut[:p]


def spectrum(x: np.ndarray) -> np.ndarray:
    """resample_quality.ml:94-101: magnitudes of the Kaiser(30)-windowed interior."""
    cut = interior(np.asarray(x, dtype=np.float64))
    return np.abs(np.fft.rfft(kaiser(30.0, cut.shape[0]) * cut))


def _fundamental(mags):
    p = int(np.argmax(mags))    # first maximum, as peak_index (:103-106)
    return p, max(0, p - FUNDAMENTAL_HALF_WIDTH), min(mags.shape[0], p + FUNDAMENTAL_HALF_WIDTH + 1)


def sfdr(mags: np.ndarray) -> float:
    """resample_quality.ml:108-117: fundamental over the largest bin outside its +-16-bin skirt (bins 0, 1 ignored), dB."""
    p, lo, hi = _fundamental(mags)
    mask = np.ones(mags.shape[0], dtype=bool)
    mask[:2] = False
    mask[lo:hi] = False
    return 20.0 * math.log10(mags[p] / float(mags[mask].max()))


def thdn(mags: np.ndarray) -> float:
    """resample_quality.ml:119-126: energy outside the fundamental's skirt over the energy inside it, dB."""
    p, lo, hi = _fundamental(mags)
    e = mags.astype(np.float64) ** 2
    fund = float(e[lo:hi].sum())
    rest = float(e[:lo].sum()) + float(e[hi:].sum())   # summed apart, as the reference does (the two differ by 250 dB)
    return 10.0 * math.log10(rest / fund)


def amp_at(sr: int, f: float, x: np.ndarray) -> float:
    """resample_quality.ml:128-144: amplitude of the f-hertz component over the trimmed interior by windowed projection
    at the exact frequency."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    i0 = int(float(n) * TRIM_FRACTION)
    ln = n - 2 * i0
    w = kaiser(30.0, ln)
    ph = 2.0 * math.pi * f * np.arange(i0, i0 + ln, dtype=np.float64) / float(sr)
    v = w * x[i0:i0 + ln]
    return 2.0 * math.hypot(float(np.sum(v * np.cos(ph))), float(np.sum(v * np.sin(ph)))) / float(np.sum(w))


def peak_dbfs(x: np.ndarray) -> float:
    """resample_quality.ml:146-154: largest magnitude over the trimmed interior, dBFS."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    i0 = int(float(n) * TRIM_FRACTION)
    peak = float(np.max(np.abs(x[i0:n - i0]))) if n - 2 * i0 > 0 else 0.0
    return 20.0 * math.log10(max(peak, np.finfo(np.float64).tiny))


def measured_edge(convert, sr: int, target: int) -> float:
    """resample_quality.ml:262-277: the -3 dB frequency of `convert` (tone at sr -> signal at target) by 40 bisections
    between 0.85 and 0.9995 of the lower Nyquist."""
    goal = 1.0 / math.sqrt(2.0)

    def gain(f):
        return amp_at(target, f, convert(tone(sr, f, 1.0)))
    nyq = float(min(sr, target)) / 2.0
    lo, hi = 0.85 * nyq, 0.9995 * nyq
    if not (gain(lo) > goal > gain(hi)):
        raise AssertionError("edge bisection unbracketed for %d->%d" % (sr, target))
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        if gain(mid) > goal:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def sweep_worst(convert, sr: int, target: int, seconds: float = 30.0, nfft: int = 8192, hop: int = 4096) -> float:
    """resample_quality.ml:294-352: worst alias of a linear sweep 1 kHz .. 0.9 of the output Nyquist: per Kaiser(16)
    frame of the output, the largest bin outside a guard band around the instantaneous frequency over the largest one
    inside it, dB; the first and last 0.2 s are skipped."""
    nyq_out = float(target) / 2.0
    f0, f1 = 1000.0, 0.9 * nyq_out
    rate = (f1 - f0) / seconds
    t = np.arange(int(float(sr) * seconds), dtype=np.float64) / float(sr)
    y = np.asarray(convert(np.sin(2.0 * math.pi * (f0 * t + 0.5 * rate * t * t))), dtype=np.float64)
    w = kaiser(16.0, nfft)
    guard = 3.0 * rate * (float(nfft) / float(target)) + 400.0
    skip = int(0.2 * float(target))
    bin_hz = float(target) / float(nfft)
    freqs = np.arange(nfft // 2 + 1, dtype=np.float64) * bin_hz
    worst = -math.inf
    start = skip
    while start + nfft <= y.shape[0] - skip:
        f_inst = f0 + rate * (float(start + nfft // 2) / float(target))
        mags = np.abs(np.fft.rfft(w * y[start:start + nfft]))
        inside = np.abs(freqs - f_inst) <= guard
        inside[:3] = False
        outside = ~inside
        outside[:3] = False
        peak = float(mags[inside].max()) if inside.any() else 0.0
        alias = float(mags[outside].max()) if outside.any() else 0.0
        worst = max(worst, 20.0 * math.log10(alias / peak))
        start += hop
    return worst


# ---- the stage by its definition, polyphase form (float64; what the thresholds are asserted on for the oracle itself) ----

def stage_polyphase(proto, l: int, m: int, k: int, x: np.ndarray) -> np.ndarray:
    """y[i] = sum_t proto[t] xu[i M + K L - t], xu[q L] = x[q] (resample.ml:1318-1326): only the taps that meet a
    nonzero of the zero-stuffed input are visited (phase (i M + K L) mod L, 2 K + 1 of them) -- the same sum as
    soundml_oracle.resample_stage_direct, which convolves the stuffed signal and is O(n L taps); ceil(n L / M) outputs."""
    x = np.asarray(x, dtype=np.float64)
    proto = np.asarray(proto, dtype=np.float64)
    n = x.shape[0]
    n_out = -(-n * l // m)
    taps = proto.shape[0]
    s = np.arange(n_out, dtype=np.int64) * m + k * l           # position in the stuffed signal
    p, q = s % l, s // l                                       # proto index p + j L meets x[q - j]
    jmax = (taps - 1) // l + 1
    bank = np.zeros((l, jmax), dtype=np.float64)
    for ph in range(l):
        col = proto[ph::l]
        bank[ph, :col.shape[0]] = col
    pad = jmax + 1
    xp = np.concatenate([np.zeros(pad), x, np.zeros(pad + k + 2)])
    out = np.empty(n_out, dtype=np.float64)
    step = max(1, (1 << 22) // jmax)
    j = np.arange(jmax, dtype=np.int64)
    for a in range(0, n_out, step):
        b = min(n_out, a + step)
        idx = q[a:b, None] - j[None, :] + pad
        valid = (idx >= 0) & (idx < xp.shape[0])
        g = np.where(valid, xp[np.clip(idx, 0, xp.shape[0] - 1)], 0.0)
        out[a:b] = np.einsum("ij,ij->i", bank[p[a:b]], g)
    return out

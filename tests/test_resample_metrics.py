"""The reference pins its resampler by decibel metrics only (soundml/test/resample/resample_quality.ml; the measured soxr
edge in vectors/soxr_reference.json is the one committed number).  This file holds the oracle to that ruler on CPU:
the metric restatements (oracle/resample_metrics.py), the single-stage plans of the reference's design formulas
(resample.ml:113-116, 919-932) and the float64 stage arithmetic of the oracle meet the reference's own float64
thresholds -- Q1/Q2 tone SFDR >= 130 dB and THD+N <= -125 dB, Q3 out-of-band residue <= -130 dBFS, Q4 passband within
0.01 dB, Q5 the -3 dB edge within 1 % of measured soxr HQ -- for the x2 / x3 / x4 and /2 / /3 / /4 overlap-save classes
(resample.ml:949-953) and, for Q5, the three conversions the reference measures.  tests/test_gpu_resample_quality.py
holds the device's float32 stage and the C4 filter to the float32 columns of the same ruler."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import resample_metrics as M
from oracle import soundml_oracle as O

CLASSES = [(2, 1, 24000), (3, 1, 16000), (4, 1, 12000), (1, 2, 48000), (1, 3, 48000), (1, 4, 48000)]


def soxr(name):
    data = json.load(open(os.path.join(ROOT, "tests", "golden", "resample", "soxr_reference.json")))
    for c in data["cases"]:
        if c["name"] == name:
            return c["values"][0]
    raise KeyError(name)


def test_kaiser_window_of_the_metrics_is_the_oracles():
    for n in (16, 257, 1024):
        assert np.max(np.abs(M.kaiser(30.0, n) - O.window("kaiser", n, True, 30.0))) < 1e-12


def test_single_stage_plans_follow_the_reference_formulas():
    # resample.ml:113-116 (odd kaiserord length) and :919-932; x2 at `High: 381 taps -> K = 95
    assert M.kaiser_numtaps(126.0, 0.087 / 2) == 381.0
    k, fc, beta = M.single_stage(2, 1, "high")
    assert (k, fc) == (95, (1.0 + 0.913) / 4.0) and abs(beta - 0.1102 * (126.0 - 8.7)) < 1e-15
    for l, m, _ in CLASSES:
        k, fc, beta = M.single_stage(l, m)
        assert 2 * k * l + 1 <= 16384          # what one device block holds (smx_resample_stage_create)


def test_polyphase_form_is_the_direct_definition():
    rng = np.random.default_rng(5)
    for l, m in ((2, 1), (1, 3), (3, 2), (160, 147)):
        k = 4
        proto = O.resample_prototype(l, k, 0.4 / max(l, m), 6.0)
        x = rng.standard_normal(90)
        want = O.resample_stage_direct(proto, l, m, k, x[None, :])[0]
        got = M.stage_polyphase(proto, l, m, k, x)
        assert got.shape == want.shape and np.max(np.abs(got - want)) < 1e-13


@pytest.mark.parametrize("l,m,sr", CLASSES)
def test_oracle_stage_meets_the_float64_thresholds(l, m, sr):
    """Q1/Q2 (resample_quality.ml:163-190, thresholds 130 / -125), Q3 (:206-231, -130 dBFS) and Q4 (:235-262, 0.01 dB)
    on the reference's single-stage `High design, float64."""
    target = sr * l // m
    k, fc, beta = M.single_stage(l, m)
    proto = O.resample_prototype(l, k, fc, beta)
    conv = lambda x: M.stage_polyphase(proto, l, m, k, x)
    nyq = min(sr, target) / 2.0
    for frac in (0.045, 0.23, 0.45, 0.79):                      # Q10's scaled tone positions (:451-463)
        mags = M.spectrum(conv(M.tone(sr, frac * nyq, 2.0)))
        assert M.sfdr(mags) >= 130.0, (frac, M.sfdr(mags))
        assert M.thdn(mags) <= -125.0, (frac, M.thdn(mags))
    for frac in (0.02, 0.5, 0.913):                              # Q4 / Q10 flatness
        f = frac * nyq
        dev = abs(20.0 * np.log10(M.amp_at(target, f, conv(M.tone(sr, f, 1.0)))))
        assert dev <= 0.01, (frac, dev)
    if m > 1:                                                    # Q3: tones above the output Nyquist vanish
        for f in (1.125 * nyq, 1.5 * nyq, min(2.25 * nyq, 0.49 * sr)):
            assert M.peak_dbfs(conv(M.tone(sr, f, 2.0))) <= -130.0, f


@pytest.mark.parametrize("sr,target", [(44100, 48000), (48000, 44100), (44100, 16000)])
def test_passband_edge_within_one_percent_of_measured_soxr(sr, target):
    """Q5 (resample_quality.ml:279-289) on the single-stage plan of the design formulas: the one committed number of
    the reference's resample suite."""
    g = np.gcd(sr, target)
    l, m = target // g, sr // g
    k, fc, beta = M.single_stage(l, m)
    proto = O.resample_prototype(l, k, fc, beta)
    got = M.measured_edge(lambda x: M.stage_polyphase(proto, l, m, k, x), sr, target)
    want = soxr("edge_hq_%d_%d" % (sr, target))
    assert abs(got - want) / want <= 0.01, (got, want)

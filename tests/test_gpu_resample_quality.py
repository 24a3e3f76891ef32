"""The device's float32 FIR arithmetic under the reference's own ruler (soundml/test/resample/resample_quality.ml): the
float32 columns of Q1/Q2 (tone SFDR >= 125 dB, THD+N <= -125 dB), Q3 (out-of-band residue <= -125 dBFS) and Q4 (passband
within 0.02 dB), measured by the metric restatements of oracle/resample_metrics.py on
  * `smx_resample_stage_*` for the x2 / x3 / x4 and /2 / /3 / /4 classes the reference runs by overlap-save
    (resample.ml:949-953), each with the single-stage `High design of resample.ml:919-932, and
  * the BASELINE C4 filter (8192-tap Kaiser lowpass) through `smx_fir_apply_*`.
This is how the reference pins this arithmetic -- it holds no sample vector for it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import resample_metrics as M
from oracle import soundml_oracle as O

import soundml_amd as S
from soundml_amd import Fir, Resample

CLASSES = [(2, 1, 24000), (3, 1, 16000), (4, 1, 12000), (1, 2, 48000), (1, 3, 48000), (1, 4, 48000)]


def f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32)


@pytest.mark.parametrize("l,m,sr", CLASSES)
def test_device_stage_meets_the_float32_thresholds(l, m, sr):
    target = sr * l // m
    k, fc, beta = M.single_stage(l, m)
    proto = Resample.prototype(l, k, fc, beta)
    assert np.array_equal(proto, O.resample_prototype(l, k, fc, beta))      # the design itself: bit for bit the oracle's
    st = Resample.Stage.create(proto, l, m, k)
    conv = lambda x: np.asarray(Resample.Stage.apply(st, f32(x)[None, :]))[0].astype(np.float64)
    nyq = min(sr, target) / 2.0
    report = []
    for frac in (0.045, 0.23, 0.45, 0.79):                      # Q1 / Q2 at Q10's scaled tone positions
        mags = M.spectrum(conv(M.tone(sr, frac * nyq, 2.0)))
        report.append((frac, round(M.sfdr(mags), 1), round(M.thdn(mags), 1)))
    for frac, d, t in report:
        assert d >= 125.0 and t <= -125.0, report
    for frac in (0.02, 0.5, 0.913):                              # Q4
        f = frac * nyq
        dev = abs(20.0 * np.log10(M.amp_at(target, f, conv(M.tone(sr, f, 1.0)))))
        assert dev <= 0.02, (frac, dev)
    if m > 1:                                                    # Q3
        for f in (1.125 * nyq, 1.5 * nyq, min(2.25 * nyq, 0.49 * sr)):
            assert M.peak_dbfs(conv(M.tone(sr, f, 2.0))) <= -125.0, (f, M.peak_dbfs(conv(M.tone(sr, f, 2.0))))


def test_c4_filter_meets_the_float32_thresholds():
    """BASELINE C4: the 8192-tap lowpass (cutoff 0.25 of Nyquist, 100 dB) at 48 kHz: passband tones keep the float32
    SFDR / THD+N of the reference's ruler and their level (Q4's 0.02 dB), stopband tones fall under the design's attenuation."""
    sr = 48000
    h = Fir.design_lowpass(8192, 0.25, 100.0)
    plan = Fir.Plan.create(h)
    conv = lambda x: np.asarray(Fir.apply(plan, f32(x)[None, :]))[0].astype(np.float64)
    report = []
    for f in (300.0, 1000.0, 3000.0, 5000.0):                   # passband: cutoff 6 kHz
        y = conv(M.tone(sr, f, 2.0))
        mags = M.spectrum(y)
        dev = abs(20.0 * np.log10(M.amp_at(sr, f, y)))
        report.append((f, round(M.sfdr(mags), 1), round(M.thdn(mags), 1), round(dev, 5)))
    for f, d, t, dev in report:
        assert d >= 125.0 and t <= -125.0 and dev <= 0.02, report
    for f in (6500.0, 9000.0, 15000.0, 23000.0):                # stopband (the transition of 8192 taps is ~40 Hz wide)
        assert M.peak_dbfs(conv(M.tone(sr, f, 2.0))) <= -95.0, f

"""Measured values behind tests/test_gpu_resample_quality.py: the reference's decibel ruler (resample_quality.ml) on the
device's float32 stages and on the C4 filter, next to the float64 oracle.  Prints one line per stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import resample_metrics as M
from oracle import soundml_oracle as O
from soundml_amd import Fir, Resample
f32 = lambda x: np.asarray(x, dtype=np.float64).astype(np.float32)
for l, m, sr in [(2, 1, 24000), (3, 1, 16000), (4, 1, 12000), (1, 2, 48000), (1, 3, 48000), (1, 4, 48000)]:
    target = sr * l // m
    k, fc, beta = M.single_stage(l, m)
    proto = Resample.prototype(l, k, fc, beta)
    st = Resample.Stage.create(proto, l, m, k)
    dev = lambda x: np.asarray(Resample.Stage.apply(st, f32(x)[None, :]))[0].astype(np.float64)
    ora = lambda x: M.stage_polyphase(proto, l, m, k, x)
    nyq = min(sr, target) / 2.0
    rows = []
    for name, conv in (("device f32", dev), ("oracle f64", ora)):
        sf, th = [], []
        for frac in (0.045, 0.23, 0.45, 0.79):
            mags = M.spectrum(conv(M.tone(sr, frac * nyq, 2.0)))
            sf.append(M.sfdr(mags)); th.append(M.thdn(mags))
        flat = max(abs(20 * np.log10(M.amp_at(target, fr * nyq, conv(M.tone(sr, fr * nyq, 1.0))))) for fr in (0.02, 0.5, 0.913))
        oob = max(M.peak_dbfs(conv(M.tone(sr, f, 2.0))) for f in (1.125 * nyq, 1.5 * nyq)) if m > 1 else float("nan")
        rows.append("%s: SFDR min %.1f dB, THD+N max %.1f dB, flatness %.5f dB, out-of-band %.1f dBFS" % (name, min(sf), max(th), flat, oob))
    print("x%d / %d (%d -> %d Hz, K = %d, %d taps): %s | %s" % (l, m, sr, target, k, 2 * k * l + 1, rows[0], rows[1]))
h = Fir.design_lowpass(8192, 0.25, 100.0)
plan = Fir.Plan.create(h)
conv = lambda x: np.asarray(Fir.apply(plan, f32(x)[None, :]))[0].astype(np.float64)
sf, th = [], []
for f in (300.0, 1000.0, 3000.0, 5000.0):
    mags = M.spectrum(conv(M.tone(48000, f, 2.0)))
    sf.append(M.sfdr(mags)); th.append(M.thdn(mags))
stop = max(M.peak_dbfs(conv(M.tone(48000, f, 2.0))) for f in (6500.0, 9000.0, 15000.0, 23000.0))
print("C4 filter (8192 taps, cutoff 0.25, 100 dB) float32: SFDR min %.1f dB, THD+N max %.1f dB, stopband residue %.1f dBFS" % (min(sf), max(th), stop))

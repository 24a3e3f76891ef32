#!/usr/bin/env python3
"""Generates tests/golden/fir/scipy_pins.json: independent third-party evidence for the FIR half of the oracle.

The reference has no FIR filter module (README "planned"); its only FIR arithmetic is the resample prototype design
(resample.ml:105-163) and the overlap-save executor.  The oracle's `kaiser_beta`, `design_lowpass` and `fir_filter`
are therefore pinned here against scipy.signal (scipy 1.15 in the build container): `kaiser_beta`, `firwin` with a
Kaiser window (the same sinc x I0 window, unit DC gain) and `lfilter` (direct-form float64 convolution).
Inputs are the reference's 31-bit LCG test signal (stft_goldens.ml:19-23).  Run from the repo root.
"""
import json
import os
import sys

import numpy as np
import scipy
import scipy.signal as sig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import soundml_oracle as O   # only for lcg_signal (bit-exact generator)

cases = []
for taps, fc, att in [(63, 0.25, 80.0), (64, 0.5, 40.0), (255, 0.3, 70.0), (8192, 0.25, 100.0), (1025, 0.45, 120.0)]:
    beta = float(sig.kaiser_beta(att))
    h = sig.firwin(taps, fc, window=("kaiser", beta), scale=True)
    x = O.lcg_signal(3 * taps + 777 if taps < 1000 else 20000, seed=20250803 + taps)
    y = sig.lfilter(h, [1.0], x)
    keep = np.unique(np.concatenate([np.arange(0, 48), np.arange(len(y) // 2, len(y) // 2 + 48), np.arange(len(y) - 48, len(y))]))
    cases.append({"taps": taps, "cutoff": fc, "attenuation": att, "beta": beta,
                  "h": [float(v) for v in h] if taps <= 255 else None,
                  "h_sample_index": None if taps <= 255 else [0, 1, 2, taps // 4, taps // 2 - 1, taps // 2, taps - 2, taps - 1],
                  "h_sample": None if taps <= 255 else [float(h[i]) for i in [0, 1, 2, taps // 4, taps // 2 - 1, taps // 2, taps - 2, taps - 1]],
                  "h_sum_abs": float(np.sum(np.abs(h))),
                  "x_seed": 20250803 + taps, "x_len": len(x),
                  "y_index": [int(i) for i in keep], "y": [float(y[i]) for i in keep]})
out = {"generator": "tools/gen_fir_fixtures.py", "scipy": scipy.__version__, "numpy": np.__version__, "cases": cases}
path = os.path.join(ROOT, "tests", "golden", "fir", "scipy_pins.json")
with open(path, "w") as fh:
    json.dump(out, fh)
print(path, os.path.getsize(path))

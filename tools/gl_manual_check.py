"""One Griffin-Lim step by hand (plain Stft.invert / Stft.transform and the unit() step in numpy float32) against griffin_lim's own fused
synthesis, on a kept draw.  python tools/gl_manual_check.py case.npz"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import soundml_oracle as O
from soundml_amd import Stft
d = np.load(sys.argv[1])
mag, init, mom = d["mag"], d["init"], float(d["mom"])
c, o = Stft.Config.create(fft_size=2048, hop=512), O.stft_config(2048, hop=512)
n = O.output_length(o, mag.shape[-1])
ang = (np.cos(init.astype(np.float64)) + 1j * np.sin(init.astype(np.float64))).astype(np.complex64)
def unit(e):
    m = np.hypot(e.real.astype(np.float64), e.imag.astype(np.float64)).astype(np.float32) + np.float32(np.finfo(np.float32).tiny)
    return (e.real / m + 1j * (e.imag / m)).astype(np.complex64)
beta = np.float32(mom / (1.0 + mom))
prev = None
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))
for it in range(1, 5):
    y = Stft.invert(c, (mag * ang).astype(np.complex64), length=n)          # plain synthesis of mag * angles
    gl = Stft.griffin_lim(c, mag, n_iter=it, momentum=mom, init=init) if it > 1 else None
    rebuilt = Stft.transform(c, y)
    e = rebuilt if prev is None else (rebuilt - beta * prev).astype(np.complex64)
    ang = unit(e)
    prev = rebuilt
    final = Stft.invert(c, (mag * ang).astype(np.complex64), length=n)     # what griffin_lim(n_iter = it) returns
    glit = Stft.griffin_lim(c, mag, n_iter=it, momentum=mom, init=init)
    want = O.griffin_lim(o, mag, n_iter=it, momentum=mom, init=init)
    print("n_iter %d: by hand vs griffin_lim %.3e | by hand vs oracle %.3e | griffin_lim vs oracle %.3e" % (it, rel(final, glit), rel(final, want), rel(glit, want)))

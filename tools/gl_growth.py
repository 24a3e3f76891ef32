"""Griffin-Lim at the reference's defaults (momentum 0.99): how the distance to the float64 oracle grows with the iteration count,
for the device (float32 interior; SMX_INVERT_PIPELINE=0 selects the older synthesis kernel) and for the ORACLE ITSELF when what it
stores between the transforms is rounded to float32 (rebuilt spectra to complex64, signals to float32: the least any float32
interior can do).  python tools/gl_growth.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import soundml_oracle as O
from soundml_amd import Stft

def rounded_gl(c, s, n_iter, momentum, init, seed=None):
    """O.griffin_lim with the stored intermediates rounded to float32 (arithmetic still float64)."""
    magnitudes = s.astype(np.float64).astype(np.complex128)
    p = init.astype(np.float64)
    angles = np.cos(p) + 1j * np.sin(p)
    frames_ = s.shape[-1]
    beta = momentum / (1.0 + momentum)
    previous = None
    tiny = float(np.finfo(np.float64).tiny)
    for _ in range(n_iter):
        y = O.synthesise(c, magnitudes * angles).astype(np.float32).astype(np.float64)
        rebuilt = O.transform_range(c, y, 0, frames_, np.complex128).astype(np.complex64).astype(np.complex128)
        extrapolated = rebuilt if previous is None else rebuilt - previous * beta
        angles = extrapolated / (np.abs(extrapolated) + tiny)
        previous = rebuilt
    return O.synthesise(c, magnitudes * angles, None).astype(np.float32)

rng = np.random.default_rng(int(os.environ.get("SEED", "99")))
x = rng.uniform(-1, 1, size=(2, 24000)).astype(np.float32)
c = Stft.Config.create(fft_size=2048, hop=512)
o = O.stft_config(2048, hop=512)
mag = np.abs(Stft.transform(c, x)).astype(np.float32)
phase = rng.uniform(-np.pi, np.pi, size=mag.shape).astype(np.float32)
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
print("iterations: device rel l2 to the float64 oracle | float32-storage oracle rel l2 | oracle moved by 1 ulp of the magnitudes")
for it in (1, 2, 4, 8, 16, 24, 32):
    want = O.griffin_lim(o, mag, it, 0.99, phase, None).astype(np.float64)
    got = Stft.griffin_lim(c, mag, n_iter=it, momentum=0.99, init=phase)
    stored = rounded_gl(o, mag, it, 0.99, phase)
    moved = O.griffin_lim(o, np.nextafter(mag, np.float32(np.inf)), it, 0.99, phase, None)
    print("%2d: %.3e | %.3e | %.3e" % (it, rel(got, want), rel(stored, want), rel(moved, want)))

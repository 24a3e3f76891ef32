"""Stft.invert of a RANDOM (inconsistent) spectrum against the float64 oracle, error by hop, for the shipped kernel and the one of rounds 1-4.
python tools/invert_tail_check.py [frames=46] [length=23040]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import soundml_oracle as O
from soundml_amd import Stft
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 46
length = int(sys.argv[2]) if len(sys.argv) > 2 else 23040
rng = np.random.default_rng(3)
z = (rng.standard_normal((2, 1025, frames)) + 1j * rng.standard_normal((2, 1025, frames))).astype(np.complex64)
c, o = Stft.Config.create(fft_size=2048, hop=512), O.stft_config(2048, hop=512)
want = O.synthesise(o, z.astype(np.complex128), length)
for env in ("1", "0"):
    os.environ["SMX_INVERT_PIPELINE"] = env
    got = Stft.invert(c, z, length=length).astype(np.float64)
    e = np.abs(got - want)
    nb = length // 512
    per = e[0, : nb * 512].reshape(nb, 512).max(axis=1)
    print("pipeline" if env == "1" else "rounds 1-4", "max err %.3e (peak %.2f); by hop:" % (e.max(), np.abs(want).max()), " ".join("%.0e" % v for v in per))
    print("   last 6 samples err:", " ".join("%.2e" % v for v in e[0, -6:]), " want:", " ".join("%.3f" % v for v in want[0, -6:]))

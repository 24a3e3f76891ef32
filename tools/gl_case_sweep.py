"""Griffin-Lim at a fixed geometry over many seeds: relative l2 distance to the float64 oracle (SMX_INVERT_PIPELINE=0: the synthesis kernel of
rounds 1-4).  python tools/gl_case_sweep.py [frames=45] [n_iter=4] [momentum=0.0] [seeds=40]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import soundml_oracle as O
from soundml_amd import Stft
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 45
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mom = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
seeds = int(sys.argv[4]) if len(sys.argv) > 4 else 40
c = Stft.Config.create(fft_size=2048, hop=512)
o = O.stft_config(2048, hop=512)
n = (frames - 1) * 512
out = []
for seed in range(seeds):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((1, 2, n))
    mag = np.abs(O.transform(o, x)).astype(np.float32)
    init = rng.uniform(-np.pi, np.pi, size=mag.shape).astype(np.float32)
    got = Stft.griffin_lim(c, mag, n_iter=n_iter, momentum=mom, init=init)
    want = O.griffin_lim(o, mag, n_iter=n_iter, momentum=mom, init=init)
    out.append(float(np.linalg.norm(got - want) / np.linalg.norm(want)))
out = np.array(out)
print("frames %d n_iter %d momentum %g, %d seeds: rel l2 median %.2e  p90 %.2e  max %.2e  (> 1e-3: %d)" % (frames, n_iter, mom, seeds, np.median(out), np.percentile(out, 90), out.max(), int((out > 1e-3).sum())))
print(" ".join("%.1e" % v for v in out))

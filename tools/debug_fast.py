import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundml_amd as S
from soundml_amd import Stft
from oracle import soundml_oracle as O
rng = np.random.default_rng(0)
n = 1024 + 40 * 512
x = rng.uniform(-1, 1, size=(1, n)).astype(np.float32)
c = Stft.Config.create(fft_size=2048, hop=512)
got = Stft.power_spectrum(c, x)[0]
want = O.power_spectrum(O.stft_config(2048, hop=512), x)[0]
err = np.abs(got - want) / want.max()
print("frames", got.shape, "max rel err per frame:", np.round(err.max(axis=0), 4))
bad_bins = np.where(err.max(axis=1) > 1e-4)[0]
print("bad bins:", len(bad_bins), bad_bins[:40])
f = 5
print("frame 5 got[:8]", got[:8, f], "want", want[:8, f])
# is got a permutation / scaled version?
print("sum ratio", got[:, f].sum() / want[:, f].sum())
sg, sw = np.sort(got[:, f]), np.sort(want[:, f])
print("sorted match:", np.allclose(sg, sw, rtol=1e-3))

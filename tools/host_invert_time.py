"""Stft.invert of the C2 spectrogram through the host-pointer entry point (1.97 GB up, 0.49 GB down): the result in a block of the
page-locked pool against an ordinary numpy array, interleaved.
  python tools/host_invert_time.py"""
import gc, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundml_amd as S
from soundml_amd import Stft
rng = np.random.default_rng(0)
c = Stft.Config.create(fft_size=2048, hop=512)
x = rng.uniform(-1, 1, size=(256, 480000)).astype(np.float32)
z = np.array(Stft.transform(c, x))          # an ordinary complex64 array
ts = {True: [], False: []}
for rep in range(7):
    for pinned in (True, False):
        S.set_pinned_results(pinned)
        t0 = time.perf_counter()
        y = Stft.invert(c, z, length=480000)
        dt = (time.perf_counter() - t0) * 1e3
        if rep:
            ts[pinned].append(dt)
        else:
            assert np.max(np.abs(y - x)) < 2e-6
        del y
        gc.collect()
for pinned in (True, False):
    v = sorted(ts[pinned])
    print("%-26s min %.1f  median %.1f  max %.1f ms" % ("page-locked result" if pinned else "ordinary numpy result", v[0], v[len(v) // 2], v[-1]))

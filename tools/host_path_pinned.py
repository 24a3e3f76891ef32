"""C2 through the host-pointer entry point with the result array from the page-locked pool (default since round 5) against an
ordinary fresh numpy array: 12 calls each, interleaved; the result is dropped outside the timed region (as nx's GC would).
  python tools/host_path_pinned.py [clips=256]"""
import gc, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundml_amd as S
from soundml_amd import Stft
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, size=(clips, 480000)).astype(np.float32)
c = Stft.Config.create(fft_size=2048, hop=512)
ts = {True: [], False: []}
ref = None
for rep in range(13):
    for pinned in (True, False):
        S.set_pinned_results(pinned)
        t0 = time.perf_counter()
        p = Stft.power_spectrum(c, x)
        dt = (time.perf_counter() - t0) * 1e3
        if rep == 0:
            if ref is None:
                ref = p[::37].copy()
            else:
                assert np.array_equal(ref, p[::37])
        else:
            ts[pinned].append(dt)
        del p
        gc.collect()
for pinned in (True, False):
    v = sorted(ts[pinned])
    print("%-28s min %.1f  q1 %.1f  median %.1f  q3 %.1f  max %.1f ms   (%s)" % ("page-locked result (pool)" if pinned else "ordinary numpy result", v[0], v[len(v) // 4], v[len(v) // 2], v[3 * len(v) // 4], v[-1],
          " ".join("%.0f" % t for t in ts[pinned])))
zt = {True: [], False: []}
for rep in range(6):
    for pinned in (True, False):
        S.set_pinned_results(pinned)
        t0 = time.perf_counter()
        z = Stft.transform(c, x)
        dt = (time.perf_counter() - t0) * 1e3
        if rep:
            zt[pinned].append(dt)
        del z
        gc.collect()
print("Stft.transform (1.97 GB down): page-locked median %.1f ms, ordinary %.1f" % (sorted(zt[True])[2], sorted(zt[False])[2]))

"""The host-pointer power spectrogram at C2 (numpy in, fresh numpy out) with the clip-unit pipeline on and off, interleaved in one
process: 12 calls each, every call timed.   python tools/host_path_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from soundml_amd import Stft
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = np.random.default_rng(0).uniform(-1, 1, size=(clips, 480000)).astype(np.float32)
c = Stft.Config.create(fft_size=2048, hop=512)
res = {"1": [], "0": []}
ref = None
for rnd in range(12):
    for mode in ("1", "0"):
        os.environ["SMX_HOST_PIPELINE"] = mode
        a = time.perf_counter(); y = Stft.power_spectrum(c, x); b = time.perf_counter()
        res[mode].append((b - a) * 1e3)
        if ref is None: ref = y[:3].copy()
        assert np.array_equal(ref, y[:3])
        del y
for mode, name in (("1", "pipelined"), ("0", "serial")):
    v = sorted(res[mode][1:])
    print("%-9s min %.1f  q1 %.1f  median %.1f  q3 %.1f  max %.1f ms   (%s)" % (name, v[0], v[len(v) // 4], v[len(v) // 2], v[3 * len(v) // 4], v[-1], " ".join("%.0f" % t for t in res[mode])))

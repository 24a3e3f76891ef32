"""Stage times of the pipelined host call (SMX_HOST_TRACE=1 prints them from inside the library): C2's power spectrogram through
the host-pointer entry point into a page-locked result block (pool) and into an ordinary numpy array, alternating, then four
page-locked calls in a row.
  SMX_HOST_TRACE=1 python tools/host_path_trace.py"""
import gc, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundml_amd import Stft, _lib
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, size=(256, 480000)).astype(np.float32)
c = Stft.Config.create(fft_size=2048, hop=512)
shape = (256, 1025, 938)
ref = None
for rep in range(12):
    pinned = rep % 2 == 0 or rep > 7
    out = _lib.host_result(shape, np.float32) if pinned else np.zeros(shape, np.float32)
    t1 = time.perf_counter()
    _lib.check(_lib.lib.smx_stft_power_spectrum_f32(c._h, x.ctypes.data, 256, 480000, 2.0, out.ctypes.data))
    t2 = time.perf_counter()
    if ref is None:
        ref = out[::31].copy()
    assert np.array_equal(ref, out[::31])
    print("rep %d: %s call %.2f ms" % (rep, "page-locked" if pinned else "ordinary", (t2 - t1) * 1e3), flush=True)
    del out
    gc.collect()

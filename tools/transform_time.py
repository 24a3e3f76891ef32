"""Stft.transform (complex64 spectrogram) timing on a C2-shaped batch, device resident, through the C ABI.
    python tools/transform_time.py [samples_per_clip ...]      480000 -> 938 frames (row pitch 7504 B), 482816 -> 944 (7552 B = 59 x 128)
Also times the power spectrogram of the same batch for comparison."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundml_amd import Stft
from soundml_amd._lib import check, lib
vp = ctypes.c_void_p
clips = int(os.environ.get("CLIPS", 256))
c = Stft.Config.create(fft_size=2048, hop=512)
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]
for n in ([int(a) for a in sys.argv[1:]] or [480000, 482816]):
    frames = Stft.frames(c, n)
    x = torch.empty(clips, n, device="cuda").uniform_(-1, 1)
    oc = torch.empty(clips, 1025, frames, 2, device="cuda")
    op = torch.empty(clips, 1025, frames, device="cuda")
    mc, _ = timeit(lambda: check(lib.smx_stft_transform_range_f32_dev(c._h, vp(x.data_ptr()), clips, n, n, 0, frames, vp(oc.data_ptr()), None)))
    mp, _ = timeit(lambda: check(lib.smx_stft_power_range_f32_dev(c._h, vp(x.data_ptr()), clips, n, n, 0, frames, 2.0, vp(op.data_ptr()), None)))
    print("n %d frames %d (row pitch %d B complex): transform %.4f ms = %.1f Mframes/s, %.2f TB/s algorithmic | power %.4f ms = %.1f Mframes/s" % (
        n, frames, frames * 8, mc, clips * frames / mc / 1e3, clips * frames * 10248 / mc / 1e9, mp, clips * frames / mp / 1e3))

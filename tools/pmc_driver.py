"""Driver for tools/profile_round.sh: REPS back-to-back launches of every hot kernel of the BASELINE configurations, one kernel
after the other, nothing else on the device (C2 power spectrogram, C3 fused mel, Mel.apply, Stft.transform and Stft.invert of the C2 batch, the power spectrogram
at fft 1024, 512 and 256 on 256 clips of C1's length, C4 FIR, the C2 power spectrogram under the float64 interior)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import soundml_amd as S
from soundml_amd import Fir, Mel, Stft
from soundml_amd._lib import check, lib
vp = ctypes.c_void_p
x = torch.rand(256, 480000, device="cuda") * 2 - 1
sc = Stft.Config.create(fft_size=2048, hop=512)
mc = Mel.Config.create(n_mels=128, sample_rate=48000, fft_size=2048)
frames = Stft.frames(sc, 480000)
out = torch.empty(256, 1025, frames, device="cuda")
mout = torch.empty(256, 128, frames, device="cuda")
zc = torch.empty(256, 1025, frames, 2, device="cuda")
h = Fir.design_lowpass(8192, 0.25, 100.0)
plan = Fir.Plan.create(h)
c1k = Stft.Config.create(fft_size=1024, hop=256)      # the lanes kernels at the sizes of tools/bench_extra.py
c512 = Stft.Config.create(fft_size=512, hop=128)
c256 = Stft.Config.create(fft_size=256, hop=64)
f1k, f512, f256 = Stft.frames(c1k, 441000), Stft.frames(c512, 441000), Stft.frames(c256, 441000)
x1 = torch.rand(256, 441000, device="cuda") * 2 - 1
o1k = torch.empty(256, 513, f1k, device="cuda")
o512 = torch.empty(256, 257, f512, device="cuda")
o256 = torch.empty(256, 129, f256, device="cuda")
xs = torch.rand(8, 2880000, device="cuda") * 2 - 1
ys = torch.empty_like(xs)
# Every kernel is launched at least REPS times (default 24) and for at least SUSTAIN_MS (600 ms) in back-to-back groups of 8 before the next one starts, and the profile's durations and
# counters are taken from its LAST 10 launches (tools/profile_round.sh): the sustained state bench.py measures, not the burst out
# of an idle device that round 4's four round-robin launches sampled (8-28 % above the bench line's figures: VERDICT r4, weak 6).
reps = int(os.environ.get("REPS", "24"))
sustain_ms = float(os.environ.get("SUSTAIN_MS", "600"))   # ... and for at least this long: the chip needs ~0.5 s of one kernel to settle (profiles/r07/NOTES.md: 613 -> 482 us within 60 launches, 447 after ~1000: what bench.py's adaptive preconditioning waits for)
import time
zv = torch.view_as_complex(zc)
yr = torch.empty(256, 480000, device="cuda")
lib.smx_stft_invert_f32_dev.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, vp, vp]
def each(fn):
    t0, n = time.perf_counter(), 0
    while n < reps or (time.perf_counter() - t0) * 1e3 < sustain_ms:
        for _ in range(8):
            fn()
        n += 8
        torch.cuda.synchronize()
each(lambda: check(lib.smx_stft_power_range_f32_dev(sc._h, vp(x.data_ptr()), 256, 480000, 480000, 0, frames, 2.0, vp(out.data_ptr()), None)))
each(lambda: check(lib.smx_stft_transform_range_f32_dev(sc._h, vp(x.data_ptr()), 256, 480000, 480000, 0, frames, vp(zc.data_ptr()), None)))
each(lambda: check(lib.smx_stft_invert_f32_dev(sc._h, vp(zc.data_ptr()), 256, 1025, frames, 1, 480000, vp(yr.data_ptr()), None)))
each(lambda: check(lib.smx_mel_spectrogram_f32_dev(sc._h, mc._h, vp(x.data_ptr()), 256, 480000, 480000, 2.0, vp(mout.data_ptr()), None)))
each(lambda: check(lib.smx_mel_apply_f32_dev(mc._h, vp(out.data_ptr()), 256, 1025, frames, vp(mout.data_ptr()), None)))
each(lambda: check(lib.smx_stft_power_range_f32_dev(c1k._h, vp(x1.data_ptr()), 256, 441000, 441000, 0, f1k, 2.0, vp(o1k.data_ptr()), None)))
each(lambda: check(lib.smx_stft_power_range_f32_dev(c512._h, vp(x1.data_ptr()), 256, 441000, 441000, 0, f512, 2.0, vp(o512.data_ptr()), None)))
each(lambda: check(lib.smx_stft_power_range_f32_dev(c256._h, vp(x1.data_ptr()), 256, 441000, 441000, 0, f256, 2.0, vp(o256.data_ptr()), None)))
each(lambda: check(lib.smx_fir_apply_f32_dev(plan._h, vp(xs.data_ptr()), 8, 2880000, 2880000, vp(ys.data_ptr()), 2880000, None)))
# fft 4096 / hop 1024 on the same 256 x 441000 batch: stft4096_power64_kernel (round 6)
c4k = Stft.Config.create(fft_size=4096, hop=1024)
f4k = Stft.frames(c4k, 441000)
o4k = torch.empty(256, 2049, f4k, device="cuda")
each(lambda: check(lib.smx_stft_power_range_f32_dev(c4k._h, vp(x1.data_ptr()), 256, 441000, 441000, 0, f4k, 2.0, vp(o4k.data_ptr()), None)))
del o4k
# Griffin-Lim's loop (6 iterations a call): its frame-major kernels, stft2048_complex_fm_kernel and istft2048_pipe_kernel<true, true>
lib.smx_stft_griffin_lim_f32_dev.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_double, vp, ctypes.c_int, ctypes.c_int64, vp, vp]
out.copy_(torch.rand_like(out))
each(lambda: check(lib.smx_stft_griffin_lim_f32_dev(sc._h, vp(out.data_ptr()), 256, 1025, frames, 6, 0.99, None, 1, 480000, vp(yr.data_ptr()), None)))
S.set_interior("float64")   # the reference's own numerics at C2: stft2048_power_wide_kernel
each(lambda: check(lib.smx_stft_power_range_f32_dev(sc._h, vp(x.data_ptr()), 256, 480000, 480000, 0, frames, 2.0, vp(out.data_ptr()), None)))
S.set_interior("float32")
torch.cuda.synchronize()

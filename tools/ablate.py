"""Diagnostic: time the fused STFT kernel (stamps build) under timing-only ablations.
SMX_ABLATE: 0 none, 1 no HBM stores, 2 no sample loads, 3 neither, 4 no post-pass permutes, 5 no transposes, 6 no FFT (memory + sync only), 7 no FFT and no stores,
8 no FFT + stores as 128-byte runs, 9 no FFT and no loads (stores + sync only)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    import torch
    lib = ctypes.CDLL(os.path.join(ROOT, "soundml_amd", os.environ.get("SMX_DIAG_LIB", "lib_diag"), "libsoundml_amd.so"))
    i64, vp = ctypes.c_int64, ctypes.c_void_p
    h = vp()
    lib.smx_stft_config_create.argtypes = [i64, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(vp)]
    assert lib.smx_stft_config_create(2048, -(2**63), 512, 0, 0, 0.0, 0, 0, None, ctypes.byref(h)) == 0
    clips, n = 256, int(os.environ.get("AB_N", "480000"))
    frames = 1 + n // 512
    x = torch.rand(clips, n, device="cuda") * 2 - 1
    out = torch.empty(clips, 1025, frames, device="cuda")
    lib.smx_stft_power_range_f32_dev.argtypes = [vp, vp, i64, i64, i64, i64, i64, ctypes.c_double, vp, vp]
    def run():
        assert lib.smx_stft_power_range_f32_dev(h, vp(x.data_ptr()), clips, n, n, 2, frames - 2, 2.0, vp(out.data_ptr()), None) == 0
    for _ in range(3): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print("ablate=%s  frames/clip %d  median %.3f ms  min %.3f ms  (%.1f Mframes/s)" % (os.environ.get("SMX_ABLATE", "0"), frames, sorted(ts)[5], min(ts), clips * (frames - 4) / sorted(ts)[5] / 1e3))
else:
    for abl in os.environ.get("ABLS", "016789"):
        env = dict(os.environ, SMX_ABLATE=abl)
        subprocess.call([sys.executable, __file__, "run"], env=env)

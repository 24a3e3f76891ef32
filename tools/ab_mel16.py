"""Diagnostic: interleaved A/B of two builds on Soundml.mel_spectrogram at fft 1024 / hop 256 and fft 512 / hop 128 (80 mels)."""
import ctypes, os, sys, torch
i64, vp, ci, f64 = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_double
paths = sys.argv[1:]
for fft, hop, sr, n in ((1024, 256, 22050, 441000), (512, 128, 16000, 441000)):
    clips, frames = 256, 1 + n // hop
    x = torch.rand(clips, n, device="cuda") * 2 - 1
    out = torch.empty(clips, 80, frames, device="cuda")
    libs = []
    for p in paths:
        lib = ctypes.CDLL(os.path.abspath(p))
        h, m = vp(), vp()
        lib.smx_stft_config_create.argtypes = [i64, i64, i64, ci, ci, f64, ci, ci, vp, ctypes.POINTER(vp)]
        assert lib.smx_stft_config_create(fft, -(2**63), hop, 0, 0, 0.0, 0, 0, None, ctypes.byref(h)) == 0
        lib.smx_mel_config_create.argtypes = [i64, i64, i64, f64, ci, f64, ci, ci, ctypes.POINTER(vp)]
        assert lib.smx_mel_config_create(80, sr, fft, 0.0, 0, 0.0, 0, 0, ctypes.byref(m)) == 0
        lib.smx_mel_spectrogram_f32_dev.argtypes = [vp, vp, vp, i64, i64, i64, f64, vp, vp]
        libs.append((p, lib, h, m))
    def run(lib, h, m):
        assert lib.smx_mel_spectrogram_f32_dev(h, m, vp(x.data_ptr()), clips, n, n, 2.0, vp(out.data_ptr()), None) == 0
    for p, lib, h, m in libs:
        for _ in range(3): run(lib, h, m)
    torch.cuda.synchronize()
    ts = {p: [] for p in paths}
    for rnd in range(30):
        for p, lib, h, m in libs:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(lib, h, m); b.record(); torch.cuda.synchronize(); ts[p].append(a.elapsed_time(b))
    for p in paths:
        t = sorted(ts[p])
        print("fft %4d  %-45s min %.4f  median %.4f ms" % (fft, p, t[0], t[len(t) // 2]))

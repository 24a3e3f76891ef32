"""In-process A/B timing of the fused mel-spectrogram entry point of several builds (see tools/ab.py)."""
import ctypes, os, sys
import torch
i64, vp = ctypes.c_int64, ctypes.c_void_p
paths = sys.argv[1:]
clips, n = 256, 480000
frames = 1 + n // 512
x = torch.rand(clips, n, device="cuda") * 2 - 1
out = torch.empty(clips, 128, frames, device="cuda")
libs = []
for p in paths:
    lib = ctypes.CDLL(os.path.abspath(p.split("@")[0]))
    h, mh = vp(), vp()
    lib.smx_stft_config_create.argtypes = [i64, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(vp)]
    assert lib.smx_stft_config_create(2048, -(2**63), 512, 0, 0, 0.0, 0, 0, None, ctypes.byref(h)) == 0
    lib.smx_mel_config_create.argtypes = [i64, i64, i64, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]
    assert lib.smx_mel_config_create(128, 48000, 2048, 0.0, 0, 0.0, 0, 0, ctypes.byref(mh)) == 0, lib.smx_last_error()
    lib.smx_mel_spectrogram_f32_dev.argtypes = [vp, vp, vp, i64, i64, i64, ctypes.c_double, vp, vp]
    libs.append((p, lib, h, mh))
def setenv(p):
    for kv in p.split("@")[1:]:
        k, v = kv.split("="); os.environ[k] = v
def clearenv(p):
    for kv in p.split("@")[1:]: os.environ.pop(kv.split("=")[0], None)
def run(lib, h, mh):
    assert lib.smx_mel_spectrogram_f32_dev(h, mh, vp(x.data_ptr()), clips, n, n, 2.0, vp(out.data_ptr()), None) == 0
for p, lib, h, mh in libs:
    setenv(p)
    for _ in range(3): run(lib, h, mh)
    clearenv(p)
torch.cuda.synchronize()
ts = {p: [] for p in paths}
for rnd in range(int(os.environ.get("AB_ROUNDS", "30"))):
    for p, lib, h, mh in libs:
        setenv(p)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(4): run(lib, h, mh)
        b.record(); torch.cuda.synchronize()
        clearenv(p)
        ts[p].append(a.elapsed_time(b) / 4)
for p in paths:
    v = sorted(ts[p])
    print("%-52s min %.4f  q1 %.4f  median %.4f  q3 %.4f ms  (%.1f Mframes/s at median)"
          % (p[-52:], v[0], v[len(v) // 4], v[len(v) // 2], v[3 * len(v) // 4], clips * frames / v[len(v) // 2] / 1e3))

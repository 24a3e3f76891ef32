"""Interleaved timing of ONE build under several environments (options the library reads per launch):
  python tools/ab_env.py <lib.so> "" "SMX_POWER_SKEW=0" "SMX_NOSTORE=1"
C2 power spectrogram (256 x 480000), HIP events, median / min over interleaved rounds.  AB_POWER=<p> in a variant sets the
exponent of |X|^p for that variant (default 2)."""
import ctypes, os, sys
import torch
i64, vp = ctypes.c_int64, ctypes.c_void_p
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
envs = sys.argv[2:]
clips, n = int(os.environ.get("AB_CLIPS", "256")), int(os.environ.get("AB_N", "480000"))
frames = 1 + n // 512
x = torch.rand(clips, n, device="cuda") * 2 - 1
out = torch.empty(clips, 1025, frames, device="cuda")
h = vp()
lib.smx_stft_config_create.argtypes = [i64, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(vp)]
assert lib.smx_stft_config_create(2048, -(2**63), 512, 0, 0, 0.0, 0, 0, None, ctypes.byref(h)) == 0
lib.smx_stft_power_range_f32_dev.argtypes = [vp, vp, i64, i64, i64, i64, i64, ctypes.c_double, vp, vp]
def run():
    assert lib.smx_stft_power_range_f32_dev(h, vp(x.data_ptr()), clips, n, n, 0, frames, float(os.environ.get("AB_POWER", "2.0")), vp(out.data_ptr()), None) == 0
def setenv(e, on):
    for kv in filter(None, e.split(",")):
        k, v = kv.split("=")
        if on: os.environ[k] = v
        else: os.environ.pop(k, None)
for e in envs:
    setenv(e, True)
    for _ in range(3): run()
    setenv(e, False)
torch.cuda.synchronize()
ts = {e: [] for e in envs}
for rnd in range(int(os.environ.get("AB_ROUNDS", "30"))):
    for e in envs:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        setenv(e, True)
        a.record()
        for _ in range(4): run()
        b.record(); torch.cuda.synchronize()
        setenv(e, False)
        ts[e].append(a.elapsed_time(b) / 4)
for e in envs:
    v = sorted(ts[e])
    print("%-44s min %.4f  median %.4f  q3 %.4f ms  (%.1f Mframes/s at median)" % (e or "(default)", v[0], v[len(v) // 2], v[3 * len(v) // 4], clips * frames / v[len(v) // 2] / 1e3))

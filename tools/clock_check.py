"""The clock the chip holds under a kernel (MI355X_MICROARCH.md 'DVFS give-back' item 6): a `make COARSE=1` build stamps
s_memtime / s_memrealtime once around the tile loop of the fft-2048 power kernel and of the float64-interior kernel (no per-phase
stamps: the kernels run as shipped).  After 2 s of back-to-back launches:  python tools/clock_check.py [power|f64]"""
import ctypes, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "soundml_amd", "lib_clock", "libsoundml_amd.so"))
i64, vp = ctypes.c_int64, ctypes.c_void_p
mode = sys.argv[1] if len(sys.argv) > 1 else "power"
h = vp()
lib.smx_stft_config_create.argtypes = [i64, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(vp)]
assert lib.smx_stft_config_create(2048, -(2**63), 512, 0, 0, 0.0, 0, 0, None, ctypes.byref(h)) == 0
if mode == "f64":
    assert lib.smx_set_interior(1) == 0
clips, n = 256, 480000
frames = 1 + n // 512
x = torch.rand(clips, n, device="cuda") * 2 - 1
out = torch.empty(clips, 1025, frames, device="cuda")
lib.smx_stft_power_range_f32_dev.argtypes = [vp, vp, i64, i64, i64, i64, i64, ctypes.c_double, vp, vp]
def run():
    assert lib.smx_stft_power_range_f32_dev(h, vp(x.data_ptr()), clips, n, n, 0, frames, 2.0, vp(out.data_ptr()), None) == 0
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(20): run()
    torch.cuda.synchronize()
ev = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize(); ev.append(a.elapsed_time(b))
S, nwg = 24, 256
buf = np.zeros(nwg * 16 * S, dtype=np.uint64)
read = lib.smx_debug_read_stamps64 if mode == "f64" else lib.smx_debug_read_stamps
assert read(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), buf.size) == 0
st = buf.reshape(nwg, 16, S).astype(np.float64)[:, :8, :]
cyc, ref = st[:, :, 20], st[:, :, 21]
clk = np.median(cyc / ref) / 10.0
tiles = 256 * ((frames + 15) // 16) / nwg
print("%s: launch %.4f ms (median of 10 after 2 s of launches); tile loop %.0f cycles per workgroup = %.0f per tile of 16 frames; in-kernel clock %.2f GHz (median over workgroups; min %.2f, max %.2f)"
      % (mode, sorted(ev)[5], np.median(cyc), np.median(cyc) / tiles, clk, (cyc / ref).min() / 10.0, (cyc / ref).max() / 10.0))

"""Diagnostic: the clock the chip holds inside the fused STFT kernel (make COARSE=1 -> lib_clock).
In-kernel clock = shader cycles (s_memtime) / reference ticks (s_memrealtime, 100 MHz) around the tile loop, median over
workgroups, after a few hundred back-to-back launches on random data (MI355X_MICROARCH.md 'DVFS give-back' item 6).
SMX_ABLATE selects a timing-only ablation (tools/ablate.py lists them)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "soundml_amd", "lib_clock", "libsoundml_amd.so"))
i64, vp = ctypes.c_int64, ctypes.c_void_p
h = vp()
lib.smx_stft_config_create.argtypes = [i64, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(vp)]
assert lib.smx_stft_config_create(2048, -(2**63), 512, 0, 0, 0.0, 0, 0, None, ctypes.byref(h)) == 0
clips, n = 256, 480000
x = torch.rand(clips, n, device="cuda") * 2 - 1
out = torch.empty(clips, 1025, 938, device="cuda")
lib.smx_stft_power_range_f32_dev.argtypes = [vp, vp, i64, i64, i64, i64, i64, ctypes.c_double, vp, vp]
def run():
    assert lib.smx_stft_power_range_f32_dev(h, vp(x.data_ptr()), clips, n, n, 2, 936, 2.0, vp(out.data_ptr()), None) == 0
for _ in range(int(os.environ.get("WARM", "400"))):   # let the clock settle under this load
    run()
torch.cuda.synchronize()
ev = []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize(); ev.append(e0.elapsed_time(e1))
S, nwg = 24, 256
buf = np.zeros(nwg * 16 * S, dtype=np.uint64)
assert lib.smx_debug_read_stamps(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), buf.size) == 0
st = buf.reshape(nwg, 16, S).astype(np.float64)[:, :int(os.environ.get('WAVES', '16')), :]
cyc, ref = st[:, :, 20], st[:, :, 21]
clock = np.median(cyc / ref) * 100.0
tiles = 256 * 934 / 16 / nwg
print("SMX_ABLATE=%s  wall %.3f ms (median of 20)  loop %.0f cycles per wave = %.0f per tile  in-kernel clock %.0f MHz  (loop %.3f ms)"
      % (os.environ.get("SMX_ABLATE", "0"), sorted(ev)[10], np.median(cyc), np.median(cyc) / tiles, clock, np.median(ref) / 1e5))

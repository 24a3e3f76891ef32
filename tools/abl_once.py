"""Diagnostic: a few launches of the interior kernel of the diagnostic build (SMX_ABLATE / SMX_ABL_RUN from the
environment), for use under rocprofv3 --pmc."""
import ctypes, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "soundml_amd", "lib_diag", "libsoundml_amd.so"))
i64, vp = ctypes.c_int64, ctypes.c_void_p
h = vp()
lib.smx_stft_config_create.argtypes = [i64, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(vp)]
assert lib.smx_stft_config_create(2048, -(2**63), 512, 0, 0, 0.0, 0, 0, None, ctypes.byref(h)) == 0
clips, n = 256, 480000
x = torch.rand(clips, n, device="cuda") * 2 - 1
out = torch.empty(clips, 1025, 938, device="cuda")
lib.smx_stft_power_range_f32_dev.argtypes = [vp, vp, i64, i64, i64, i64, i64, ctypes.c_double, vp, vp]
for _ in range(3):
    assert lib.smx_stft_power_range_f32_dev(h, vp(x.data_ptr()), clips, n, n, 2, 936, 2.0, vp(out.data_ptr()), None) == 0
torch.cuda.synchronize()

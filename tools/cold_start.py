"""Where the first host-pointer call of a process spends its time (252 ms for C1, device time 34 us).
Separates the HIP runtime's own start (context, first allocation) from the library's: code-object load at the first kernel
launch, table build + upload for a new configuration, pinned staging, and the steady state.
  python tools/cold_start.py          (SMX_HOST_TRACE=1 prints the library's own upload / kernels / download split)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
t0 = time.perf_counter()
import torch
t_import = time.perf_counter() - t0
t0 = time.perf_counter(); torch.cuda.init(); a = torch.zeros(1, device="cuda"); torch.cuda.synchronize(); t_runtime = time.perf_counter() - t0
t0 = time.perf_counter()
import soundml_amd as S
from soundml_amd import Stft
t_lib = time.perf_counter() - t0
n = 441000
x1 = np.sin(2 * np.pi * 440.0 * np.arange(n) / 44100.0).astype(np.float32)
def call(c, x):
    t0 = time.perf_counter(); p = Stft.power_spectrum(c, x); return (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); c1 = Stft.Config.create(fft_size=1024, hop=256); t_cfg = (time.perf_counter() - t0) * 1e3
first = call(c1, x1)
second = call(c1, x1)
third = call(c1, x1)
c2 = Stft.Config.create(fft_size=2048, hop=512)
new_cfg = call(c2, x1)
new_cfg2 = call(c2, x1)
c3 = Stft.Config.create(fft_size=400, hop=160)
other_kernel = call(c3, x1)
other_kernel2 = call(c3, x1)
print(json.dumps({"import_torch_s": round(t_import, 2), "hip_runtime_first_use_ms": round(t_runtime * 1e3, 1), "import_library_ms": round(t_lib * 1e3, 1),
                  "config_create_ms": round(t_cfg, 3), "C1_first_call_ms": round(first, 2), "C1_second_ms": round(second, 2), "C1_third_ms": round(third, 2),
                  "fft2048_first_call_ms": round(new_cfg, 2), "fft2048_second_ms": round(new_cfg2, 2),
                  "fft400_first_call_ms": round(other_kernel, 2), "fft400_second_ms": round(other_kernel2, 2)}))

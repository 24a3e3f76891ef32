"""C4 FIR timing (8192 taps, 8 x 2 880 000 samples, device resident; FIR_CH / FIR_NS: other batches) plus a parity check on one channel against the oracle.
A/B of the two N = 32768 kernels: run once plain and once with SMX_FIR_SPLIT=0 (the switch is read once per process).
    python tools/fir_time.py [taps ...]
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundml_amd import Fir
from soundml_amd._lib import check, lib
from oracle import soundml_oracle as O

vp = ctypes.c_void_p
ch, ns = int(os.environ.get("FIR_CH", "8")), int(os.environ.get("FIR_NS", "2880000"))   # FIR_CH=64: the per-block cost without C4's tail (938 blocks on 256 CUs = 3.7 rounds)
x = torch.empty(ch, ns, device="cuda").uniform_(-1, 1)
y = torch.empty_like(x)
for taps in ([int(a) for a in sys.argv[1:]] or [8192]):
    h = Fir.design_lowpass(taps, 0.25, 100.0)
    plan = Fir.Plan.create(h)
    call = lambda: check(lib.smx_fir_apply_f32_dev(plan._h, vp(x.data_ptr()), ch, ns, ns, vp(y.data_ptr()), ns, None))
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    want = O.fir_filter(h, x[3].cpu().numpy().astype(np.float64))
    err = float(np.max(np.abs(y[3].cpu().numpy() - want)))
    print("taps %6d  N %6d  SMX_FIR_SPLIT=%s  min %.4f  median %.4f ms  (%.1f Gsamples/s, %.3f of 8 TB/s)  max abs err %.3g (bound %.3g)" % (
        taps, plan.block, os.environ.get("SMX_FIR_SPLIT", "-"), ts[0], ts[len(ts) // 2], ch * ns / ts[len(ts) // 2] / 1e6,
        ch * ns * 8 / ts[len(ts) // 2] / 1e6 / 8000, err, 1e-5 * float(np.sum(np.abs(h)))))

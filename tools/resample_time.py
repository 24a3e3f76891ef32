"""Resample OLS stage timing (device resident, 8 channels x 60 s at 48 kHz): x2, /2, /3, x3 with k = 160 (the reference's default
quality region), against the oracle on a short prefix."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundml_amd import Fir, Resample
from oracle import soundml_oracle as O
def t(fn, reps=9):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
x = torch.rand(8, 2880000, device="cuda") * 2 - 1
for l, m in ((2, 1), (1, 2), (1, 3), (3, 1)):
    k = 160
    proto = Resample.prototype(l=max(l, m), k=k, fc=0.45 / max(l, m), beta=Fir.kaiser_beta(100.0)) if l > 1 else Resample.prototype(l=1, k=k * m, fc=0.45 / m, beta=Fir.kaiser_beta(100.0))
    try:
        st = Resample.Stage.create(proto, l=l, m=m, k=(k if l > 1 else k * m))
    except Exception as e:
        print("x%d /%d: %s" % (l, m, e)); continue
    ms = t(lambda: Resample.Stage.apply(st, x))
    n_out = st.out_length(2880000)
    print("x%d /%d  taps %d: %.3f ms for 8 x 2880000 -> %d  (%.1f Gsamples/s in, %.0f GB/s of in + out)" % (
        l, m, proto.shape[0], ms, n_out, 8 * 2880000 / ms / 1e6, 8 * (2880000 + n_out) * 4 / ms / 1e6), flush=True)

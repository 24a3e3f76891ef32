"""FIR block convolution (BASELINE config 4; the reference lists a generic FIR
filter as planned, README.md:27-34, and its only FIR arithmetic is the
overlap-save resample executor, resample.ml:383-415, which this follows).

    h = Fir.design_lowpass(8192, cutoff=0.25, attenuation=80.0)
    p = Fir.Plan.create(h)
    y = Fir.apply(p, x)        # [...; n] -> [...; n], y[i] = sum_k h[k] x[i-k]
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib
from ._tensor import Batch, out_ptr, prod


def kaiser_beta(attenuation: float) -> float:
    """resample.ml:105-109."""
    out = C.c_double()
    check(lib.smx_fir_kaiser_beta(float(attenuation), C.byref(out)))
    return out.value


def design_lowpass(taps: int, cutoff: float, attenuation: float = 80.0) -> np.ndarray:
    """Kaiser-windowed sinc (arithmetic of resample.ml:145-163), float64, unit DC gain."""
    h = np.empty(max(int(taps), 1), dtype=np.float64)
    check(lib.smx_fir_design_lowpass(int(taps), float(cutoff), kaiser_beta(attenuation),
                                     C.c_void_p(h.ctypes.data)))
    return h[:taps]


class Plan:
    def __init__(self, handle, taps):
        self._h, self.taps = handle, taps

    @staticmethod
    def create(h) -> "Plan":
        h = np.ascontiguousarray(np.asarray(h, dtype=np.float64))
        handle = C.c_void_p()
        check(lib.smx_fir_plan_create(C.c_void_p(h.ctypes.data), int(h.shape[0]), C.byref(handle)))
        return Plan(handle, int(h.shape[0]))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            try:
                lib.smx_fir_plan_destroy(h)
            except Exception:
                pass

    block = property(lambda self: lib.smx_fir_plan_block(self._h))


def apply(p: Plan, x):
    b = Batch(x, "fir_apply")
    if b.bytes != 4:
        raise _lib.InvalidArgument("fir_apply: cannot filter float64 audio (this path is float32)")
    n = int(b.shape[-1])
    lead = prod(b.shape[:-1])
    out = b.empty(b.shape)
    if b.device:
        with b.device_guard():
            check(lib.smx_fir_apply_f32_dev(p._h, b.ptr(), lead, n, n, out_ptr(out), n, b.stream()))
        return out
    check(lib.smx_fir_apply_f32(p._h, b.ptr(), lead, n, out_ptr(out)))
    return b.wrap(out)

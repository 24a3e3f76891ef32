"""Resample: the overlap-save executor's pieces on the device.

The reference's planner / polyphase bank / cascade logic stay where they are (resample.ml); this mirrors the three
internal pieces a maintainer would route to the device: the OLS geometry, the prototype design, the block identity of
``soundml_resample_shape`` (resample_stubs.c:329-422) and one whole stage (``ols_run`` + drain) as a block convolution.

    proto = Resample.prototype(l=2, k=160, fc=0.45 / 2, beta=Fir.kaiser_beta(100.0))
    st = Resample.Stage.create(proto, l=2, m=1, k=160)
    y = Resample.Stage.apply(st, x)            # [...; n] -> [...; ceil(n L / M)]
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib
from ._tensor import Batch, out_ptr, prod


def ols_geom(rate: int, l: int, m: int, k: int):
    """resample.ml:292-300: (N, B, delta), or None when the stage is not OLS-eligible."""
    n, b, d, ok = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
    check(lib.smx_resample_ols_geom(int(rate), int(l), int(m), int(k), C.byref(n), C.byref(b), C.byref(d), C.byref(ok)))
    return (n.value, b.value, d.value) if ok.value else None


def prototype(l: int, k: int, fc: float, beta: float) -> np.ndarray:
    """resample.ml:145-163 `design_prototype`: 2 K L + 1 taps, float64, sum = L."""
    h = np.empty(2 * int(k) * int(l) + 1, dtype=np.float64)
    check(lib.smx_resample_prototype(int(l), int(k), float(fc), float(beta), C.c_void_p(h.ctypes.data)))
    return h


def shape(x: np.ndarray, h: np.ndarray, n: int, sl: int = 1, sm: int = 1) -> np.ndarray:
    """`soundml_resample_shape` (resample_stubs.c:329-422): complex128 half spectra [lines; n/2+1] -> [lines; w/2+1]."""
    x = np.ascontiguousarray(x, dtype=np.complex128)
    h = np.ascontiguousarray(h, dtype=np.complex128)
    lines = int(np.prod(x.shape[:-1])) if x.ndim > 1 else 1
    if n < 2 or n % 2 or sl < 1 or sm < 1 or (sl > 1 and sm > 1) or (sm > 1 and n % sm) or (n // sm if sm > 1 else n) < 2:
        raise _lib.Failure("soundml_resample_shape: invalid geometry")      # resample_stubs.c:383-389, checked first
    w = n * sl if sl > 1 else (n // sm if sm > 1 else n)
    if x.shape[-1] < n // 2 + 1 or h.shape[-1] < (w // 2 + 1 if sl > 1 else n // 2 + 1):
        raise _lib.Failure("soundml_resample_shape: buffer extents disagree")
    out = np.empty(x.shape[:-1] + (max(w, 0) // 2 + 1,), dtype=np.complex128)
    check(lib.smx_resample_shape_c128(C.c_void_p(x.ctypes.data), C.c_void_p(h.ctypes.data), C.c_void_p(out.ctypes.data),
                                      lines, int(n), int(sl), int(sm)))
    return out


class Stage:
    def __init__(self, handle, l, m, k):
        self._h, self.l, self.m, self.k = handle, l, m, k

    @staticmethod
    def create(proto, l: int, m: int, k: int) -> "Stage":
        proto = np.ascontiguousarray(np.asarray(proto, dtype=np.float64))
        if proto.shape != (2 * int(k) * int(l) + 1,):
            raise _lib.InvalidArgument("resample_stage_create: cannot use a %d-tap prototype for l = %d, k = %d (the "
                                       "prototype has 2 K L + 1 taps)" % (proto.shape[0], l, k))
        handle = C.c_void_p()
        check(lib.smx_resample_stage_create(C.c_void_p(proto.ctypes.data), int(l), int(m), int(k), C.byref(handle)))
        return Stage(handle, int(l), int(m), int(k))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            try:
                lib.smx_resample_stage_destroy(h)
            except Exception:
                pass

    def out_length(self, n: int) -> int:
        return lib.smx_resample_stage_out_length(self._h, int(n))

    @staticmethod
    def apply(st: "Stage", x):
        b = Batch(x, "resample_stage")
        if b.bytes != 4:
            raise _lib.InvalidArgument("resample_stage: cannot resample float64 audio (this path is float32)")
        n = int(b.shape[-1])
        lead = prod(b.shape[:-1])
        n_out = st.out_length(n)
        out = b.empty(tuple(b.shape[:-1]) + (n_out,))
        if b.device:
            with b.device_guard():
                check(lib.smx_resample_stage_apply_f32_dev(st._h, b.ptr(), lead, n, n, out_ptr(out), n_out, b.stream()))
            return out
        check(lib.smx_resample_stage_apply_f32(st._h, b.ptr(), lead, n, out_ptr(out)))
        return b.wrap(out)


class Kernel:
    """``Resample.Kernel`` (resample.mli:270-319) of one pure xL or /M stage: ``prepare`` / ``step`` / ``flush`` / ``reset``.
    One state carries all channels; the block carry lives on the device.  ``step`` returns the newly computable samples
    ``[channels; k]`` or None (burst emission: whole block pairs), ``flush`` the tail or None; the concatenation of every
    step plus flush equals ``Stage.apply`` on the concatenated input bit for bit.  Host chunks give host arrays,
    device-resident (torch CUDA) chunks stay on the device."""

    def __init__(self, handle, stage, channels, max_block):
        self._h, self._stage, self.channels, self.max_block = handle, stage, channels, max_block

    @staticmethod
    def prepare(stage: Stage, channels: int, max_block: int) -> "Kernel":
        handle = C.c_void_p()
        check(lib.smx_resample_kernel_prepare(stage._h, int(channels), int(max_block), C.byref(handle)))
        return Kernel(handle, stage, int(channels), int(max_block))     # (the stage must outlive the kernel: held here)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            try:
                lib.smx_resample_kernel_destroy(h)
            except Exception:
                pass

    def reset(self) -> None:
        check(lib.smx_resample_kernel_reset(self._h))

    def _check(self, shape, what):
        lead = prod(shape[:-1]) if len(shape) > 1 else 1
        if len(shape) < 1 or lead != self.channels:      # resample.mli:309-311
            raise _lib.InvalidArgument("%s: cannot feed a chunk of shape %s to a kernel of %d channels (the leading axes must hold "
                                       "the channels)" % (what, tuple(shape), self.channels))

    def step(self, chunk):
        from ._tensor import is_device, is_torch
        self._check(tuple(chunk.shape), "step")
        n = int(chunk.shape[-1])
        got = C.c_int64()
        bound = max(1, lib.smx_resample_kernel_out_bound(self._h, n))
        if is_device(chunk):
            import torch
            x = chunk.to(torch.float32).reshape(self.channels, n).contiguous()
            out = torch.empty((self.channels, bound), device=chunk.device, dtype=torch.float32)
            with torch.cuda.device(chunk.device):
                stream = C.c_void_p(torch.cuda.current_stream(chunk.device).cuda_stream)
                check(lib.smx_resample_kernel_step_f32_dev(self._h, C.c_void_p(x.data_ptr()), n, max(n, 1), C.c_void_p(out.data_ptr()),
                                                           bound, C.byref(got), stream))
            return None if got.value == 0 else out[:, :got.value].contiguous()
        a = chunk.detach().cpu().numpy() if is_torch(chunk) else np.asarray(chunk)
        if a.dtype != np.float32:
            raise _lib.InvalidArgument("step: cannot resample float64 audio (this path is float32)")
        a = np.ascontiguousarray(a).reshape(self.channels, n)
        out = np.empty((self.channels, bound), dtype=np.float32)
        check(lib.smx_resample_kernel_step_f32(self._h, C.c_void_p(a.ctypes.data), n, max(n, 1), C.c_void_p(out.ctypes.data), bound,
                                               C.byref(got)))
        return None if got.value == 0 else np.ascontiguousarray(out[:, :got.value])

    def flush(self, device=None):
        """The delayed tail (None when there is none, and on a second flush).  ``device``: a torch device to receive it there."""
        got = C.c_int64()
        pending = lib.smx_resample_kernel_pending(self._h)
        cap = max(1, pending)
        if device is not None:
            import torch
            out = torch.empty((self.channels, cap), device=device, dtype=torch.float32)
            with torch.cuda.device(device):
                stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
                check(lib.smx_resample_kernel_flush_f32_dev(self._h, C.c_void_p(out.data_ptr()), cap, C.byref(got), stream))
            return None if got.value == 0 else out[:, :got.value].contiguous()
        out = np.empty((self.channels, cap), dtype=np.float32)
        check(lib.smx_resample_kernel_flush_f32(self._h, C.c_void_p(out.ctypes.data), cap, C.byref(got)))
        return None if got.value == 0 else np.ascontiguousarray(out[:, :got.value])

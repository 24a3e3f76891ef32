"""Tensor plumbing between callers and the C ABI.

numpy arrays (and CPU torch tensors) take the host entry points of the library
(upload / run / download inside the C ABI, what an nx-tensor caller gets);
CUDA/HIP torch tensors take the ``*_dev`` entry points on torch's current stream
and stay device-resident.  torch is only used for device memory and streams.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

try:  # torch is optional plumbing: host-array callers never need it
    import torch
except Exception:  # pragma: no cover
    torch = None


def is_torch(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


def is_device(x) -> bool:
    return is_torch(x) and x.is_cuda


class Batch:
    """A real tensor viewed as [lead; n] (time axis last, leading axes flattened)."""

    def __init__(self, x, op: str, min_rank: int = 1, what: str = "analyse"):
        self.torch = is_torch(x)
        self.device = is_device(x)
        shape = tuple(x.shape)
        if len(shape) < min_rank:
            if min_rank == 1:   # stft.ml:289-293 check_rank
                raise _lib.InvalidArgument(
                    "%s: cannot analyse a rank-zero tensor (the time axis must exist)" % op)
        self.shape = shape
        if self.device:
            if x.dtype not in (torch.float32, torch.float64):
                x = x.to(torch.float32)
            self.data = x.contiguous()
            self.bytes = 4 if self.data.dtype == torch.float32 else 8
        else:
            a = x.detach().cpu().numpy() if self.torch else np.asarray(x)
            if a.dtype not in (np.float32, np.float64):
                a = a.astype(np.float32)
            self.data = np.ascontiguousarray(a)
            self.bytes = self.data.dtype.itemsize

    def ptr(self):
        if self.device:
            return C.c_void_p(self.data.data_ptr())
        return C.c_void_p(self.data.ctypes.data)

    def empty(self, shape, complex_=False, overwritten=False):
        """Fresh output tensor of the caller's kind; complex outputs get the
        matching complex dtype (stft.ml:681-685 spectrum_witness).  overwritten: the library
        writes every element (a large host result may then come from its page-locked pool)."""
        if self.device:
            if complex_:
                dt = torch.complex64 if self.bytes == 4 else torch.complex128
            else:
                dt = self.data.dtype
            return torch.empty(shape, dtype=dt, device=self.data.device)
        if complex_:
            dt = np.complex64 if self.bytes == 4 else np.complex128
        else:
            dt = self.data.dtype
        if overwritten:
            return _lib.host_result(shape, dt)
        return np.zeros(shape, dtype=dt)

    def wrap(self, out):
        """Return host results in the caller's container type."""
        if self.torch and not self.device:
            return torch.from_numpy(out)
        return out

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.data.device).cuda_stream)

    def device_guard(self):
        return torch.cuda.device(self.data.device)


def out_ptr(out):
    if is_torch(out):
        return C.c_void_p(out.data_ptr())
    return C.c_void_p(out.ctypes.data)


def prod(shape) -> int:
    n = 1
    for d in shape:
        n *= int(d)
    return n


class Stateless:
    """``Pipeline.stateless f`` (the memoryless stage of the reference): every chunk maps through ``f`` on its own;
    nothing is withheld, so ``flush`` is empty and ``reset`` does nothing."""
    latency = 0

    def __init__(self, f):
        self._f = f

    def step(self, chunk):
        return self._f(chunk)

    __call__ = step

    def flush(self):
        return []

    def reset(self):
        pass

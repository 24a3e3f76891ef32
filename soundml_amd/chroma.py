"""``Soundml.Chroma`` over a linear-frequency spectrum on MI355X (reference: soundml/lib/chroma.ml:95-317).

    c = Chroma.Config.create(sample_rate=22050, fft_size=2048)       # n_chroma=12, octave envelope, C-based
    w = Chroma.filterbank(np.float64, c)                              # [n_chroma; bins] copy
    y = Chroma.apply(c, s, norm="inf")                                # [...; bins; frames] -> [...; n_chroma; frames]

The float64 projection matrix is built once on the host (chroma.ml:109-175); the product and the per-frame
normalisation run in float64 on the device with one rounding to the dtype of ``s``.  The constant-Q side of
the reference's module (``of_cqt``) is out of scope: there is no CQT here.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import check, lib
from ._tensor import Batch, out_ptr, prod

NORM_NONE, NORM_INF, NORM_P = 0, 1, 2


def norm_args(norm):
    """``?norm:[`Inf | `P of float | `None]`` as the ABI's (kind, exponent): "inf" | None | a positive float."""
    if norm is None or norm == "none":
        return NORM_NONE, 0.0
    if norm == "inf" or norm == float("inf"):
        return NORM_INF, 0.0
    return NORM_P, float(norm)


class Config:
    """``Chroma.Config.t`` (chroma.ml:95-257)."""

    def __init__(self, handle, params):
        self._h = handle
        self._params = params

    @staticmethod
    def create(sample_rate: int, fft_size: int, n_chroma: int = 12, tuning: float = 0.0, ctroct: float = 5.0,
               octwidth: Optional[float] = 2.0, base_c: bool = True) -> "Config":
        handle = C.c_void_p()
        check(lib.smx_chroma_config_create(int(n_chroma), float(tuning), float(ctroct), 0 if octwidth is None else 1,
                                           0.0 if octwidth is None else float(octwidth), 1 if base_c else 0,
                                           int(sample_rate), int(fft_size), C.byref(handle)))
        return Config(handle, dict(n_chroma=int(n_chroma), tuning=float(tuning), ctroct=float(ctroct),
                                   octwidth=None if octwidth is None else float(octwidth), base_c=bool(base_c),
                                   sample_rate=int(sample_rate), fft_size=int(fft_size)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            try:
                lib.smx_chroma_config_destroy(h)
            except Exception:
                pass

    n_chroma = property(lambda self: lib.smx_chroma_config_n_chroma(self._h))
    bins = property(lambda self: lib.smx_chroma_config_bins(self._h))
    fft_size = property(lambda self: lib.smx_chroma_config_fft_size(self._h))
    tuning = property(lambda self: self._params["tuning"])
    ctroct = property(lambda self: self._params["ctroct"])
    octwidth = property(lambda self: self._params["octwidth"])
    base_c = property(lambda self: self._params["base_c"])
    sample_rate = property(lambda self: self._params["sample_rate"])

    def __eq__(self, other):   # chroma.ml:242-257 equal
        return isinstance(other, Config) and self._params == other._params

    __hash__ = None

    def __repr__(self):        # chroma.ml:229-240 pp
        p = self._params
        return ("chroma(n_chroma=%d, sample_rate=%d, fft_size=%d, tuning=%g, ctroct=%g, octwidth=%s, base_c=%s)"
                % (p["n_chroma"], p["sample_rate"], p["fft_size"], p["tuning"], p["ctroct"],
                   "none" if p["octwidth"] is None else "%g" % p["octwidth"], "true" if p["base_c"] else "false"))


def filterbank(dtype, c: Config) -> np.ndarray:
    """``Chroma.filterbank dtype c`` (chroma.ml:259): a fresh copy of the weights."""
    out = np.empty((c.n_chroma, c.bins), dtype=np.float64)
    check(lib.smx_chroma_filterbank(c._h, C.c_void_p(out.ctypes.data)))
    return out.astype(dtype)


def apply(c: Config, s, norm="inf"):
    """``Chroma.apply ?norm c s`` (chroma.ml:285-317)."""
    kind, p = norm_args(norm)
    if kind == NORM_P and not (p > 0.0 and p != float("inf")):     # chroma.ml:30-40, before the tensor is looked at
        check(lib.smx_chroma_apply_f32(c._h, None, 0, c.bins, 0, kind, p, None))
    nd = len(s.shape)
    if nd < 2:
        raise _lib.InvalidArgument(
            "apply: cannot project a rank-%d tensor (the projection needs [...; bins; frames])" % nd)
    b = Batch(s, "apply")
    bins, frames = int(b.shape[-2]), int(b.shape[-1])
    lead_shape = b.shape[:-2]
    lead = prod(lead_shape)
    if bins != c.bins:
        check(lib.smx_chroma_apply_f32(c._h, None, lead, bins, frames, kind, p, None))
    out = b.empty(lead_shape + (c.n_chroma, frames))
    if b.device:
        if b.bytes != 4:
            raise _lib.Failure("apply: device-resident float64 spectrograms are not supported; pass a host array")
        if out.numel() > 0:
            out.zero_()
        with b.device_guard():
            check(lib.smx_chroma_apply_f32_dev(c._h, b.ptr(), lead, bins, frames, kind, p, out_ptr(out), b.stream()))
        return out
    fn = lib.smx_chroma_apply_f32 if b.bytes == 4 else lib.smx_chroma_apply_f64
    check(fn(c._h, b.ptr(), lead, bins, frames, kind, p, out_ptr(out)))
    return b.wrap(out)

"""``Soundml.Mel`` on MI355X (reference: soundml/lib/mel.mli:123-148, mel.ml:22-233).

    m = Mel.Config.create(n_mels=128, sample_rate=48000, fft_size=2048)
    w = Mel.filterbank(np.float64, m)        # [n_mels; bins] copy
    y = Mel.apply(m, s)                      # [...; bins; frames] -> [...; n_mels; frames]

The float64 weights are built once on the host (mel.ml:67-117) and uploaded; the
product runs on the fp32 MFMA (float32) or in float64 VALU (float64).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import check, lib
from ._tensor import Batch, out_ptr, prod


class Config:
    """``Mel.Config.t`` (mel.ml:22-164)."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def create(n_mels: int, sample_rate: int, fft_size: int, f_min: float = 0.0,
               f_max: Optional[float] = None, scale: str = "slaney", norm: str = "slaney") -> "Config":
        if scale not in _lib.MEL_SCALE:
            raise _lib.InvalidArgument("create: unknown mel scale %r" % scale)
        if norm not in _lib.MEL_NORM:
            raise _lib.InvalidArgument("create: unknown mel norm %r" % norm)
        handle = C.c_void_p()
        check(lib.smx_mel_config_create(int(n_mels), int(sample_rate), int(fft_size), float(f_min),
                                        0 if f_max is None else 1, 0.0 if f_max is None else float(f_max),
                                        _lib.MEL_SCALE[scale], _lib.MEL_NORM[norm], C.byref(handle)))
        return Config(handle)

    @staticmethod
    def from_weights(weights, fft_size: int) -> "Config":
        """A projection with caller-supplied weights [rows; fft_size // 2 + 1] (e.g. the reference's chroma
        filters, chroma.ml:307): `apply` and `mel_spectrogram` then compute W @ spectrogram with it."""
        w = np.ascontiguousarray(np.asarray(weights, dtype=np.float64))
        if w.ndim != 2 or w.shape[1] != int(fft_size) // 2 + 1:
            raise _lib.InvalidArgument("from_weights: cannot use weights of shape %s with an FFT of size %d (the "
                                       "bin axis must hold fft_size / 2 + 1 values)" % (list(w.shape), int(fft_size)))
        handle = C.c_void_p()
        check(lib.smx_mel_config_from_weights(int(w.shape[0]), int(fft_size), C.c_void_p(w.ctypes.data), C.byref(handle)))
        return Config(handle)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            try:
                lib.smx_mel_config_destroy(h)
            except Exception:
                pass

    n_mels = property(lambda self: lib.smx_mel_config_n_mels(self._h))
    bins = property(lambda self: lib.smx_mel_config_bins(self._h))
    fft_size = property(lambda self: lib.smx_mel_config_fft_size(self._h))
    f_max = property(lambda self: lib.smx_mel_config_f_max(self._h))


def filterbank(dtype, c: Config) -> np.ndarray:
    """``Mel.filterbank dtype c`` (mel.ml:198-200): a fresh copy of the weights."""
    out = np.empty((c.n_mels, c.bins), dtype=np.float64)
    check(lib.smx_mel_filterbank(c._h, C.c_void_p(out.ctypes.data)))
    return out.astype(dtype)


def apply(c: Config, s):
    """``Mel.apply c s`` (mel.ml:202-231)."""
    nd = len(s.shape)
    if nd < 2:
        raise _lib.InvalidArgument(
            "apply: cannot project a rank-%d tensor (the mel projection needs [...; bins; frames])" % nd)
    b = Batch(s, "apply")
    lead_shape = b.shape[:-2]
    bins, frames = int(b.shape[-2]), int(b.shape[-1])
    lead = prod(lead_shape)
    out = b.empty(lead_shape + (c.n_mels, frames))
    sfx = "f32" if b.bytes == 4 else "f64"
    if b.device:
        with b.device_guard():
            fn = getattr(lib, "smx_mel_apply_%s_dev" % sfx)
            check(fn(c._h, b.ptr(), lead, bins, frames, out_ptr(out), b.stream()))
        return out
    fn = getattr(lib, "smx_mel_apply_%s" % sfx)
    check(fn(c._h, b.ptr(), lead, bins, frames, out_ptr(out)))
    return b.wrap(out)


def stage(c: Config):
    """``Mel.stage c`` (mel.ml:233): ``apply c`` as a memoryless Pipeline stage."""
    from ._tensor import Stateless
    return Stateless(lambda s: apply(c, s))

"""``Soundml.Convert`` (convert.ml): the decibel conversions (device) and the mel-scale maps (host scalars).

    Convert.power_to_db(s, top_db=80.0)          # see features.power_to_db
    Convert.hz_to_mel(f, scale="slaney")         # float64 host values, shape preserved
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib
from .features import amplitude_to_db, power_to_db  # noqa: F401


def _scale_map(fn, values, scale):
    if scale not in _lib.MEL_SCALE:
        raise _lib.InvalidArgument("%s: unknown mel scale %r" % (fn.__name__, scale))
    v = np.ascontiguousarray(np.asarray(values, dtype=np.float64))
    out = np.empty_like(v)
    check(fn(_lib.MEL_SCALE[scale], C.c_void_p(v.ctypes.data), int(v.size), C.c_void_p(out.ctypes.data)))
    src = np.asarray(values)
    return out.astype(src.dtype) if src.dtype in (np.float32, np.float64) else out


def hz_to_mel(f, scale: str = "slaney"):
    """``Convert.hz_to_mel ?scale f`` (convert.ml:80-90)."""
    return _scale_map(lib.smx_hz_to_mel, f, scale)


def mel_to_hz(m, scale: str = "slaney"):
    """``Convert.mel_to_hz ?scale m`` (convert.ml:92-102)."""
    return _scale_map(lib.smx_mel_to_hz, m, scale)

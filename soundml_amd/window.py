"""``Soundml.Window.make`` for the generalized-cosine families (window.ml:374-405)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


def make(dtype, kind: str, n: int, periodic: bool = True) -> np.ndarray:
    if kind not in _lib.WINDOW or kind == "custom":
        raise _lib.InvalidArgument("make: unknown window family %r" % kind)
    out = np.empty(max(int(n), 1), dtype=np.float64)
    check(lib.smx_window_make(_lib.WINDOW[kind], 1 if periodic else 0, int(n), C.c_void_p(out.ctypes.data)))
    return out[:n].astype(dtype)

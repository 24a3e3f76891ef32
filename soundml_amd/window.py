"""``Soundml.Window.make`` (window.ml:374-405): all eleven families of ``Window.t``, built on the host in float64.

    Window.make(np.float64, "hann", 2048)                       # periodic (DFT-even) by default
    Window.make(np.float32, ("kaiser", 8.6), 1024, periodic=False)
    Window.make(np.float64, ("tukey", 0.25), 400)

A family is a name, or (name, shape parameter) for "kaiser" (beta), "gaussian" (standard deviation in samples) and
"tukey" (taper fraction in [0, 1])."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


def family(kind):
    """(name, parameter) of a window specification; the parametric families need their parameter."""
    name, param = (kind, None) if isinstance(kind, str) else (kind[0], float(kind[1]))
    if name not in _lib.WINDOW or name == "custom":
        raise _lib.InvalidArgument("make: unknown window family %r" % (name,))
    if name in _lib.WINDOW_PARAMETRIC and param is None:
        raise _lib.InvalidArgument("make: the %s window needs its shape parameter: pass (%r, value)" % (name, name))
    return name, param


def make(dtype, kind, n: int, periodic: bool = True) -> np.ndarray:
    name, param = family(kind)
    out = np.empty(max(int(n), 1), dtype=np.float64)
    check(lib.smx_window_make_param(_lib.WINDOW[name], 0.0 if param is None else param, 1 if periodic else 0, int(n),
                                    C.c_void_p(out.ctypes.data)))
    return out[:n].astype(dtype)


def cola(kind, length: int, hop: int) -> bool:
    """``Window.cola w ~length ~hop`` (window.ml:407-434): the periodic window's shifts by ``hop`` sum to a constant."""
    name, param = family(kind)
    flag = C.c_int()
    check(lib.smx_window_cola(_lib.WINDOW[name], 0.0 if param is None else param, int(length), int(hop), C.byref(flag)))
    return bool(flag.value)

"""ctypes access to the C restatement of the oracle (oracle/oracle_stft.c).
Test infrastructure only: used by tests/ and by bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import soundml_oracle as O

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_PAD = {"reflect": 0, "constant": 1, "edge": 2}


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    i64, vp, ci, d = C.c_int64, C.c_void_p, C.c_int, C.c_double
    for name in ("oracle_stft_f32", "oracle_stft_f64"):
        fn = getattr(lib, name)
        fn.restype = ci
        fn.argtypes = [vp, i64, i64, ci, ci, vp, i64, ci, d, i64, ci, d, vp, ci]
    lib.oracle_mel_apply_f32.restype = ci
    lib.oracle_mel_apply_f32.argtypes = [vp, ci, ci, vp, i64, i64, vp]
    lib.oracle_resample_shape.restype = ci
    lib.oracle_resample_shape.argtypes = [vp, vp, vp, i64, i64, i64, i64]
    lib.oracle_mel_apply_f32_mt.restype = ci
    lib.oracle_mel_apply_f32_mt.argtypes = [vp, ci, ci, vp, i64, i64, vp, ci]
    return lib


def effective_cpus() -> int:
    """CPUs this process may really use: the affinity mask capped by the cgroup's CPU quota (the GPU boxes show 256 logical
    CPUs and a quota of 16: more threads than that only buy throttling)."""
    import math
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, math.ceil(quota / period)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def stft(c: O.StftConfig, x: np.ndarray, power=2.0, complex_out=False, threads=1, out=None) -> np.ndarray:
    """power_spectrum / transform of the C oracle for power-of-two fft sizes.  `out`: a C-contiguous array of the result's
    shape and dtype to fill (bench.py's cpu_baseline times the call on pages that are already mapped)."""
    lib = load()
    x = np.ascontiguousarray(x)
    assert x.dtype in (np.float32, np.float64)
    lead_shape, n = x.shape[:-1], x.shape[-1]
    lead = int(np.prod(lead_shape)) if lead_shape else 1
    count = O.frames(c, n)
    f32 = x.dtype == np.float32
    shape = lead_shape + (c.bins, count)
    dtype = (np.complex64 if f32 else np.complex128) if complex_out else x.dtype
    if out is None:
        out = np.zeros(shape, dtype=dtype)
    assert out.shape == shape and out.dtype == dtype and out.flags["C_CONTIGUOUS"]
    w = np.ascontiguousarray(c.analysis_window, dtype=np.float64)
    fn = lib.oracle_stft_f32 if f32 else lib.oracle_stft_f64
    rc = fn(x.ctypes.data, lead, n, c.fft_size, c.hop, w.ctypes.data, O.left_width(c), _PAD[c.pad],
            float(c.pad_value), count, 1 if complex_out else 0, float(power), out.ctypes.data, int(threads))
    if rc != 0:
        raise ValueError("C oracle: unsupported fft size %d" % c.fft_size)
    return out


def mel_apply(mc: O.MelConfig, s: np.ndarray, threads: int = 1) -> np.ndarray:
    lib = load()
    s = np.ascontiguousarray(s, dtype=np.float32)
    lead_shape, frames = s.shape[:-2], s.shape[-1]
    lead = int(np.prod(lead_shape)) if lead_shape else 1
    out = np.zeros(lead_shape + (mc.n_mels, frames), dtype=np.float32)
    w = np.ascontiguousarray(mc.weights, dtype=np.float64)
    lib.oracle_mel_apply_f32_mt(w.ctypes.data, mc.n_mels, mc.bins, s.ctypes.data, lead, frames, out.ctypes.data,
                                int(threads))
    return out


_REF_SO = os.path.join(_HERE, "_ref", "libresample_shape_ref.so")


def have_ref() -> bool:
    """oracle/_ref/libresample_shape_ref.so: the reference's own `soundml_resample_shape_run`, compiled from
    /root/reference/soundml/lib/resample_stubs.c where it lies (oracle/Makefile; prebuilt when the reference is absent)"""
    if not os.path.exists(_REF_SO) and os.path.exists("/root/reference/soundml/lib/resample_stubs.c"):
        build()
    return os.path.exists(_REF_SO)


def ref_resample_shape(xs: np.ndarray, h: np.ndarray, n: int, sl: int, sm: int) -> np.ndarray:
    """THE REFERENCE's shaping identity (resample_stubs.c:329-372, its own compiled code) on complex128 [lines; n/2+1]"""
    lib = C.CDLL(_REF_SO)
    lib.ref_resample_shape.restype = C.c_int
    lib.ref_resample_shape.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64]
    xs = np.ascontiguousarray(xs, dtype=np.complex128)
    h = np.ascontiguousarray(h, dtype=np.complex128)
    w = n * sl if sl > 1 else (n // sm if sm > 1 else n)
    out = np.zeros((xs.shape[0], w // 2 + 1), dtype=np.complex128)
    if lib.ref_resample_shape(xs.ctypes.data, h.ctypes.data, out.ctypes.data, xs.shape[0], n, sl, sm) != 0:
        raise ValueError("resample_shape: invalid geometry")
    return out


def resample_shape(xs: np.ndarray, h: np.ndarray, n: int, sl: int, sm: int) -> np.ndarray:
    """the C restatement of resample_stubs.c:329-372 on complex128 [lines; n/2+1]"""
    lib = load()
    xs = np.ascontiguousarray(xs, dtype=np.complex128)
    h = np.ascontiguousarray(h, dtype=np.complex128)
    w = n * sl if sl > 1 else (n // sm if sm > 1 else n)
    out = np.zeros((xs.shape[0], w // 2 + 1), dtype=np.complex128)
    rc = lib.oracle_resample_shape(xs.ctypes.data, h.ctypes.data, out.ctypes.data, xs.shape[0], n, sl, sm)
    if rc != 0:
        raise ValueError("resample_shape: invalid geometry")
    return out

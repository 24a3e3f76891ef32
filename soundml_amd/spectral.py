"""Spectral-shape features on MI355X (reference: soundml/lib/spectral.ml:171-255, re-exported flat as
``Soundml.spectral_centroid`` etc., soundml.ml:119-133).

    c = spectral_centroid(s, sample_rate=22050)          # [...; bins; frames] -> [...; 1; frames]
    b = spectral_bandwidth(s, sample_rate=22050, p=2.0, centroid=c)
    r = spectral_rolloff(s, sample_rate=22050, roll_percent=0.85)
    f = spectral_flatness(s, amin=1e-10, power=2.0)

One reduction along the bin axis in a float64 interior, one rounding to the dtype of ``s``; the magnitudes
must be non-negative (checked on the device, worded like the reference).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib
from ._tensor import Batch, out_ptr, prod


def _spectrogram(op: str, s):
    nd = len(s.shape)
    if nd < 2:   # spectral.ml:27-33
        raise _lib.InvalidArgument(
            "%s: cannot analyse a rank-%d tensor (a spectrogram is [...; bins; frames])" % (op, nd))
    b = Batch(s, op)
    bins, frames = int(b.shape[-2]), int(b.shape[-1])
    return b, b.shape[:-2], bins, frames


def _freqs(op: str, freqs):
    """The caller's grid as a host float64 vector (the reference casts it to float64, spectral.ml:119)."""
    if freqs is None:
        return None, None, 0
    f = freqs.detach().cpu().numpy() if hasattr(freqs, "detach") else np.asarray(freqs)
    if f.ndim != 1:   # spectral.ml:43-52
        raise _lib.InvalidArgument(
            "%s: cannot use a rank-%d freqs tensor (freqs is rank-one, one frequency per bin)" % (op, f.ndim))
    f = np.ascontiguousarray(f.astype(np.float64))
    return f, C.c_void_p(f.ctypes.data), int(f.shape[0])


def _run(op, b, lead_shape, bins, frames, host_fns, dev_fn, args_before, args_after):
    lead = prod(lead_shape)
    out = b.empty(lead_shape + (1, frames))
    if b.device:
        if b.bytes != 4:
            raise _lib.Failure("%s: device-resident float64 spectrograms are not supported; pass a host array" % op)
        if out.numel() > 0:
            out.zero_()
        with b.device_guard():
            check(dev_fn(b.ptr(), lead, bins, frames, *args_before, *args_after, out_ptr(out), b.stream()))
        return out
    fn = host_fns[0] if b.bytes == 4 else host_fns[1]
    check(fn(b.ptr(), lead, bins, frames, *args_before, *args_after, out_ptr(out)))
    return b.wrap(out)


def spectral_centroid(s, sample_rate: int, freqs=None):
    """``Soundml.spectral_centroid ?freqs ~sample_rate s`` (spectral.ml:171-177)."""
    op = "spectral_centroid"
    b, lead_shape, bins, frames = _spectrogram(op, s)
    f, fptr, nf = _freqs(op, freqs)
    return _run(op, b, lead_shape, bins, frames, (lib.smx_spectral_centroid_f32, lib.smx_spectral_centroid_f64),
                lib.smx_spectral_centroid_f32_dev, (), (fptr, nf, int(sample_rate)))


def spectral_bandwidth(s, sample_rate: int, p: float = 2.0, freqs=None, centroid=None):
    """``Soundml.spectral_bandwidth ?p ?freqs ?centroid ~sample_rate s`` (spectral.ml:179-218)."""
    op = "spectral_bandwidth"
    b, lead_shape, bins, frames = _spectrogram(op, s)
    f, fptr, nf = _freqs(op, freqs)
    cptr, c_rows, c_frames, keep = None, 0, 0, None
    if centroid is not None:
        if len(centroid.shape) < 2:   # spectral.ml:186-192
            raise _lib.InvalidArgument("%s: cannot reuse a rank-%d centroid (centroid must be [...; 1; frames])"
                                       % (op, len(centroid.shape)))
        c_rows, c_frames = int(centroid.shape[-2]), int(centroid.shape[-1])
        if c_rows == 1 and c_frames == frames:
            # same container and dtype as s, leading axes broadcast like the reference's subtraction
            if b.device:
                import torch
                keep = torch.broadcast_to(centroid.to(device=b.data.device, dtype=b.data.dtype),
                                          lead_shape + (1, frames)).contiguous()
                cptr = C.c_void_p(keep.data_ptr())
            else:
                c = centroid.detach().cpu().numpy() if hasattr(centroid, "detach") else np.asarray(centroid)
                keep = np.ascontiguousarray(np.broadcast_to(c.astype(b.data.dtype), lead_shape + (1, frames)))
                cptr = C.c_void_p(keep.ctypes.data)
        else:
            cptr = C.c_void_p(1)   # never dereferenced: the shape check fails first and words the error
    return _run(op, b, lead_shape, bins, frames, (lib.smx_spectral_bandwidth_f32, lib.smx_spectral_bandwidth_f64),
                lib.smx_spectral_bandwidth_f32_dev, (float(p),), (fptr, nf, cptr, c_rows, c_frames, int(sample_rate)))


def spectral_rolloff(s, sample_rate: int, roll_percent: float = 0.85, freqs=None):
    """``Soundml.spectral_rolloff ?roll_percent ?freqs ~sample_rate s`` (spectral.ml:220-243)."""
    op = "spectral_rolloff"
    b, lead_shape, bins, frames = _spectrogram(op, s)
    f, fptr, nf = _freqs(op, freqs)
    return _run(op, b, lead_shape, bins, frames, (lib.smx_spectral_rolloff_f32, lib.smx_spectral_rolloff_f64),
                lib.smx_spectral_rolloff_f32_dev, (float(roll_percent),), (fptr, nf, int(sample_rate)))


def spectral_flatness(s, amin: float = 1e-10, power: float = 2.0):
    """``Soundml.spectral_flatness ?amin ?power s`` (spectral.ml:245-255)."""
    op = "spectral_flatness"
    b, lead_shape, bins, frames = _spectrogram(op, s)
    return _run(op, b, lead_shape, bins, frames, (lib.smx_spectral_flatness_f32, lib.smx_spectral_flatness_f64),
                lib.smx_spectral_flatness_f32_dev, (float(amin), float(power)), ())


# ---- memoryless Pipeline stages (spectral.ml:257-284): every chunk-independent parameter is validated where the
# stage is built; the freqs / bins pairing and the non-negativity of the data are left to the first chunk ------------

def _g(v: float) -> str:
    return "%g" % v


def _check_sample_rate(op, sample_rate):
    if int(sample_rate) < 1:
        raise _lib.InvalidArgument("%s: cannot use a sample rate of %d Hz (sample_rate must be at least 1)" % (op, int(sample_rate)))


def _check_freqs_rank(op, freqs):
    if freqs is not None and len(np.shape(freqs)) != 1:
        raise _lib.InvalidArgument("%s: cannot use a rank-%d freqs tensor (freqs is rank-one, one frequency per bin)"
                                   % (op, len(np.shape(freqs))))


def _finite_positive(v) -> bool:
    v = float(v)
    return v > 0.0 and v != float("inf")


def spectral_centroid_stage(sample_rate: int, freqs=None):
    op = "spectral_centroid_stage"
    _check_sample_rate(op, sample_rate)
    _check_freqs_rank(op, freqs)
    from ._tensor import Stateless
    return Stateless(lambda s: spectral_centroid(s, sample_rate=sample_rate, freqs=freqs))


def spectral_bandwidth_stage(sample_rate: int, p: float = 2.0, freqs=None):
    op = "spectral_bandwidth_stage"
    if not _finite_positive(p):
        raise _lib.InvalidArgument("%s: cannot raise deviations to the power %s (p must be finite and positive)" % (op, _g(p)))
    _check_sample_rate(op, sample_rate)
    _check_freqs_rank(op, freqs)
    from ._tensor import Stateless
    return Stateless(lambda s: spectral_bandwidth(s, sample_rate=sample_rate, p=p, freqs=freqs))


def spectral_rolloff_stage(sample_rate: int, roll_percent: float = 0.85, freqs=None):
    op = "spectral_rolloff_stage"
    if not (0.0 < float(roll_percent) < 1.0):
        raise _lib.InvalidArgument("%s: cannot keep %s of the spectral energy (roll_percent must lie strictly between 0 "
                                   "and 1)" % (op, _g(roll_percent)))
    _check_sample_rate(op, sample_rate)
    _check_freqs_rank(op, freqs)
    from ._tensor import Stateless
    return Stateless(lambda s: spectral_rolloff(s, sample_rate=sample_rate, roll_percent=roll_percent, freqs=freqs))


def spectral_flatness_stage(amin: float = 1e-10, power: float = 2.0):
    op = "spectral_flatness_stage"
    if not _finite_positive(amin):
        raise _lib.InvalidArgument("%s: cannot floor the spectrum at %s (amin must be finite and positive)" % (op, _g(amin)))
    if not _finite_positive(power):
        raise _lib.InvalidArgument("%s: cannot raise magnitudes to the power %s (power must be finite and positive)" % (op, _g(power)))
    from ._tensor import Stateless
    return Stateless(lambda s: spectral_flatness(s, amin=amin, power=power))

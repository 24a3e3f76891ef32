"""Flat feature API: ``Soundml.mel_spectrogram`` / ``mfcc`` / ``chroma_stft`` (soundml.ml:12-107)."""
from __future__ import annotations

from . import _lib
from ._lib import check, lib
from ._tensor import Batch, out_ptr, prod
from . import stft as Stft


def mel_spectrogram(stft_config, mel_config, x, power: float = 2.0):
    """``Soundml.mel_spectrogram stft mel ?power x`` = Mel.apply mel (Stft.power_spectrum ~power stft x),
    computed by one fused device pass where the geometry allows."""
    if stft_config.fft_size != mel_config.fft_size:  # soundml.ml:12-20, before touching x
        raise _lib.InvalidArgument(
            "mel_spectrogram: cannot project a %d-point STFT through a filterbank built for an FFT of "
            "size %d (the two configurations must agree on fft_size)"
            % (stft_config.fft_size, mel_config.fft_size))
    b = Batch(x, "power_spectrum")
    n = int(b.shape[-1])
    lead_shape = b.shape[:-1]
    lead = prod(lead_shape)
    count = Stft.frames(stft_config, n)
    out = b.empty(lead_shape + (mel_config.n_mels, count))
    if b.device:
        if b.bytes != 4:
            from . import mel as Mel
            return Mel.apply(mel_config, Stft.power_spectrum(stft_config, x, power))
        with b.device_guard():
            check(lib.smx_mel_spectrogram_f32_dev(stft_config._h, mel_config._h, b.ptr(), lead, n, n,
                                                  float(power), out_ptr(out), b.stream()))
        return out
    fn = lib.smx_mel_spectrogram_f32 if b.bytes == 4 else lib.smx_mel_spectrogram_f64
    check(fn(stft_config._h, mel_config._h, b.ptr(), lead, n, float(power), out_ptr(out)))
    return b.wrap(out)


def mfcc(stft_config, mel_config, x, n_mfcc: int = 20, lifter=None):
    """``Soundml.mfcc stft mel ?n_mfcc ?lifter x`` (soundml.ml:50-95): log-mel spectrogram (80 dB clamp under the
    maximum of the whole tensor), orthonormal DCT-II along the mel axis, optional sinusoidal lifter;
    [...; n] -> [...; n_mfcc; frames] in x's dtype."""
    has_lifter = lifter is not None
    lift = float(lifter) if has_lifter else 0.0
    if stft_config.fft_size != mel_config.fft_size or not (1 <= int(n_mfcc) <= mel_config.n_mels) or \
            (has_lifter and not (lift >= 0.0 and lift != float("inf"))):
        # the reference's checks come before the tensor is looked at (soundml.ml:51-70): let the ABI word them
        check(lib.smx_mfcc_f32(stft_config._h, mel_config._h, None, 0, 0, int(n_mfcc), 1 if has_lifter else 0, lift, None))
    b = Batch(x, "power_spectrum")
    n = int(b.shape[-1])
    lead_shape = b.shape[:-1]
    lead = prod(lead_shape)
    count = Stft.frames(stft_config, n)
    out = b.empty(lead_shape + (int(n_mfcc), count))
    if b.device:
        if b.bytes != 4:
            raise _lib.Failure("mfcc: device-resident float64 audio is not supported; pass a host array")
        with b.device_guard():
            check(lib.smx_mfcc_f32_dev(stft_config._h, mel_config._h, b.ptr(), lead, n, n, int(n_mfcc),
                                       1 if has_lifter else 0, lift, out_ptr(out), b.stream()))
        return out
    fn = lib.smx_mfcc_f32 if b.bytes == 4 else lib.smx_mfcc_f64
    check(fn(stft_config._h, mel_config._h, b.ptr(), lead, n, int(n_mfcc), 1 if has_lifter else 0, lift, out_ptr(out)))
    return b.wrap(out)


def chroma_stft(stft_config, chroma_config, x, power: float = 2.0, norm="inf"):
    """``Soundml.chroma_stft stft chroma ?power ?norm x`` (soundml.ml:97-107) =
    Chroma.apply ?norm chroma (Stft.power_spectrum ~power stft x); [...; n] -> [...; n_chroma; frames]."""
    from . import chroma as Chroma
    kind, p = Chroma.norm_args(norm)
    if stft_config.fft_size != chroma_config.fft_size or (kind == Chroma.NORM_P and not (p > 0.0 and p != float("inf"))):
        # the reference's checks come before the tensor is looked at: let the ABI word them
        check(lib.smx_chroma_stft_f32(stft_config._h, chroma_config._h, None, 0, 0, float(power), kind, p, None))
    b = Batch(x, "power_spectrum")
    n = int(b.shape[-1])
    lead_shape = b.shape[:-1]
    lead = prod(lead_shape)
    count = Stft.frames(stft_config, n)
    out = b.empty(lead_shape + (chroma_config.n_chroma, count))
    if b.device:
        if b.bytes != 4:
            return Chroma.apply(chroma_config, Stft.power_spectrum(stft_config, x, power), norm)
        with b.device_guard():
            check(lib.smx_chroma_stft_f32_dev(stft_config._h, chroma_config._h, b.ptr(), lead, n, n, float(power),
                                              kind, p, out_ptr(out), b.stream()))
        return out
    fn = lib.smx_chroma_stft_f32 if b.bytes == 4 else lib.smx_chroma_stft_f64
    check(fn(stft_config._h, chroma_config._h, b.ptr(), lead, n, float(power), kind, p, out_ptr(out)))
    return b.wrap(out)


def _to_db(which, s, reference, amin, top_db):
    if not hasattr(s, "shape"):
        import numpy as np
        s = np.asarray(s)
    b = Batch(s, which, min_rank=0)
    total = prod(b.shape)
    out = b.empty(b.shape)
    has_top = top_db is not None
    args = (total, float(reference), float(amin), 1 if has_top else 0, float(top_db) if has_top else 0.0)
    if b.device:
        if b.bytes != 4:
            raise _lib.Failure("%s: device-resident float64 data is not supported; pass a host array" % which)
        with b.device_guard():
            check(getattr(lib, "smx_%s_f32_dev" % which)(b.ptr(), *args, out_ptr(out), b.stream()))
        return out
    fn = getattr(lib, "smx_%s_%s" % (which, "f32" if b.bytes == 4 else "f64"))
    check(fn(b.ptr(), *args, out_ptr(out)))
    return b.wrap(out)


def power_to_db(s, reference: float = 1.0, amin: float = 1e-10, top_db=None):
    """``Convert.power_to_db ?reference ?amin ?top_db s`` (convert.ml:52-56): 10 log10 of powers in s's own dtype,
    floored at amin, relative to reference, optionally clamped top_db under the maximum of the whole tensor."""
    return _to_db("power_to_db", s, reference, amin, top_db)


def amplitude_to_db(s, reference: float = 1.0, amin: float = 1e-5, top_db=None):
    """``Convert.amplitude_to_db`` (convert.ml:58-62): 20 log10 of magnitudes."""
    return _to_db("amplitude_to_db", s, reference, amin, top_db)

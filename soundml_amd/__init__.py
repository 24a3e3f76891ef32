"""soundml_amd -- MI355X-native spectral path for SoundML (STFT / power
spectrogram, mel filterbank, FIR block convolution) behind the reference's own
module names:

    from soundml_amd import Stft, Mel, Chroma, Window, Fir, mel_spectrogram, mfcc, chroma_stft, spectral_centroid

Everything computes in hand-written HIP kernels (gfx950) behind the C ABI of
``include/soundml_amd.h``; this package is the thin host mirror of
``Soundml.Stft`` / ``Soundml.Mel`` / ``Soundml.mel_spectrogram`` and fails loudly
if the HIP library is missing (no CPU fallback).
"""
from . import _lib
from ._lib import Failure, InvalidArgument, LIB_PATH, set_pinned_results, pinned_empty
from . import stft as Stft
from . import mel as Mel
from . import window as Window
from . import fir as Fir
from . import resample as Resample
from . import chroma as Chroma
from . import convert as Convert
from .features import mel_spectrogram, mfcc, chroma_stft, power_to_db, amplitude_to_db
from .spectral import (spectral_centroid, spectral_bandwidth, spectral_rolloff, spectral_flatness,
                       spectral_centroid_stage, spectral_bandwidth_stage, spectral_rolloff_stage, spectral_flatness_stage)
from . import shard


def set_interior(name: str) -> None:
    """"float32" (default, fast) or "float64" (the reference's interior, stft.ml:28-35)."""
    _lib.check(_lib.lib.smx_set_interior(_lib.INTERIOR[name]))


def set_scratch_retention(nbytes: int) -> None:
    """Bytes of freed scratch the library keeps per device for reuse (-1: the default, 1/8 of device memory)."""
    _lib.check(_lib.lib.smx_set_scratch_retention(int(nbytes)))


def device_count() -> int:
    import ctypes
    n = ctypes.c_int()
    _lib.check(_lib.lib.smx_device_count(ctypes.byref(n)))
    return n.value


__all__ = ["Stft", "Mel", "Chroma", "Convert", "Window", "Fir", "Resample", "mel_spectrogram", "mfcc", "chroma_stft", "power_to_db", "amplitude_to_db", "spectral_centroid",
           "spectral_bandwidth", "spectral_rolloff", "spectral_flatness", "shard", "set_interior", "set_pinned_results", "pinned_empty", "set_scratch_retention", "device_count",
           "InvalidArgument", "Failure", "LIB_PATH"]

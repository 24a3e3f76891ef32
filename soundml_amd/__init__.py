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


def set_devices(devices) -> None:
    """Clip sharding of the host-array batch calls over several GPUs from ONE process (the reference's caller is one process:
    stft.mli:211-250).  ``devices`` is a list of device ordinals (a device may appear more than once); ``Stft.transform`` /
    ``transform_range`` / ``power_spectrum`` / ``invert`` and ``mel_spectrogram`` on host arrays then split their leading
    axes into contiguous clip ranges (``shard.clip_range``), one host thread, staging ring pair and PCIe link per listed
    device, each writing its slice of the one result: bit-equal to the single-device call (stft_grid.ml:180-205).
    ``[]`` or ``None`` restores the single-device behaviour."""
    import ctypes
    ids = [int(d) for d in (devices or [])]
    arr = (ctypes.c_int * max(1, len(ids)))(*ids)
    _lib.check(_lib.lib.smx_set_devices(arr, len(ids)))


def get_devices():
    import ctypes
    n = ctypes.c_int()
    _lib.check(_lib.lib.smx_get_devices(None, 0, ctypes.byref(n)))
    arr = (ctypes.c_int * max(1, n.value))()
    _lib.check(_lib.lib.smx_get_devices(arr, n.value, ctypes.byref(n)))
    return [int(arr[i]) for i in range(n.value)]


__all__ = ["Stft", "Mel", "Chroma", "Convert", "Window", "Fir", "Resample", "mel_spectrogram", "mfcc", "chroma_stft", "power_to_db", "amplitude_to_db", "spectral_centroid",
           "spectral_bandwidth", "spectral_rolloff", "spectral_flatness", "shard", "set_interior", "set_pinned_results", "pinned_empty", "set_scratch_retention", "device_count", "set_devices", "get_devices",
           "InvalidArgument", "Failure", "LIB_PATH"]

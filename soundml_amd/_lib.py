"""ctypes binding of libsoundml_amd.so (the C ABI declared in include/soundml_amd.h).

The product path never computes on the CPU: if the HIP library is missing this
module raises at import time with build instructions, and every compute entry
point fails with ``Failure`` when no HIP device is visible.
"""
from __future__ import annotations

import ctypes as C
import os

# ONE HIP runtime per process: torch bundles its own libamdhip64 (same soname as
# /opt/rocm's).  If torch is going to be used it must be loaded BEFORE this
# library is dlopen'ed, so that our DT_NEEDED libamdhip64.so.7 resolves to the
# copy torch already mapped; two runtimes in one process cannot both see the GPU.
try:  # torch is optional plumbing (device tensors, streams); host-array callers work without it
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
# SOUNDML_AMD_LIB: another build of the same library (tools/ A/B timing of `make VARIANT=...` builds)
LIB_PATH = os.environ.get("SOUNDML_AMD_LIB") or os.path.join(_HERE, "lib", "libsoundml_amd.so")


class InvalidArgument(ValueError):
    """The reference's ``Invalid_argument``: a user-facing precondition failed.
    ``str(e)`` is the reference's message, verbatim."""


class Failure(RuntimeError):
    """The reference's ``Failure``: bookkeeping / runtime (HIP) error."""


SMX_OK, SMX_INVALID_ARGUMENT, SMX_FAILURE = 0, 1, 2
SMX_DEFAULT = -(2 ** 63)

ALIGNMENT = {"centered": 0, "left": 1, "right": 2}
PAD = {"reflect": 0, "constant": 1, "edge": 2}
SCALE = {"none": 0, "magnitude": 1, "psd": 2}
WINDOW = {"hann": 0, "rectangular": 1, "hamming": 2, "blackman": 3, "blackman_harris": 4,
          "nuttall": 5, "flat_top": 6, "bartlett": 7, "kaiser": 8, "gaussian": 9, "tukey": 10, "custom": 100}
WINDOW_PARAMETRIC = ("kaiser", "gaussian", "tukey")
MEL_SCALE = {"slaney": 0, "htk": 1}
MEL_NORM = {"slaney": 0, "none": 1}
INTERIOR = {"float32": 0, "float64": 1}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "soundml_amd: %s is missing. Build it with `python -c \"import __graft_entry__ as g; "
        "g.build()\"` or `make -C soundml_amd/csrc` (hipcc, gfx950). There is no CPU fallback."
        % LIB_PATH)

lib = C.CDLL(LIB_PATH)

i64, f64, cint, vp = C.c_int64, C.c_double, C.c_int, C.c_void_p
pf32, pf64, pi64 = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int64)

# name -> (restype, argtypes); mirrors include/soundml_amd.h one to one
SIGNATURES = {
    "smx_last_error": (C.c_char_p, []),
    "smx_version": (cint, []),
    "smx_debug_kernel_launches": (C.c_ulonglong, []),
    "smx_debug_stft_transform_frame_major_f32_dev": (cint, [vp, vp, i64, i64, vp, i64, i64, vp]),
    "smx_device_count": (cint, [C.POINTER(cint)]),
    "smx_set_device": (cint, [cint]),
    "smx_set_devices": (cint, [C.POINTER(cint), cint]),
    "smx_get_devices": (cint, [C.POINTER(cint), cint, C.POINTER(cint)]),
    "smx_shard_clip_range": (cint, [i64, i64, i64, pi64, pi64]),
    "smx_debug_staging_peak": (cint, [C.POINTER(cint), C.POINTER(cint), cint]),
    "smx_set_interior": (cint, [cint]),
    "smx_get_interior": (cint, []),
    "smx_synchronize": (cint, [vp]),
    "smx_host_alloc": (cint, [C.c_size_t, C.POINTER(vp)]),
    "smx_host_free": (cint, [vp]),
    "smx_window_make": (cint, [cint, cint, i64, vp]),
    "smx_stft_config_create": (cint, [i64, i64, i64, cint, cint, f64, cint, cint, vp, C.POINTER(vp)]),
    "smx_stft_config_destroy": (None, [vp]),
    "smx_stft_config_fft_size": (i64, [vp]),
    "smx_stft_config_hop": (i64, [vp]),
    "smx_stft_config_win_length": (i64, [vp]),
    "smx_stft_config_bins": (i64, [vp]),
    "smx_stft_config_left_width": (i64, [vp]),
    "smx_stft_config_right_width": (i64, [vp]),
    "smx_stft_config_latency": (i64, [vp]),
    "smx_stft_config_analysis_window": (cint, [vp, vp]),
    "smx_stft_frames": (cint, [vp, i64, pi64]),
    "smx_stft_first_complete": (cint, [vp, pi64]),
    "smx_stft_last_complete": (cint, [vp, i64, pi64]),
    "smx_stft_times": (cint, [vp, i64, i64, vp]),
    "smx_stft_frequencies": (cint, [vp, i64, vp]),
    "smx_stft_transform_f32": (cint, [vp, vp, i64, i64, vp]),
    "smx_stft_transform_f64": (cint, [vp, vp, i64, i64, vp]),
    "smx_stft_transform_range_f32": (cint, [vp, vp, i64, i64, i64, i64, vp]),
    "smx_stft_transform_range_f64": (cint, [vp, vp, i64, i64, i64, i64, vp]),
    "smx_stft_power_spectrum_f32": (cint, [vp, vp, i64, i64, f64, vp]),
    "smx_stft_power_spectrum_f64": (cint, [vp, vp, i64, i64, f64, vp]),
    "smx_stft_power_range_f32": (cint, [vp, vp, i64, i64, i64, i64, f64, vp]),
    "smx_stft_power_range_f64": (cint, [vp, vp, i64, i64, i64, i64, f64, vp]),
    "smx_stft_transform_range_f32_dev": (cint, [vp, vp, i64, i64, i64, i64, i64, vp, vp]),
    "smx_stft_transform_range_f64_dev": (cint, [vp, vp, i64, i64, i64, i64, i64, vp, vp]),
    "smx_stft_power_range_f32_dev": (cint, [vp, vp, i64, i64, i64, i64, i64, f64, vp, vp]),
    "smx_stft_power_range_f64_dev": (cint, [vp, vp, i64, i64, i64, i64, i64, f64, vp, vp]),
    "smx_stft_griffin_lim_f32": (cint, [vp, vp, i64, i64, i64, i64, f64, vp, cint, i64, vp]),
    "smx_stft_griffin_lim_f64": (cint, [vp, vp, i64, i64, i64, i64, f64, vp, cint, i64, vp]),
    "smx_stft_griffin_lim_f32_dev": (cint, [vp, vp, i64, i64, i64, i64, f64, vp, cint, i64, vp, vp]),
    "smx_mfcc_f32": (cint, [vp, vp, vp, i64, i64, i64, cint, f64, vp]),
    "smx_mfcc_f64": (cint, [vp, vp, vp, i64, i64, i64, cint, f64, vp]),
    "smx_mfcc_f32_dev": (cint, [vp, vp, vp, i64, i64, i64, i64, cint, f64, vp, vp]),
    "smx_spectral_centroid_f32": (cint, [vp, i64, i64, i64, vp, i64, i64, vp]),
    "smx_spectral_centroid_f64": (cint, [vp, i64, i64, i64, vp, i64, i64, vp]),
    "smx_spectral_centroid_f32_dev": (cint, [vp, i64, i64, i64, vp, i64, i64, vp, vp]),
    "smx_spectral_bandwidth_f32": (cint, [vp, i64, i64, i64, f64, vp, i64, vp, i64, i64, i64, vp]),
    "smx_spectral_bandwidth_f64": (cint, [vp, i64, i64, i64, f64, vp, i64, vp, i64, i64, i64, vp]),
    "smx_spectral_bandwidth_f32_dev": (cint, [vp, i64, i64, i64, f64, vp, i64, vp, i64, i64, i64, vp, vp]),
    "smx_spectral_rolloff_f32": (cint, [vp, i64, i64, i64, f64, vp, i64, i64, vp]),
    "smx_spectral_rolloff_f64": (cint, [vp, i64, i64, i64, f64, vp, i64, i64, vp]),
    "smx_spectral_rolloff_f32_dev": (cint, [vp, i64, i64, i64, f64, vp, i64, i64, vp, vp]),
    "smx_spectral_flatness_f32": (cint, [vp, i64, i64, i64, f64, f64, vp]),
    "smx_spectral_flatness_f64": (cint, [vp, i64, i64, i64, f64, f64, vp]),
    "smx_spectral_flatness_f32_dev": (cint, [vp, i64, i64, i64, f64, f64, vp, vp]),
    "smx_chroma_config_create": (cint, [i64, f64, f64, cint, f64, cint, i64, i64, C.POINTER(vp)]),
    "smx_chroma_config_destroy": (None, [vp]),
    "smx_chroma_config_n_chroma": (i64, [vp]),
    "smx_chroma_config_bins": (i64, [vp]),
    "smx_chroma_config_fft_size": (i64, [vp]),
    "smx_chroma_filterbank": (cint, [vp, vp]),
    "smx_chroma_apply_f32": (cint, [vp, vp, i64, i64, i64, cint, f64, vp]),
    "smx_chroma_apply_f64": (cint, [vp, vp, i64, i64, i64, cint, f64, vp]),
    "smx_chroma_apply_f32_dev": (cint, [vp, vp, i64, i64, i64, cint, f64, vp, vp]),
    "smx_chroma_stft_f32": (cint, [vp, vp, vp, i64, i64, f64, cint, f64, vp]),
    "smx_chroma_stft_f64": (cint, [vp, vp, vp, i64, i64, f64, cint, f64, vp]),
    "smx_chroma_stft_f32_dev": (cint, [vp, vp, vp, i64, i64, i64, f64, cint, f64, vp, vp]),
    "smx_stft_kernel_prepare_power": (cint, [vp, cint, i64, i64, f64, C.POINTER(vp)]),
    "smx_stft_stage_latency": (i64, [vp]),
    "smx_stft_frame_bound": (i64, [vp, i64]),
    "smx_set_scratch_retention": (cint, [i64]),
    "smx_power_to_db_f32": (cint, [vp, i64, f64, f64, cint, f64, vp]),
    "smx_power_to_db_f64": (cint, [vp, i64, f64, f64, cint, f64, vp]),
    "smx_power_to_db_f32_dev": (cint, [vp, i64, f64, f64, cint, f64, vp, vp]),
    "smx_amplitude_to_db_f32": (cint, [vp, i64, f64, f64, cint, f64, vp]),
    "smx_amplitude_to_db_f64": (cint, [vp, i64, f64, f64, cint, f64, vp]),
    "smx_amplitude_to_db_f32_dev": (cint, [vp, i64, f64, f64, cint, f64, vp, vp]),
    "smx_window_make_param": (cint, [cint, f64, cint, i64, vp]),
    "smx_window_cola": (cint, [cint, f64, i64, i64, C.POINTER(cint)]),
    "smx_hz_to_mel": (cint, [cint, vp, i64, vp]),
    "smx_mel_to_hz": (cint, [cint, vp, i64, vp]),
    "smx_stft_nola": (cint, [vp, C.POINTER(cint)]),
    "smx_stft_output_length": (cint, [vp, i64, pi64]),
    "smx_stft_invert_f32": (cint, [vp, vp, i64, i64, i64, cint, i64, vp]),
    "smx_stft_invert_f64": (cint, [vp, vp, i64, i64, i64, cint, i64, vp]),
    "smx_stft_invert_f32_dev": (cint, [vp, vp, i64, i64, i64, cint, i64, vp, vp]),
    "smx_stft_invert_f64_dev": (cint, [vp, vp, i64, i64, i64, cint, i64, vp, vp]),
    "smx_stft_kernel_prepare": (cint, [vp, cint, i64, i64, C.POINTER(vp)]),
    "smx_stft_kernel_destroy": (None, [vp]),
    "smx_stft_kernel_frame_bound": (cint, [vp, pi64]),
    "smx_stft_kernel_step": (cint, [vp, vp, i64, vp, i64, pi64]),
    "smx_stft_kernel_flush": (cint, [vp, vp, i64, pi64]),
    "smx_stft_kernel_reset": (cint, [vp]),
    "smx_stft_kernel_step_dev": (cint, [vp, vp, i64, i64, vp, i64, pi64, vp]),
    "smx_stft_kernel_flush_dev": (cint, [vp, vp, i64, pi64, vp]),
    "smx_stft_kernel_channels": (cint, [vp, pi64]),
    "smx_stft_kernel_set_channels": (cint, [vp, i64]),
    "smx_stft_kernel_config": (vp, [vp]),
    "smx_stft_synthesis_prepare": (cint, [vp, cint, i64, i64, C.POINTER(vp)]),
    "smx_stft_synthesis_destroy": (None, [vp]),
    "smx_stft_synthesis_latency": (i64, [vp]),
    "smx_stft_synthesis_sample_bound": (cint, [vp, pi64]),
    "smx_stft_synthesis_step": (cint, [vp, vp, i64, i64, vp, i64, pi64]),
    "smx_stft_synthesis_flush": (cint, [vp, vp, i64, pi64]),
    "smx_stft_synthesis_reset": (cint, [vp]),
    "smx_stft_synthesis_step_dev": (cint, [vp, vp, i64, i64, vp, i64, pi64, vp]),
    "smx_stft_synthesis_flush_dev": (cint, [vp, vp, i64, pi64, vp]),
    "smx_mel_config_create": (cint, [i64, i64, i64, f64, cint, f64, cint, cint, C.POINTER(vp)]),
    "smx_mel_config_from_weights": (cint, [i64, i64, vp, C.POINTER(vp)]),
    "smx_mel_config_destroy": (None, [vp]),
    "smx_mel_config_n_mels": (i64, [vp]),
    "smx_mel_config_bins": (i64, [vp]),
    "smx_mel_config_fft_size": (i64, [vp]),
    "smx_mel_config_f_max": (f64, [vp]),
    "smx_mel_filterbank": (cint, [vp, vp]),
    "smx_mel_apply_f32": (cint, [vp, vp, i64, i64, i64, vp]),
    "smx_mel_apply_f64": (cint, [vp, vp, i64, i64, i64, vp]),
    "smx_mel_apply_f32_dev": (cint, [vp, vp, i64, i64, i64, vp, vp]),
    "smx_mel_apply_f64_dev": (cint, [vp, vp, i64, i64, i64, vp, vp]),
    "smx_mel_spectrogram_f32": (cint, [vp, vp, vp, i64, i64, f64, vp]),
    "smx_mel_spectrogram_f64": (cint, [vp, vp, vp, i64, i64, f64, vp]),
    "smx_mel_spectrogram_f32_dev": (cint, [vp, vp, vp, i64, i64, i64, f64, vp, vp]),
    "smx_resample_ols_geom": (cint, [i64, i64, i64, i64, pi64, pi64, pi64, C.POINTER(cint)]),
    "smx_resample_prototype": (cint, [i64, i64, f64, f64, vp]),
    "smx_resample_shape_c128": (cint, [vp, vp, vp, i64, i64, i64, i64]),
    "smx_resample_shape_c128_dev": (cint, [vp, vp, vp, i64, i64, i64, i64, vp]),
    "smx_resample_stage_create": (cint, [vp, i64, i64, i64, C.POINTER(vp)]),
    "smx_resample_stage_destroy": (None, [vp]),
    "smx_resample_stage_out_length": (i64, [vp, i64]),
    "smx_resample_stage_apply_f32": (cint, [vp, vp, i64, i64, vp]),
    "smx_resample_stage_apply_f32_dev": (cint, [vp, vp, i64, i64, i64, vp, i64, vp]),
    "smx_resample_kernel_prepare": (cint, [vp, i64, i64, C.POINTER(vp)]),
    "smx_resample_kernel_destroy": (None, [vp]),
    "smx_resample_kernel_reset": (cint, [vp]),
    "smx_resample_kernel_out_bound": (i64, [vp, i64]),
    "smx_resample_kernel_pending": (i64, [vp]),
    "smx_resample_kernel_step_f32": (cint, [vp, vp, i64, i64, vp, i64, pi64]),
    "smx_resample_kernel_flush_f32": (cint, [vp, vp, i64, pi64]),
    "smx_resample_kernel_step_f32_dev": (cint, [vp, vp, i64, i64, vp, i64, pi64, vp]),
    "smx_resample_kernel_flush_f32_dev": (cint, [vp, vp, i64, pi64, vp]),
    "smx_fir_kaiser_beta": (cint, [f64, pf64]),
    "smx_fir_design_lowpass": (cint, [i64, f64, f64, vp]),
    "smx_fir_plan_create": (cint, [vp, i64, C.POINTER(vp)]),
    "smx_fir_plan_destroy": (None, [vp]),
    "smx_fir_plan_block": (i64, [vp]),
    "smx_fir_apply_f32": (cint, [vp, vp, i64, i64, vp]),
    "smx_fir_apply_f32_dev": (cint, [vp, vp, i64, i64, i64, vp, i64, vp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = the library lacks a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


def check(status: int) -> None:
    """Turn a status code into the reference's exception."""
    if status == SMX_OK:
        return
    message = (lib.smx_last_error() or b"").decode("utf-8", "replace")
    if status == SMX_INVALID_ARGUMENT:
        raise InvalidArgument(message)
    raise Failure(message)


# ---- result arrays of the host faces --------------------------------------------------------------------------------------
# The reference's faces return a fresh host tensor per call (stft.ml:356-364).  Large results come from the library's pool of
# page-locked blocks (smx_host_alloc): the DMA engine writes them directly, and a released block serves the next result --
# fresh pageable memory costs a page fault per 4 KB and a second pass over the bytes (include/soundml_amd.h).
PINNED_RESULT_MIN_BYTES = 32 << 20
_pinned_results = True


def set_pinned_results(flag: bool) -> None:
    """False: every host result is an ordinary numpy array again (np.zeros), e.g. where page-locked memory is scarce."""
    global _pinned_results
    _pinned_results = bool(flag)


class _PinnedBlock:
    """A block of smx_host_alloc as a buffer numpy can view; released to the library's pool when the last view dies."""

    def __init__(self, nbytes: int):
        p = vp()
        check(lib.smx_host_alloc(nbytes, C.byref(p)))
        self._ptr = p.value
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (self._ptr, False), "version": 3}

    def __del__(self):
        ptr, self._ptr = getattr(self, "_ptr", None), None
        if ptr:
            try:
                lib.smx_host_free(ptr)
            except Exception:   # interpreter shutdown
                pass


def host_result(shape, dtype):
    """A fresh host array for a result the library overwrites completely: page-locked when it is large."""
    import numpy as np
    dt = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
    if not _pinned_results or nbytes < PINNED_RESULT_MIN_BYTES:
        return np.zeros(shape, dtype=dt)
    try:
        block = _PinnedBlock(nbytes)
    except Failure:
        return np.zeros(shape, dtype=dt)
    return np.asarray(block).view(dt).reshape(shape)


def pinned_empty(shape, dtype):
    """An uninitialised numpy array over a block of the library's page-locked pool, whatever its size (smx_host_alloc): an INPUT
    batch kept in one is uploaded by the DMA engine directly, as a result in one is downloaded -- no staging copy on either side
    (tests/test_gpu_pinned_results.py: both directions).  Released to the pool when the array and its views are gone."""
    import numpy as np
    dt = np.dtype(dtype)
    nbytes = max(1, int(np.prod(shape, dtype=np.int64)) * dt.itemsize)
    return np.asarray(_PinnedBlock(nbytes))[:int(np.prod(shape, dtype=np.int64)) * dt.itemsize].view(dt).reshape(shape)

"""``Soundml.Stft`` (analysis half) on MI355X -- host-side mirror of the reference's
module interface over the C ABI (reference: soundml/lib/stft.mli:211-250,
435-471; stft.ml:48-691).  Same names, argument meaning and error behaviour:

    c = Stft.Config.create(fft_size=2048, hop=512)
    s = Stft.power_spectrum(c, x)            # [...; bins; frames], x's dtype
    z = Stft.transform(c, x)                 # complex64 / complex128
    z = Stft.transform_range(c, x, p0=, p1=) # frames [p0, p1)
    k = Stft.Kernel.prepare(c, dtype, channels=, max_block=); k.step(chunk); k.flush()

All arithmetic runs in the HIP kernels behind the C ABI; this file only maps
tensors to pointers and return codes to exceptions.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import check, lib
from ._tensor import Batch, out_ptr, prod


class Config:
    """``Stft.Config.t`` (stft.ml:48-129).  Immutable; owns the float64 analysis
    window and (lazily, per device) its device tables."""

    def __init__(self, handle, window_name):
        self._h = handle
        self._window_name = window_name

    @staticmethod
    def create(fft_size: int, window="hann", win_length: Optional[int] = None,
               hop: Optional[int] = None, alignment: str = "centered", pad="reflect",
               scale: str = "none") -> "Config":
        """``Stft.Config.create ?window ?win_length ?hop ?alignment ?pad ?scale ~fft_size ()``
        (stft.ml:61-111).  ``pad`` is "reflect", "edge" or ("constant", v);
        ``window`` is a family name or a float64 table of ``win_length`` points."""
        pad_value = 0.0
        if isinstance(pad, tuple):
            pad, pad_value = pad[0], float(pad[1])
        custom = None
        if isinstance(window, tuple) or (isinstance(window, str) and window in ("bartlett",) + _lib.WINDOW_PARAMETRIC):
            # the families the config's C face does not build itself: their float64 table (Window.make, periodic)
            from . import window as Window
            label = window if isinstance(window, str) else "%s(%g)" % (window[0], float(window[1]))
            window = Window.make(np.float64, window, int(fft_size) if win_length is None else int(win_length))
            if win_length is None:
                win_length = int(fft_size)
        else:
            label = None
        if isinstance(window, str):
            if window not in _lib.WINDOW:
                raise _lib.InvalidArgument("create: unknown window family %r" % window)
            kind, name = _lib.WINDOW[window], window
        else:
            table = np.ascontiguousarray(np.asarray(window, dtype=np.float64))
            if win_length is None:
                win_length = int(table.shape[0])
            if table.shape[0] != win_length:
                raise _lib.InvalidArgument("create: custom window table must have win_length points")
            custom, kind, name = table, _lib.WINDOW["custom"], (label or "custom")
        for value, table_, what in ((alignment, _lib.ALIGNMENT, "alignment"), (pad, _lib.PAD, "pad"),
                                    (scale, _lib.SCALE, "scale")):
            if value not in table_:
                raise _lib.InvalidArgument("create: unknown %s %r" % (what, value))
        handle = C.c_void_p()
        check(lib.smx_stft_config_create(
            int(fft_size), _lib.SMX_DEFAULT if win_length is None else int(win_length),
            _lib.SMX_DEFAULT if hop is None else int(hop), _lib.ALIGNMENT[alignment], _lib.PAD[pad],
            pad_value, _lib.SCALE[scale], kind,
            None if custom is None else C.c_void_p(custom.ctypes.data), C.byref(handle)))
        return Config(handle, name)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            try:
                lib.smx_stft_config_destroy(h)
            except Exception:
                pass

    # accessors (stft.ml:113-127)
    fft_size = property(lambda self: lib.smx_stft_config_fft_size(self._h))
    hop = property(lambda self: lib.smx_stft_config_hop(self._h))
    win_length = property(lambda self: lib.smx_stft_config_win_length(self._h))
    bins = property(lambda self: lib.smx_stft_config_bins(self._h))
    latency = property(lambda self: lib.smx_stft_config_latency(self._h))

    @property
    def analysis_window(self) -> np.ndarray:
        out = np.empty(self.fft_size, dtype=np.float64)
        check(lib.smx_stft_config_analysis_window(self._h, C.c_void_p(out.ctypes.data)))
        return out


def left_width(c: Config) -> int:
    return lib.smx_stft_config_left_width(c._h)


def right_width(c: Config) -> int:
    return lib.smx_stft_config_right_width(c._h)


def frames(c: Config, n: int) -> int:
    """stft.ml:217-223."""
    out = C.c_int64()
    check(lib.smx_stft_frames(c._h, int(n), C.byref(out)))
    return out.value


def first_complete(c: Config) -> int:
    out = C.c_int64()
    check(lib.smx_stft_first_complete(c._h, C.byref(out)))
    return out.value


def last_complete(c: Config, n: int) -> int:
    out = C.c_int64()
    check(lib.smx_stft_last_complete(c._h, int(n), C.byref(out)))
    return out.value


def times(dtype, c: Config, sample_rate: int, n: int) -> np.ndarray:
    """stft.ml:245-254."""
    count = frames(c, n) if (sample_rate >= 1 and n >= 0) else 0
    out = np.empty(max(count, 1), dtype=np.float64)
    check(lib.smx_stft_times(c._h, int(sample_rate), int(n), C.c_void_p(out.ctypes.data)))
    return out[:count].astype(dtype)


def frequencies(dtype, c: Config, sample_rate: int) -> np.ndarray:
    """stft.ml:256-261."""
    out = np.empty(c.bins, dtype=np.float64)
    check(lib.smx_stft_frequencies(c._h, int(sample_rate), C.c_void_p(out.ctypes.data)))
    return out.astype(dtype)


def _range(c: Config, x, p0, p1, power, op):
    b = Batch(x, op)
    n = b.shape[-1]
    lead_shape = b.shape[:-1]
    lead = prod(lead_shape)
    total = frames(c, n)
    if p0 is None:
        p0, p1 = 0, total
    complex_ = power is None
    count = max(0, p1 - p0)
    out = b.empty(lead_shape + (c.bins, count), complex_=complex_, overwritten=True)
    sfx = "f32" if b.bytes == 4 else "f64"
    if b.device:
        with b.device_guard():
            if complex_:
                fn = getattr(lib, "smx_stft_transform_range_%s_dev" % sfx)
                check(fn(c._h, b.ptr(), lead, n, n, p0, p1, out_ptr(out), b.stream()))
            else:
                fn = getattr(lib, "smx_stft_power_range_%s_dev" % sfx)
                check(fn(c._h, b.ptr(), lead, n, n, p0, p1, float(power), out_ptr(out), b.stream()))
        return out
    if complex_:
        fn = getattr(lib, "smx_stft_transform_range_%s" % sfx)
        check(fn(c._h, b.ptr(), lead, n, p0, p1, out_ptr(out)))
    else:
        if (p0, p1) == (0, total):
            fn = getattr(lib, "smx_stft_power_spectrum_%s" % sfx)
            check(fn(c._h, b.ptr(), lead, n, float(power), out_ptr(out)))
        else:
            fn = getattr(lib, "smx_stft_power_range_%s" % sfx)
            check(fn(c._h, b.ptr(), lead, n, p0, p1, float(power), out_ptr(out)))
    return b.wrap(out)


def transform(c: Config, x):
    """``Stft.transform cdtype c x`` (stft.ml:632-650): [...; n] -> complex [...; bins; frames].
    The complex storage follows the input's component width (``spectrum_witness``)."""
    return _range(c, x, None, None, None, "transform")


def transform_range(c: Config, x, p0: int, p1: int):
    """``Stft.transform_range cdtype c ~p0 ~p1 x`` (stft.ml:652-666)."""
    Batch(x, "transform_range")  # rank check first, as the reference does
    return _range(c, x, int(p0), int(p1), None, "transform_range")


def power_spectrum(c: Config, x, power: float = 2.0):
    """``Stft.power_spectrum ?power c x`` (stft.ml:687-691): |STFT|^power in x's dtype."""
    return _range(c, x, None, None, float(power), "power_spectrum")


def power_range(c: Config, x, p0: int, p1: int, power: float = 2.0):
    """Frames [p0, p1) of ``power_spectrum`` (host or device-resident audio): the seam clip / frame-range
    sharding uses."""
    return _range(c, x, int(p0), int(p1), float(power), "power_spectrum")


def nola(c: Config) -> bool:
    """``Stft.nola c`` (stft.ml:731-743): the overlap-added squared window clears 1e-10 of its maximum."""
    v = C.c_int()
    check(lib.smx_stft_nola(c._h, C.byref(v)))
    return bool(v.value)


def output_length(c: Config, frames: int) -> int:
    """``Stft.output_length c ~frames`` (stft.ml:792-796)."""
    v = C.c_int64()
    check(lib.smx_stft_output_length(c._h, int(frames), C.byref(v)))
    return v.value


def invert(c: Config, z, length=None):
    """``Stft.invert dtype c ?length z`` (stft.ml:902-939): complex [...; bins; frames] -> real [...; length],
    the least-squares synthesis.  complex64 spectra give float32 signals, complex128 float64; a device
    (torch) spectrum stays on the device."""
    from ._tensor import is_device, is_torch, torch
    shape = tuple(z.shape)
    if len(shape) < 2:  # stft.ml:760-766, before anything else
        raise _lib.InvalidArgument(
            "invert: cannot invert a rank-%d tensor (the bin and frame axes must exist)" % len(shape))
    bins, frames = int(shape[-2]), int(shape[-1])
    lead_shape = shape[:-2]
    lead = 1
    for d in lead_shape:
        lead *= int(d)
    has_length = length is not None
    if is_device(z):
        if z.dtype not in (torch.complex64, torch.complex128):
            z = z.to(torch.complex64)
        zc = z.contiguous()
        wide = zc.dtype == torch.complex128
        # the checks run inside the ABI before the output length is needed: ask it for them first with no data
        fn = lib.smx_stft_invert_f64_dev if wide else lib.smx_stft_invert_f32_dev
        if not has_length:
            check(fn(c._h, None, 0, bins, frames, 0, 0, None, None))
        out_len = int(length) if has_length else output_length(c, frames)
        if has_length and out_len < 0:
            check(fn(c._h, None, 0, bins, frames, 1, out_len, None, None))
        out = torch.zeros(lead_shape + (out_len,), dtype=torch.float64 if wide else torch.float32, device=zc.device)
        with torch.cuda.device(zc.device):
            stream = C.c_void_p(torch.cuda.current_stream(zc.device).cuda_stream)
            zr = torch.view_as_real(zc)
            check(fn(c._h, C.c_void_p(zr.data_ptr()), lead, bins, frames, 1 if has_length else 0,
                     out_len if has_length else 0, C.c_void_p(out.data_ptr()), stream))
        return out
    was_torch = is_torch(z)
    a = z.detach().cpu().numpy() if was_torch else np.asarray(z)
    if a.dtype not in (np.complex64, np.complex128):
        a = a.astype(np.complex64 if a.dtype == np.float32 else np.complex128)
    a = np.ascontiguousarray(a)
    wide = a.dtype == np.complex128
    fn = lib.smx_stft_invert_f64 if wide else lib.smx_stft_invert_f32
    if has_length and int(length) < 0 or not has_length:
        # run the checks (and nothing else: lead 0) so that the reference's errors come first
        check(fn(c._h, None, 0, bins, frames, 1 if has_length else 0, int(length) if has_length else 0, None))
    out_len = int(length) if has_length else output_length(c, frames)
    out = _lib.host_result(lead_shape + (out_len,), np.float64 if wide else np.float32)   # (the ABI copies the whole device result out: every element is written)
    check(fn(c._h, C.c_void_p(a.ctypes.data), lead, bins, frames, 1 if has_length else 0,
             out_len if has_length else 0, C.c_void_p(out.ctypes.data)))
    return torch.from_numpy(out) if was_torch else out


def griffin_lim(c: Config, s, n_iter: int = 32, momentum: float = 0.99, init=None, length=None):
    """``Stft.griffin_lim ?n_iter ?momentum ?init ?length c s`` (stft.ml:941-1017): a signal whose spectrogram
    magnitudes approach s [...; bins; frames].  ``init`` is an initial phase in radians (default: all-ones phase);
    float32 magnitudes give a float32 signal.  A device (torch) spectrogram stays on the device."""
    from ._tensor import is_device, is_torch, torch
    shape = tuple(s.shape)
    if len(shape) < 2:
        raise _lib.InvalidArgument(
            "griffin_lim: cannot invert a rank-%d tensor (the bin and frame axes must exist)" % len(shape))
    bins, frames = int(shape[-2]), int(shape[-1])
    lead_shape = shape[:-2]
    lead = 1
    for d in lead_shape:
        lead *= int(d)
    has_length = length is not None
    if init is not None and tuple(init.shape) != shape:   # stft.ml:978-986, worded there
        raise _lib.InvalidArgument(
            "griffin_lim: cannot start from a [%s] phase for a [%s] spectrogram (the initial phase must have the "
            "shape of the magnitudes)" % ("; ".join(map(str, init.shape)), "; ".join(map(str, shape))))
    args_tail = (int(n_iter), float(momentum))
    if is_device(s):
        sd = s.contiguous().to(torch.float32)
        pd = None if init is None else init.to(sd.device).contiguous().to(torch.float32)
        check(lib.smx_stft_griffin_lim_f32_dev(c._h, None, 0, bins, frames, *args_tail, None, 1 if has_length else 0,
                                               int(length) if has_length else 0, None, None))   # the checks first
        out_len = int(length) if has_length else output_length(c, frames)
        out = torch.zeros(lead_shape + (out_len,), dtype=torch.float32, device=sd.device)
        with torch.cuda.device(sd.device):
            stream = C.c_void_p(torch.cuda.current_stream(sd.device).cuda_stream)
            check(lib.smx_stft_griffin_lim_f32_dev(c._h, C.c_void_p(sd.data_ptr()), lead, bins, frames, *args_tail,
                                                   None if pd is None else C.c_void_p(pd.data_ptr()),
                                                   1 if has_length else 0, out_len if has_length else 0,
                                                   C.c_void_p(out.data_ptr()), stream))
        return out
    was_torch = is_torch(s)
    a = s.detach().cpu().numpy() if was_torch else np.asarray(s)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float32)
    a = np.ascontiguousarray(a)
    wide = a.dtype == np.float64
    p = None
    if init is not None:
        p = np.ascontiguousarray(np.asarray(init.detach().cpu().numpy() if is_torch(init) else init, dtype=a.dtype))
    fn = lib.smx_stft_griffin_lim_f64 if wide else lib.smx_stft_griffin_lim_f32
    check(fn(c._h, None, 0, bins, frames, *args_tail, None, 1 if has_length else 0, int(length) if has_length else 0, None))
    out_len = int(length) if has_length else output_length(c, frames)
    out = _lib.host_result(lead_shape + (out_len,), a.dtype)
    check(fn(c._h, C.c_void_p(a.ctypes.data), lead, bins, frames, *args_tail,
             None if p is None else C.c_void_p(p.ctypes.data), 1 if has_length else 0,
             out_len if has_length else 0, C.c_void_p(out.ctypes.data)))
    return torch.from_numpy(out) if was_torch else out


class Kernel:
    """``Stft.Kernel`` (stft.ml:597-622): streaming analysis with the carry held
    in device memory.  Chunks are host arrays [channels; m]; ``step`` / ``flush``
    return complex [channels; bins; k] or ``None``.  Mutable, single-owner."""

    def __init__(self, handle, cfg, dtype, channels):
        self._h, self._cfg, self._dtype, self._channels = handle, cfg, np.dtype(dtype), channels
        bound = C.c_int64()
        check(lib.smx_stft_kernel_frame_bound(self._h, C.byref(bound)))
        self.frame_bound = bound.value

    @staticmethod
    def prepare(c: Config, dtype, channels: int, max_block: int) -> "Kernel":
        handle = C.c_void_p()
        dt = np.dtype(dtype)
        check(lib.smx_stft_kernel_prepare(c._h, 8 if dt == np.float64 else 4, int(channels),
                                          int(max_block), C.byref(handle)))
        return Kernel(handle, c, np.float64 if dt == np.float64 else np.float32, int(channels))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            try:
                lib.smx_stft_kernel_destroy(h)
            except Exception:
                pass

    def _emit(self, call, capacity):
        cdt = np.complex128 if self._dtype == np.float64 else np.complex64
        if getattr(self, "_real", False):       # the power face of the state machine (power_stage)
            cdt = self._dtype
        out = np.zeros((self._channels, self._cfg.bins, capacity), dtype=cdt)
        emitted = C.c_int64()
        check(call(C.c_void_p(out.ctypes.data), capacity, C.byref(emitted)))
        if emitted.value == 0:
            return None
        return np.ascontiguousarray(out[:, :, :emitted.value])

    def _emit_dev(self, call, capacity, device):
        """the same into a device-resident window (smx_stft_kernel_step_dev / flush_dev): nothing crosses the host"""
        import torch
        real = torch.float64 if self._dtype == np.float64 else torch.float32
        cplx = torch.complex128 if self._dtype == np.float64 else torch.complex64
        out = torch.empty((self._channels, self._cfg.bins, capacity), device=device, dtype=real if getattr(self, "_real", False) else cplx)
        emitted = C.c_int64()
        with torch.cuda.device(device):
            stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            check(call(C.c_void_p(out.data_ptr()), capacity, C.byref(emitted), stream))
        return None if emitted.value == 0 else out[:, :, :emitted.value].contiguous()

    def step(self, chunk):
        from ._tensor import is_device
        if is_device(chunk):
            import torch
            t = chunk.to(torch.float64 if self._dtype == np.float64 else torch.float32)
            if t.dim() >= 1 and 0 in t.shape[:-1]:
                raise _lib.InvalidArgument(
                    "step: cannot analyse a chunk with a zero-size leading axis (channels must be at least 1)")
            lead = int(np.prod(t.shape[:-1], dtype=np.int64)) if t.dim() >= 1 else 1
            if lead != self._channels:
                check(lib.smx_stft_kernel_set_channels(self._h, lead))
                self._channels = lead
            t = t.contiguous().reshape(self._channels, -1)
            m = int(t.shape[-1])
            cfg = self._cfg
            capacity = (m + cfg.fft_size + left_width(cfg) + right_width(cfg)) // cfg.hop + 2
            self._device = t.device
            return self._emit_dev(lambda o, cap, e, st: lib.smx_stft_kernel_step_dev(
                self._h, C.c_void_p(t.data_ptr()), m, m, o, cap, e, st), capacity, t.device)
        a = np.asarray(chunk)
        if a.ndim >= 1 and 0 in a.shape[:-1]:
            raise _lib.InvalidArgument(
                "step: cannot analyse a chunk with a zero-size leading axis (channels must be at least 1)")
        lead = int(np.prod(a.shape[:-1], dtype=np.int64)) if a.ndim >= 1 else 1
        if lead != self._channels:
            # the reference's state takes its leading shape from the chunks (stft.ml:521-559): a fresh kernel follows the
            # first one, a stream that already holds samples refuses another count
            check(lib.smx_stft_kernel_set_channels(self._h, lead))
            self._channels = lead
        a = np.ascontiguousarray(a.astype(self._dtype, copy=False)).reshape(self._channels, -1)
        m = a.shape[-1]
        cfg = self._cfg
        capacity = (m + cfg.fft_size + left_width(cfg) + right_width(cfg)) // cfg.hop + 2
        return self._emit(lambda o, cap, e: lib.smx_stft_kernel_step(
            self._h, C.c_void_p(a.ctypes.data), m, o, cap, e), capacity)

    def flush(self):
        cfg = self._cfg
        capacity = (2 * cfg.fft_size + left_width(cfg) + right_width(cfg)) // cfg.hop + 2
        dev = getattr(self, "_device", None)
        if dev is not None:   # the stream was fed from device memory: the drain stays there
            return self._emit_dev(lambda o, cap, e, st: lib.smx_stft_kernel_flush_dev(self._h, o, cap, e, st), capacity, dev)
        return self._emit(lambda o, cap, e: lib.smx_stft_kernel_flush(self._h, o, cap, e), capacity)

    def reset(self):
        self._device = None
        check(lib.smx_stft_kernel_reset(self._h))


class Synthesis:
    """``Stft.Synthesis`` (stft.ml:1271-1298, stft.mli:519-591): incremental least-squares synthesis, the state (the last
    ``ceil(fft/hop) - 1`` spectra and the held samples) in device memory.  ``step`` takes frames ``[channels; bins; k]``
    (numpy complex64 / complex128, or a device-resident torch tensor: then nothing crosses the host) and returns the
    samples they settle ``[channels; m]`` or ``None``; ``flush`` drains the trimmed tail.  Any chunking of a stream
    totals ``Stft.invert`` of the whole stream, bit for bit.  Mutable, single-owner."""

    def __init__(self, handle, cfg, dtype, channels):
        self._h, self._cfg, self._dtype, self._channels = handle, cfg, np.dtype(dtype), channels
        bound = C.c_int64()
        check(lib.smx_stft_synthesis_sample_bound(self._h, C.byref(bound)))
        self.sample_bound = bound.value

    @staticmethod
    def prepare(c: Config, dtype, channels: int, max_block: int) -> "Synthesis":
        handle = C.c_void_p()
        dt = np.dtype(dtype)
        wide = dt in (np.dtype(np.float64), np.dtype(np.complex128))
        check(lib.smx_stft_synthesis_prepare(c._h, 8 if wide else 4, int(channels), int(max_block), C.byref(handle)))
        return Synthesis(handle, c, np.float64 if wide else np.float32, int(channels))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            try:
                lib.smx_stft_synthesis_destroy(h)
            except Exception:
                pass

    @staticmethod
    def latency(c: Config) -> int:
        """``Config.synthesis_latency`` (stft.ml:152-153), in output samples."""
        return int(lib.smx_stft_synthesis_latency(c._h))

    def _capacity(self, k):
        cfg = self._cfg
        return max(k * cfg.hop, cfg.fft_size) + cfg.hop

    def step(self, z):
        from ._tensor import is_device, is_torch
        shape = tuple(z.shape)
        if len(shape) < 2:   # check_frames, stft.ml:753-759
            raise _lib.InvalidArgument("step: cannot invert a rank-%d tensor (the bin and frame axes must exist)" % len(shape))
        if 0 in shape[:-2]:  # stft.ml:1189-1193
            raise _lib.InvalidArgument("step: cannot synthesise frames with a zero-size leading axis (channels must be at least 1)")
        bins, k = int(shape[-2]), int(shape[-1])
        lead = 1
        for d in shape[:-2]:
            lead *= int(d)
        # the library reads channels x bins rows of k frames: a chunk whose leading axes hold another number of channels than the
        # kernel was prepared for would be read past its end (or silently truncated)
        if lead != self._channels:
            raise _lib.InvalidArgument("step: the chunk holds %d channel(s) (leading axes %s), the kernel was prepared for %d"
                                       % (lead, list(shape[:-2]), self._channels))
        cdt = np.complex128 if self._dtype == np.float64 else np.complex64
        cap = self._capacity(k)
        emitted = C.c_int64()
        if is_device(z):
            import torch
            tz = z.to(torch.complex128 if self._dtype == np.float64 else torch.complex64).reshape(self._channels, bins, k).contiguous()
            zr = torch.view_as_real(tz)
            self._device = z.device
            out = torch.empty((self._channels, cap), device=z.device, dtype=torch.float64 if self._dtype == np.float64 else torch.float32)
            with torch.cuda.device(z.device):
                stream = C.c_void_p(torch.cuda.current_stream(z.device).cuda_stream)
                check(lib.smx_stft_synthesis_step_dev(self._h, C.c_void_p(zr.data_ptr()), bins, k, C.c_void_p(out.data_ptr()), cap,
                                                      C.byref(emitted), stream))
            return None if emitted.value == 0 else out[:, :emitted.value].contiguous()
        a = z.detach().cpu().numpy() if is_torch(z) else np.asarray(z)
        a = np.ascontiguousarray(a.astype(cdt, copy=False)).reshape(self._channels, bins, k)
        out = np.zeros((self._channels, cap), dtype=self._dtype)
        check(lib.smx_stft_synthesis_step(self._h, C.c_void_p(a.ctypes.data), bins, k, C.c_void_p(out.ctypes.data), cap, C.byref(emitted)))
        return None if emitted.value == 0 else np.ascontiguousarray(out[:, :emitted.value])

    def flush(self):
        cap = self._capacity(0)
        emitted = C.c_int64()
        dev = getattr(self, "_device", None)
        if dev is not None:   # the stream was fed from the device: the tail stays there (as Kernel.flush does)
            import torch
            out = torch.empty((self._channels, cap), device=dev, dtype=torch.float64 if self._dtype == np.float64 else torch.float32)
            with torch.cuda.device(dev):
                stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                check(lib.smx_stft_synthesis_flush_dev(self._h, C.c_void_p(out.data_ptr()), cap, C.byref(emitted), stream))
            return None if emitted.value == 0 else out[:, :emitted.value].contiguous()
        out = np.zeros((self._channels, cap), dtype=self._dtype)
        check(lib.smx_stft_synthesis_flush(self._h, C.c_void_p(out.ctypes.data), cap, C.byref(emitted)))
        return None if emitted.value == 0 else np.ascontiguousarray(out[:, :emitted.value])

    def reset(self):
        check(lib.smx_stft_synthesis_reset(self._h))
        self._device = None


# ---- Pipeline-stage faces (stft.ml:1301-1409) ---------------------------------------------------------------
# The reference wraps the streaming state machine as Pipeline stages; the algebra of Pipeline itself is out of
# scope here, but the numbers a stage declares and the step / flush / reset / concat bodies are these.

def stage_latency(c: Config) -> int:
    """``stage_latency`` (stft.ml:1307-1308): the geometric lookahead, or the reflected left border's reach."""
    return int(lib.smx_stft_stage_latency(c._h))


def stage_rate(c: Config):
    """``stage_rate`` (stft.ml:1340): one frame per ``hop`` samples, as (num, den)."""
    return (1, c.hop)


def frame_bound(c: Config, max_items: int) -> int:
    """``frame_bound c b`` (stft.ml:1316-1317): the most frames one step or drained tail can emit for chunks of at
    most ``b`` samples."""
    return int(lib.smx_stft_frame_bound(c._h, int(max_items)))


class _Stage:
    """A prepared stage: ``step`` returns a chunk or None, ``flush`` a list of chunks split at the threaded bound
    (stft.ml:1322-1336), ``reset`` rewinds, ``concat`` joins chunks along the frame axis."""

    def __init__(self, cfg, power, max_items):
        self._cfg, self._power = cfg, power
        self.latency = stage_latency(cfg)
        self.rate = stage_rate(cfg)
        self.bound = None if max_items is None else frame_bound(cfg, max_items)
        self._max_items = max_items
        self._k = None          # created at the first chunk: it fixes the element dtype and the channel count
        self._dtype = None

    def _kernel(self, a):
        if self._k is None:
            self._dtype = np.dtype(np.float64 if a.dtype == np.float64 else np.float32)
            handle = C.c_void_p()
            channels = int(np.prod(a.shape[:-1])) if a.ndim > 1 else 1
            block = int(self._max_items) if self._max_items is not None else max(int(a.shape[-1]), 1)
            if self._power is None:
                check(lib.smx_stft_kernel_prepare(self._cfg._h, self._dtype.itemsize, channels, block, C.byref(handle)))
            else:
                check(lib.smx_stft_kernel_prepare_power(self._cfg._h, self._dtype.itemsize, channels, block,
                                                        float(self._power), C.byref(handle)))
            self._k = Kernel(handle, self._cfg, self._dtype, channels)
            self._k._real = self._power is not None
            self._lead = a.shape[:-1]
        return self._k

    def _shape(self, out):
        return None if out is None else out.reshape(self._lead + out.shape[-2:])

    def step(self, chunk):
        a = np.asarray(chunk)
        return self._shape(self._kernel(a).step(a))

    def flush(self):
        if self._k is None:
            return []
        out = self._shape(self._k.flush())
        if out is None:
            return []
        if self.bound is None or out.shape[-1] <= self.bound:
            return [out]
        return [out[..., i:i + self.bound] for i in range(0, out.shape[-1], self.bound)]

    def reset(self):
        if self._k is not None:
            self._k.reset()

    def concat(self, parts):
        if not parts:
            if self._power is not None and self._dtype is None:
                raise _lib.InvalidArgument("power_stage: cannot concatenate zero chunks before any chunk fixed the "
                                           "element dtype")
            dt = (np.complex128 if self._dtype == np.float64 else np.complex64) if self._power is None else self._dtype
            return np.zeros((self._cfg.bins, 0), dtype=dt)
        return np.concatenate(parts, axis=-1)


class _StageFactory:
    def __init__(self, cfg, power):
        self._cfg, self._power = cfg, power
        self.latency = stage_latency(cfg)
        self.rate = stage_rate(cfg)

    def prepare(self, max_items=None) -> _Stage:
        """``~prepare`` of the stage: a fresh state; ``max_items`` is the input format's chunk bound (or None)."""
        return _Stage(self._cfg, self._power, max_items)

    def out_max_items(self, max_items):
        """``stage_out_format``'s threaded bound (stft.ml:1318-1320, 1342-1348)."""
        return None if max_items is None else frame_bound(self._cfg, max_items)


def stage(c: Config) -> _StageFactory:
    """``Stft.stage cdtype c`` (stft.ml:1350-1362): the complex spectrum, streaming."""
    return _StageFactory(c, None)


def power_stage(c: Config, power: float = 2.0) -> _StageFactory:
    """``Stft.power_stage ?power c`` (stft.ml:1364-1409): |spectrum|^power in the chunks' dtype, streaming; the
    magnitudes are taken on the device by the same kernels as ``power_spectrum``."""
    return _StageFactory(c, float(power))

"""Flat feature API: ``Soundml.mel_spectrogram`` (soundml.ml:12-24)."""
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

"""The fused mel spectrogram at fft 2048 / 1024 / 512 has two forms of the filterbank product (stft_fast_mel32.hpp): the banded
one on v_mfma_f32_4x4x1_16B_f32 (default: every 4-mel group over its own band, long bands cut in K-parts that are added in a
fixed order; operands resident in registers when a wave's share fits -- 64 steps at fft 2048, 32 at fft 1024 / 512 -- else
streamed from L2 (fft 2048) or the dense form) and the dense one on v_mfma_f32_16x16x4_f32 (`SMX_MEL_DENSE=1`).  They sum a mel's products in different orders, so they agree to rounding, not bit
for bit: both are held to the oracle (mel.ml:202-231 after stft.ml:687-691) at the regression gate of 2e-6 of the peak, and the
banded form to the batch-slice law (mel_props.ml:136-155) bit for bit."""
import os

import numpy as np
import pytest

from oracle import soundml_oracle as O

import soundml_amd as S
from soundml_amd import Mel, Stft

pytestmark = pytest.mark.gpu

BANKS = [
    # fft, n_mels, sample rate, scale, norm, f_min, f_max
    (2048, 128, 48000, "slaney", "slaney", 0.0, None),      # C3: resident operands, modes 1 / 2 / 4
    (2048, 40, 22050, "slaney", "slaney", 0.0, None),       # few wide mels: K-parts everywhere
    (2048, 80, 16000, "htk", "none", 20.0, 7600.0),
    (2048, 6, 16000, "slaney", "slaney", 0.0, None),        # fewer rows than a lane group holds
    (2048, 13, 44100, "htk", "slaney", 300.0, 12000.0),     # a last group of one row
    (2048, 250, 48000, "slaney", "none", 0.0, None),        # many narrow mels: more items than operand registers (streamed)
    (1024, 80, 22050, "slaney", "slaney", 0.0, None),       # the 16-lane kernel: two groups of 16 frame columns per operand
    (1024, 128, 44100, "htk", "none", 0.0, None),
    (1024, 5, 16000, "slaney", "slaney", 0.0, None),
    (512, 80, 16000, "slaney", "slaney", 0.0, None),        # the 8-lane kernel: four groups of columns
    (512, 40, 16000, "htk", "slaney", 50.0, 7000.0),
    (512, 20, 8000, "slaney", "none", 0.0, None),
    (2048, 1, 16000, "slaney", "slaney", 0.0, None),        # one, two, three rows: a single lane group, partly idle
    (2048, 2, 16000, "htk", "none", 1000.0, 3000.0),
    (1024, 3, 22050, "slaney", "slaney", 0.0, None),
]


@pytest.fixture(autouse=True)
def _env():
    os.environ.pop("SMX_MEL_DENSE", None)
    yield
    os.environ.pop("SMX_MEL_DENSE", None)


@pytest.mark.parametrize("fft,n_mels,sr,scale,norm,f_min,f_max", BANKS)
def test_banded_and_dense_products_meet_the_oracle(fft, n_mels, sr, scale, norm, f_min, f_max):
    rng = np.random.default_rng(21)
    x = rng.uniform(-1, 1, size=(3, 30000)).astype(np.float32)
    sc, so = Stft.Config.create(fft_size=fft, hop=fft // 4), O.stft_config(fft, hop=fft // 4)
    try:
        mc = Mel.Config.create(n_mels=n_mels, sample_rate=sr, fft_size=fft, scale=scale, norm=norm, f_min=f_min, f_max=f_max)
    except S.InvalidArgument:
        pytest.skip("the reference rejects this filterbank (an empty filter)")
    om = O.mel_config(n_mels, sr, fft, f_min=f_min, f_max=f_max, scale=scale, norm=norm)
    for power in (2.0, 1.0, 3.0):
        want = O.mel_spectrogram(so, om, x, power)
        peak = float(np.max(np.abs(want)))
        banded = S.mel_spectrogram(sc, mc, x, power)
        os.environ["SMX_MEL_DENSE"] = "1"
        try:
            dense = S.mel_spectrogram(sc, mc, x, power)
        finally:
            os.environ.pop("SMX_MEL_DENSE", None)
        for name, got in (("banded", banded), ("dense", dense)):
            assert got.shape == want.shape
            err = float(np.max(np.abs(got.astype(np.float64) - want)))
            assert err <= 2e-6 * peak, "%s product, power %g: max error %.3g of the peak" % (name, power, err / peak)


@pytest.mark.parametrize("fft", [2048, 1024, 512])
def test_a_clip_of_the_banded_product_does_not_depend_on_its_batch(fft):
    import torch
    torch.manual_seed(4)
    x = (torch.rand(37, 20000, device="cuda") * 2 - 1).float()
    sc = Stft.Config.create(fft_size=fft, hop=fft // 4)
    mc = Mel.Config.create(n_mels=128 if fft == 2048 else 80, sample_rate=48000 if fft == 2048 else 22050, fft_size=fft)
    whole = S.mel_spectrogram(sc, mc, x, 2.0)
    for i in (0, 17, 36):
        assert torch.equal(S.mel_spectrogram(sc, mc, x[i:i + 1].contiguous(), 2.0)[0], whole[i])
    assert torch.equal(S.mel_spectrogram(sc, mc, x[5:20].contiguous(), 2.0), whole[5:20])

"""The fused mel spectrogram at fft 2048 has two forms of the filterbank product (stft_fast_mel32.hpp): the banded one on
v_mfma_f32_4x4x1_16B_f32 (default: every 4-mel group over its own band, long bands cut in K-parts that are added in a fixed
order; operands resident in registers when a wave's share fits 64 steps, streamed from L2 otherwise) and the dense one on
v_mfma_f32_16x16x4_f32 (`SMX_MEL_DENSE=1`).  They sum a mel's products in different orders, so they agree to rounding, not bit
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
    # n_mels, sample rate, scale, norm, f_min, f_max
    (128, 48000, "slaney", "slaney", 0.0, None),      # C3: resident operands, modes 1 / 2 / 4
    (40, 22050, "slaney", "slaney", 0.0, None),       # few wide mels: K-parts everywhere
    (80, 16000, "htk", "none", 20.0, 7600.0),
    (6, 16000, "slaney", "slaney", 0.0, None),        # fewer rows than a lane group holds
    (13, 44100, "htk", "slaney", 300.0, 12000.0),     # a last group of one row
    (250, 48000, "slaney", "none", 0.0, None),        # many narrow mels: more items than operand registers (streamed)
]


@pytest.fixture(autouse=True)
def _env():
    os.environ.pop("SMX_MEL_DENSE", None)
    yield
    os.environ.pop("SMX_MEL_DENSE", None)


@pytest.mark.parametrize("n_mels,sr,scale,norm,f_min,f_max", BANKS)
def test_banded_and_dense_products_meet_the_oracle(n_mels, sr, scale, norm, f_min, f_max):
    rng = np.random.default_rng(21)
    x = rng.uniform(-1, 1, size=(3, 30000)).astype(np.float32)
    sc, so = Stft.Config.create(fft_size=2048, hop=512), O.stft_config(2048, hop=512)
    try:
        mc = Mel.Config.create(n_mels=n_mels, sample_rate=sr, fft_size=2048, scale=scale, norm=norm, f_min=f_min, f_max=f_max)
    except S.InvalidArgument:
        pytest.skip("the reference rejects this filterbank (an empty filter)")
    om = O.mel_config(n_mels, sr, 2048, f_min=f_min, f_max=f_max, scale=scale, norm=norm)
    for power in (2.0, 1.0):
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


def test_a_clip_of_the_banded_product_does_not_depend_on_its_batch():
    import torch
    torch.manual_seed(4)
    x = (torch.rand(37, 20000, device="cuda") * 2 - 1).float()
    sc = Stft.Config.create(fft_size=2048, hop=512)
    mc = Mel.Config.create(n_mels=128, sample_rate=48000, fft_size=2048)
    whole = S.mel_spectrogram(sc, mc, x, 2.0)
    for i in (0, 17, 36):
        assert torch.equal(S.mel_spectrogram(sc, mc, x[i:i + 1].contiguous(), 2.0)[0], whole[i])
    assert torch.equal(S.mel_spectrogram(sc, mc, x[5:20].contiguous(), 2.0), whole[5:20])

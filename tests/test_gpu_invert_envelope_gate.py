"""`Stft.invert` divides the overlap-added frames by the overlap-added squared window (stft.ml:902-939): where that envelope is far
below its interior level -- the first / last samples of a left- or right-aligned clip under a Hann window -- the float32 interior's
rounding is multiplied by level / envelope (DESIGN "Precision contract").  The gate: at EVERY output position, tiny envelopes
included (down to 4e-12 of the level here, where the raw error is 2.5 % of the peak),

    |got - want| <= 2e-6 x peak x level / envelope        float32 interior (measured: 1.8e-7 ... 7.8e-7)
    |got - want| <= 2e-6 x peak                           float64 interior (measured: equal to the oracle's float32 result)

so the float32 interior loses nothing but what the division multiplies, and the reference's own numerics are exact there."""
import numpy as np
import pytest

from oracle import soundml_oracle as O

import soundml_amd as S
from soundml_amd import Stft

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fft,hop,alignment", [(2048, 512, "left"), (2048, 512, "right"), (2048, 512, "centered"),
                                               (1024, 256, "left"), (400, 100, "left"), (512, 256, "right")])
def test_invert_error_is_the_interior_rounding_times_level_over_envelope(fft, hop, alignment):
    rng = np.random.default_rng(3)
    n = 12 * fft + 37
    x = rng.uniform(-1, 1, size=(2, n)).astype(np.float32)
    kw = dict(hop=hop, alignment=alignment)
    c, o = Stft.Config.create(fft_size=fft, **kw), O.stft_config(fft, **kw)
    z = O.transform(o, x).astype(np.complex64)
    want = O.invert(o, z, None)
    env = O.envelope(o, z.shape[-1])[O.left_width(o):][:want.shape[-1]]
    level = float(np.sum(o.analysis_window ** 2)) / hop
    peak = float(np.abs(want).max())
    amp = level / np.maximum(env, 1e-300)
    try:
        for interior in ("float32", "float64"):
            S.set_interior(interior)
            got = Stft.invert(c, z, None)
            assert got.shape == want.shape and got.dtype == np.float32
            err = np.abs(got.astype(np.float64) - want).max(axis=0)
            bound = 2e-6 * peak * (amp if interior == "float32" else 1.0)
            worst = int(np.argmax(err / bound))
            assert (err <= bound).all(), "%s interior: position %d: error %.3g, bound %.3g (envelope %.3g of its level)" % (
                interior, worst, err[worst], np.broadcast_to(bound, err.shape)[worst], env[worst] / level)
    finally:
        S.set_interior("float32")

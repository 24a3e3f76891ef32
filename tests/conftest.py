import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(suite, name):
    """One committed librosa-0.11 vector file of the reference's test suite
    (soundml/test/<suite>/vectors/<name>.json; schema: test/support/tutils.ml:22-78)."""
    with open(os.path.join(GOLDEN, suite, name + ".json")) as fh:
        return json.load(fh)


def golden_cases(suite, name):
    return [pytest.param(c, id=c["name"]) for c in load_golden(suite, name)["cases"]]


# reference tolerances: test/support/tutils.ml:80-86, stft_goldens.ml:13-17
F64_RTOL, F64_ATOL = 1e-9, 1e-12
F64_STRICT_RTOL, F64_STRICT_ATOL = 1e-12, 1e-15
F32_RTOL, F32_ATOL = 1e-6, 1e-7


def check_close(actual, expected, shape=None, rtol=F64_STRICT_RTOL, atol=F64_STRICT_ATOL, msg=""):
    """tutils.ml:91-117: |a - e| <= atol + rtol*|e| elementwise, NaN == NaN, shape checked."""
    actual = np.asarray(actual)
    if shape is not None:
        assert list(actual.shape) == list(shape), "%s: shape %s, expected %s" % (msg, actual.shape, shape)
    a = actual.astype(np.float64).reshape(-1)
    e = np.asarray(expected, dtype=np.float64).reshape(-1)
    assert a.size == e.size, "%s: %d elements, expected %d" % (msg, a.size, e.size)
    tol = atol + rtol * np.abs(e)
    bad = ~((np.isnan(e) & np.isnan(a)) | (np.abs(a - e) <= tol))
    if bad.any():
        i = int(np.argmax(bad))
        raise AssertionError(
            "%s: index %d: got %.17g, expected %.17g (delta %.3g, tolerance %.3g); %d/%d bad"
            % (msg, i, a[i], e[i], abs(a[i] - e[i]), tol[i], int(bad.sum()), a.size))


def istft_golden_spectrum(fft_size, frames):
    """The synthetic spectrum of the reference's synthesis goldens (istft_goldens.ml:31-49): two 31-bit LCG
    streams, the imaginary parts of the DC and Nyquist bins zero."""
    import numpy as np
    from oracle import soundml_oracle as O
    bins = fft_size // 2 + 1
    re = O.lcg_signal(bins * frames, 20250803)
    im = O.lcg_signal(bins * frames, 20250804)
    z = (re + 1j * im).reshape(bins, frames)
    z[0] = z[0].real
    if fft_size % 2 == 0:
        z[-1] = z[-1].real
    return z


def istft_golden_config(make, params):
    """Stft.Config of a synthesis golden case (istft_goldens.ml:63-69): constant-zero padding."""
    return make(params["fft_size"], win_length=params["win_length"], hop=params["hop"],
                alignment=params["alignment"], pad="constant", pad_value=0.0)

"""Randomised parity sweep (diagnostic; the committed tests are deterministic): random geometries through transform,
power_spectrum, mel_spectrogram, invert and the streaming faces, each against the oracle.  Prints every failure with the
drawn parameters; exit code = number of failures.   python tools/fuzz_parity.py [cases] [seed]"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import soundml_amd as S
from soundml_amd import Stft, Mel
from oracle import soundml_oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
FFTS = [16, 31, 64, 100, 256, 400, 441, 512, 1000, 1024, 1200, 2048, 4096]


def close(a, e, rtol, atol_rel, what):
    a, e = np.asarray(a), np.asarray(e)
    assert a.shape == e.shape, (what, a.shape, e.shape)
    if e.size == 0:
        return
    peak = float(np.max(np.abs(e)))
    bad = np.abs(a.astype(np.complex128) - e.astype(np.complex128)) > atol_rel * peak + rtol * np.abs(e)
    assert not bad.any(), "%s: %d/%d outside tolerance (max err %.3g, peak %.3g)" % (
        what, int(bad.sum()), e.size, float(np.max(np.abs(a.astype(np.complex128) - e))), peak)


fails = 0
for case in range(cases):
    fft = int(rng.choice(FFTS))
    hop = int(rng.integers(1, 2 * fft))
    if rng.random() < 0.5:
        hop = max(1, fft // int(rng.choice([2, 4, 8])))
    win = None if rng.random() < 0.7 else int(rng.integers(1, fft + 1))
    alignment = str(rng.choice(["centered", "left", "right"]))
    pad = [("reflect"), ("edge"), ("constant", float(rng.normal()))][int(rng.integers(0, 3))]
    lead = tuple(int(v) for v in rng.integers(1, 4, size=int(rng.integers(0, 3))))
    n = int(rng.integers(1, 6 * fft + 50))
    f64 = rng.random() < 0.3
    interior = "float64" if (not f64 and rng.random() < 0.3) else "float32"
    power = float(rng.choice([1.0, 2.0, 2.0, 0.7]))
    params = dict(fft=fft, hop=hop, win=win, alignment=alignment, pad=pad, lead=lead, n=n, f64=f64, interior=interior, power=power)
    try:
        x = rng.uniform(-1, 1, size=lead + (n,)).astype(np.float64 if f64 else np.float32)
        kw = dict(hop=hop, win_length=win, alignment=alignment)
        c = Stft.Config.create(fft_size=fft, pad=pad, **kw)
        o = (O.stft_config(fft, pad=pad[0], pad_value=pad[1], **kw) if isinstance(pad, tuple) else O.stft_config(fft, pad=pad, **kw))
        strict = f64 or interior == "float64"
        rt, at = (1e-9, 1e-12) if f64 else ((2e-6, 2e-7) if strict else (1e-5, 1e-5))
        S.set_interior(interior)
        z, wz = Stft.transform(c, x), O.transform(o, x)
        close(z, wz, rt, at, "transform")
        # |X|^p with p < 1 turns an absolute error d near a zero of the spectrum into d^p: looser floor there
        close(Stft.power_spectrum(c, x, power), O.power_spectrum(o, x, power), 4 * rt, at if power >= 1.0 else max(at, 1e-5) ** power, "power")
        total = Stft.frames(c, n)
        if total > 0:
            a, b = sorted(int(v) for v in rng.integers(0, total + 1, size=2))
            assert np.array_equal(Stft.transform_range(c, x, a, b), z[..., a:b]), "range [%d, %d) is not the slice" % (a, b)
        if fft >= 64 and fft % 2 == 0 and not f64 and interior == "float32":
            try:
                mc = Mel.Config.create(n_mels=int(rng.integers(4, 40)), sample_rate=16000, fft_size=fft)
            except S.InvalidArgument:
                mc = None
            if mc is not None:
                om = O.mel_config(mc.n_mels, 16000, fft)
                close(S.mel_spectrogram(c, mc, x, power), O.mel_spectrogram(o, om, x, power), 1e-5, 1e-5 if power >= 1.0 else 1e-5 ** power, "mel")
        if Stft.nola(c) and total > 0:
            length = None if rng.random() < 0.5 else int(rng.integers(1, n + fft))
            zz = wz.astype(np.complex128 if f64 else np.complex64)
            got_x, want_x = Stft.invert(c, zz, length), O.invert(o, zz, length)
            # the division by the envelope amplifies rounding where the overlap-added squared window is tiny (DESIGN
            # "Precision contract"): positions under 1e-3 of the envelope's interior level are not compared
            env = O.envelope(o, total)[O.left_width(o):][:want_x.shape[-1]]
            level = float(np.sum(o.analysis_window ** 2)) / float(hop)
            ok = np.zeros(want_x.shape[-1], dtype=bool)
            ok[:env.shape[0]] = env > 1e-3 * level
            if ok.any():
                if f64:
                    close(got_x[..., ok], want_x[..., ok], 1e-9, 1e-11, "invert")
                elif strict:
                    close(got_x[..., ok], want_x[..., ok], 2e-6, 2e-6, "invert")
                else:
                    close(got_x[..., ok], want_x[..., ok], 1e-5, 3e-5, "invert (well-conditioned positions)")
        # streaming: random chunking reproduces the offline transform bit for bit
        st = Stft.stage(c).prepare(max_items=max(1, n // 3))
        parts, pos = [], 0
        while pos < n:
            m = int(rng.integers(1, max(2, n // 3) + 1))
            out = st.step(x[..., pos:pos + m])
            pos += m
            if out is not None:
                parts.append(out)
        got = st.concat(parts + st.flush())
        if total == 0:
            assert got.shape[-1] == 0, "streaming emitted frames of a frameless signal"
        else:
            assert got.shape == z.shape and np.array_equal(got, z), "streaming partition differs from the offline transform"
    except Exception as e:   # noqa: BLE001
        fails += 1
        print("FAIL", params, "->", "".join(traceback.format_exception_only(type(e), e)).strip()[:400])
    finally:
        S.set_interior("float32")
print("%d cases, %d failures" % (cases, fails))
sys.exit(min(fails, 100))

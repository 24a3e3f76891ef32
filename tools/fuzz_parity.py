"""Randomised parity sweep (diagnostic; the committed tests are deterministic): random geometries through transform,
power_spectrum, mel_spectrogram, invert and the streaming faces, each against the oracle.  Prints every failure with the
drawn parameters; exit code = number of failures.   python tools/fuzz_parity.py [cases] [seed] [stft|features]
"features" draws the callers either side instead: Mel.apply, mfcc, to-dB, the spectral features, Chroma, the FIR
filter and Griffin-Lim."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import soundml_amd as S
from soundml_amd import Stft, Mel
from oracle import soundml_oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
FFTS = [16, 31, 64, 100, 256, 400, 441, 512, 1000, 1024, 1200, 2048, 4096]
if os.environ.get("FUZZ_FFTS"):   # e.g. FUZZ_FFTS=2048,1024,512: the sizes of the register pipelines only
    FFTS = [int(v) for v in os.environ["FUZZ_FFTS"].split(",")]


def close(a, e, rtol, atol_rel, what, floor=0.0):
    a, e = np.asarray(a), np.asarray(e)
    assert a.shape == e.shape, (what, a.shape, e.shape)
    if e.size == 0:
        return
    peak = float(np.max(np.abs(e)))
    bad = np.abs(a.astype(np.complex128) - e.astype(np.complex128)) > floor + atol_rel * peak + rtol * np.abs(e)
    assert not bad.any(), "%s: %d/%d outside tolerance (max err %.3g, peak %.3g)" % (
        what, int(bad.sum()), e.size, float(np.max(np.abs(a.astype(np.complex128) - e))), peak)


GL_PAIRS = []   # (device's rel l2 to the float64 oracle, the float32-storage yardstick's) of every Griffin-Lim draw


def feature_case(rng):
    """One draw of the callers around the STFT; returns the parameters it used (for the failure line)."""
    which = str(rng.choice(["mel_apply", "mfcc", "db", "spectral", "chroma", "fir", "griffin_lim"]))
    fft = int(rng.choice([64, 256, 400, 512, 1024, 2048]))
    lead = tuple(int(v) for v in rng.integers(1, 4, size=int(rng.integers(0, 3))))
    frames = int(rng.integers(1, 200))
    bins = fft // 2 + 1
    params = dict(which=which, fft=fft, lead=lead, frames=frames)
    try:
        if which == "mel_apply":
            n_mels = int(rng.integers(2, 64))
            norm = str(rng.choice(["slaney", "none"]))
            scale = str(rng.choice(["slaney", "htk"]))
            params.update(n_mels=n_mels, norm=norm, scale=scale)
            try:
                mc = Mel.Config.create(n_mels=n_mels, sample_rate=22050, fft_size=fft, norm=norm, scale=scale)
            except S.InvalidArgument:
                return None
            om = O.mel_config(n_mels, 22050, fft, norm=norm, scale=scale)
            s_ = rng.uniform(0, 4, size=lead + (bins, frames)).astype(np.float32)
            close(Mel.apply(mc, s_), O.mel_apply(om, s_), 1e-5, 1e-6, "Mel.apply")
        elif which == "mfcc":
            n_mels = int(rng.integers(8, 64))
            n_mfcc = int(rng.integers(1, n_mels + 1))
            lifter = None if rng.random() < 0.5 else float(rng.integers(0, 40))
            hop = max(1, fft // int(rng.choice([2, 4])))
            n = int(rng.integers(fft, 12 * fft))
            params.update(n_mels=n_mels, n_mfcc=n_mfcc, lifter=lifter, hop=hop, n=n)
            try:
                mc = Mel.Config.create(n_mels=n_mels, sample_rate=22050, fft_size=fft)
            except S.InvalidArgument:
                return None
            c, o = Stft.Config.create(fft_size=fft, hop=hop), O.stft_config(fft, hop=hop)
            x = rng.uniform(-1, 1, size=lead + (n,)).astype(np.float32)
            want = O.mfcc(o, O.mel_config(n_mels, 22050, fft), x, n_mfcc, lifter)
            # dB of a near-zero band amplifies the float32 interior: absolute floor in dB units
            got = S.mfcc(c, mc, x, n_mfcc, lifter)
            assert got.shape == want.shape and np.max(np.abs(got - want)) <= 2e-3 * max(1.0, float(np.max(np.abs(want)))), \
                "mfcc: max err %.3g (peak %.3g)" % (float(np.max(np.abs(got - want))), float(np.max(np.abs(want))))
        elif which == "db":
            top = None if rng.random() < 0.4 else float(rng.uniform(0, 100))
            amp = rng.random() < 0.5
            s_ = (rng.uniform(0, 1, size=lead + (bins, frames)) ** 8).astype(np.float32)
            params.update(top_db=top, amplitude=amp)
            fn, on = (S.amplitude_to_db, O.amplitude_to_db) if amp else (S.power_to_db, O.power_to_db)
            got, want = fn(s_, top_db=top), on(s_, top_db=top)
            assert got.shape == want.shape and np.max(np.abs(got - want)) <= 2e-5 * 100, "to_db: max err %.3g" % float(np.max(np.abs(got - want)))
        elif which == "spectral":
            sr = int(rng.choice([8000, 16000, 22050, 44100]))
            s_ = rng.uniform(0, 2, size=lead + (bins, frames)).astype(np.float32 if rng.random() < 0.7 else np.float64)
            if rng.random() < 0.3:
                s_[..., :, int(rng.integers(0, frames))] = 0.0      # a silent frame
            pb = float(rng.choice([1.0, 2.0, 3.0, 1.5]))
            roll = float(rng.uniform(0.01, 0.99))
            params.update(sr=sr, p=pb, roll=roll, dtype=str(s_.dtype))
            tol = 1e-5 if s_.dtype == np.float32 else 1e-9
            close(S.spectral_centroid(s_, sr), O.spectral_centroid(s_, sr), tol, tol, "centroid")
            close(S.spectral_bandwidth(s_, sr, pb), O.spectral_bandwidth(s_, sr, pb), 4 * tol, 4 * tol, "bandwidth")
            assert np.array_equal(S.spectral_rolloff(s_, sr, roll), O.spectral_rolloff(s_, sr, roll)), "rolloff bin differs"
            close(S.spectral_flatness(s_), O.spectral_flatness(s_), 4 * tol, 4 * tol, "flatness")
        elif which == "chroma":
            n_chroma = int(rng.choice([12, 12, 24, 6]))
            norm = [None, "inf", 1.0, 2.0, 3.0][int(rng.integers(0, 5))]
            tuning = float(rng.uniform(-0.5, 0.5))
            octw = None if rng.random() < 0.3 else 2.0
            params.update(n_chroma=n_chroma, norm=norm, tuning=tuning, octwidth=octw)
            cc = S.Chroma.Config.create(22050, fft, n_chroma=n_chroma, tuning=tuning, octwidth=octw)
            oc = O.chroma_config(22050, fft, n_chroma=n_chroma, tuning=tuning, octwidth=octw)
            s_ = rng.uniform(0, 2, size=lead + (bins, frames)).astype(np.float32)
            close(S.Chroma.apply(cc, s_, norm), O.chroma_apply(oc, s_, norm), 1e-5, 1e-5, "Chroma.apply")
        elif which == "fir":
            taps = int(rng.choice([1, 2, 3, 17, 64, 255, 1024, 4097, 8192]))
            n = int(rng.integers(1, 60000))
            params.update(taps=taps, n=n)
            h = rng.standard_normal(taps) / max(1.0, np.sqrt(taps))
            x = rng.uniform(-1, 1, size=lead + (n,)).astype(np.float32)
            close(S.Fir.apply(S.Fir.Plan.create(h), x), O.fir_filter(h, x), 1e-5, 1e-5, "fir")
        else:
            hop = fft // 4
            n_iter = int(rng.integers(1, 6))
            mom = float(rng.choice([0.0, 0.5, 0.99]))
            params.update(hop=hop, n_iter=n_iter, momentum=mom)
            c, o = Stft.Config.create(fft_size=fft, hop=hop), O.stft_config(fft, hop=hop)
            x = rng.uniform(-1, 1, size=lead + (hop * frames,)).astype(np.float32)
            mag = np.abs(O.transform(o, x)).astype(np.float32)
            init = rng.uniform(-np.pi, np.pi, size=mag.shape).astype(np.float32)
            got = Stft.griffin_lim(c, mag, n_iter=n_iter, momentum=mom, init=init)
            want = O.griffin_lim(o, mag, n_iter=n_iter, momentum=mom, init=init)
            if os.environ.get("FUZZ_DUMP_GL") and not (np.linalg.norm(got - want) < 1e-3 * np.linalg.norm(want)):   # keep the draw for a replay
                np.savez(os.path.join(os.environ["FUZZ_DUMP_GL"], "gl_case_%d_%d_%g.npz" % (fft, n_iter, mom)), mag=mag, init=init, fft=fft, hop=hop, n_iter=n_iter, mom=mom)
            # The unit-modulus step is ill-conditioned where a bin is nearly silent, and with momentum the loop is chaotic: one draw's
            # distance to the float64 oracle says little (round 5 gated it at 1e-3 and fell back to a convergence check on the draws
            # that turned).  Round 6: every draw is paired with the YARDSTICK's distance on the same draw (the float64 oracle with its
            # stored intermediates rounded to float32, O.griffin_lim_float32_storage) and the sweep is judged at its end by the gate
            # of tests/test_gpu_parity.py::test_griffin_lim_defaults_statistical_gate: median and maximum of the device's distances
            # within 16 x the yardstick's.  Per draw only what holds for every trajectory: finite, and inside the oracle's basin.
            rel_gl = float(np.linalg.norm(got - want) / np.linalg.norm(want))
            rel_yard = float(np.linalg.norm(O.griffin_lim_float32_storage(o, mag, n_iter, mom, init) - want) / np.linalg.norm(want))
            GL_PAIRS.append((rel_gl, rel_yard))
            assert got.shape == want.shape and np.isfinite(got).all() and rel_gl < 5e-2, "griffin_lim: relative l2 error %.3g (the yardstick's %.3g)" % (rel_gl, rel_yard)
            S.set_interior("float64")
            try:
                close(Stft.griffin_lim(c, mag, n_iter=n_iter, momentum=mom, init=init), want, 1e-5, 1e-5, "griffin_lim (float64 interior)")
            finally:
                S.set_interior("float32")
    except Exception as e:   # noqa: BLE001
        print("FAIL", params, "->", "".join(traceback.format_exception_only(type(e), e)).strip()[:400])
        return False
    return True


fails = 0
if len(sys.argv) > 3 and sys.argv[3] == "features":
    skipped = 0
    for case in range(cases):
        r = feature_case(rng)
        fails += r is False
        skipped += r is None
    if GL_PAIRS:   # the statistical gate of tests/test_gpu_parity.py::test_griffin_lim_defaults_statistical_gate over the sweep's draws
        d, y = np.asarray(GL_PAIRS).T
        ok = bool(np.median(d) <= 16.0 * np.median(y) and d.max() <= 16.0 * y.max())
        print("griffin_lim: %d draws, device median %.3g / max %.3g, float32-storage oracle median %.3g / max %.3g: %s"
              % (len(d), np.median(d), d.max(), np.median(y), y.max(), "inside 16 x the yardstick" if ok else "OUTSIDE 16 x the yardstick"))
        fails += not ok
    print("%d feature cases (%d drew a filterbank the reference rejects), %d failures" % (cases, skipped, fails))
    sys.exit(min(fails, 100))
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
        if not f64 and rng.random() < 0.4:
            # the device-resident entry points (torch tensors in, torch tensors out) are the same kernels: bit-equal
            import torch
            xd = torch.from_numpy(x).cuda()
            zd = Stft.transform(c, xd)
            assert zd.is_cuda and np.array_equal(zd.cpu().numpy(), z), "device transform differs from the host entry point"
            pd = Stft.power_spectrum(c, xd, power)
            assert np.array_equal(pd.cpu().numpy(), Stft.power_spectrum(c, x, power)), "device power differs from the host entry point"
            if Stft.nola(c) and Stft.frames(c, n) > 0:
                xi = Stft.invert(c, zd)
                assert np.array_equal(xi.cpu().numpy(), Stft.invert(c, z)), "device invert differs from the host entry point"
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
                # p < 1: every near-zero bin under a filter carries an error of the same sign (d^p), so the floor grows with the
                # filter's width in bins (a 41-sample signal edge-padded into fft 1200 is almost all such bins)
                width = max(1.0, (fft // 2 + 1) / mc.n_mels)
                # the float32 interior's error scales with the SPECTRUM's peak, which a filterbank may not see at all (a
                # constant pad puts everything into bin 0, whose weight is 0): floor from the spectrum's peak through
                # the widest filter, at 2e-7 of that peak per bin (seed 2026 drew such a case: fuzz_stft.log of round 4)
                wsum = float(np.max(np.sum(np.abs(om.weights), axis=-1)))
                floor = 2e-7 * float(np.max(np.abs(wz))) ** power * wsum if power >= 1.0 and wz.size else 0.0
                close(S.mel_spectrogram(c, mc, x, power), O.mel_spectrogram(o, om, x, power), 1e-5,
                      1e-5 if power >= 1.0 else min(0.05, 4 * 1e-5 ** power * width), "mel", floor)
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

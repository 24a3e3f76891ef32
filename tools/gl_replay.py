"""Replays a Griffin-Lim draw kept by tools/fuzz_parity.py (FUZZ_DUMP_GL): distance to the float64 oracle after 1 .. n_iter iterations
for the shipped synthesis kernel and for the one of rounds 1-4, and where in the signal the difference sits.  python tools/gl_replay.py case.npz"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import soundml_oracle as O
import soundml_amd as S
from soundml_amd import Stft
d = np.load(sys.argv[1])
mag, init, fft, hop, n_iter, mom = d["mag"], d["init"], int(d["fft"]), int(d["hop"]), int(d["n_iter"]), float(d["mom"])
c, o = Stft.Config.create(fft_size=fft, hop=hop), O.stft_config(fft, hop=hop)
print("mag", mag.shape, "n_iter", n_iter, "momentum", mom)
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
for it in range(1, n_iter + 1):
    want = O.griffin_lim(o, mag, n_iter=it, momentum=mom, init=init).astype(np.float64)
    row = []
    for env in ("1", "0"):
        os.environ["SMX_INVERT_PIPELINE"] = env
        got = Stft.griffin_lim(c, mag, n_iter=it, momentum=mom, init=init)
        row.append(rel(got, want))
        if it == n_iter:
            e = np.abs(got.astype(np.float64) - want).reshape(-1, got.shape[-1])
            k = np.unravel_index(np.argmax(e), e.shape)
            print("   kernel %s: max |err| %.3e at clip %d sample %d of %d (peak %.3f); err energy by 2048-sample block:" % ("pipeline" if env == "1" else "rounds 1-4", e.max(), k[0], k[1], e.shape[1], np.abs(want).max()),
                  " ".join("%.1e" % v for v in np.sqrt((e[k[0]] ** 2).reshape(-1)[: (e.shape[1] // 2048) * 2048].reshape(-1, 2048).sum(axis=1))[:24]))
    S.set_interior("float64")
    strict = Stft.griffin_lim(c, mag, n_iter=it, momentum=mom, init=init)
    S.set_interior("float32")
    print("iterations %d: pipeline %.3e | rounds 1-4 %.3e | float64 interior %.3e" % (it, row[0], row[1], rel(strict, want)))

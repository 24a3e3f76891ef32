"""Griffin-Lim at the reference's defaults (32 iterations, momentum 0.99: stft.ml:961-1017) over MANY seeds: the distribution of
the device's distance to the float64 oracle beside the yardstick -- the ORACLE ITSELF with nothing but its stored intermediates
rounded to float32 (O.griffin_lim_float32_storage).  The statistical gate of tests/test_gpu_parity.py
(test_griffin_lim_defaults_statistical_gate) takes its bound from this table.
  python tools/gl_stat.py [seeds=48] [first_seed=1000]      GL_ITERS / GL_MOMENTUM / GL_N / GL_INTERIOR=float64 vary the case"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import soundml_oracle as O
import soundml_amd as S
from soundml_amd import Stft


rounded_gl = O.griffin_lim_float32_storage   # the yardstick (oracle/soundml_oracle.py)


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    iters = int(os.environ.get("GL_ITERS", "32"))
    momentum = float(os.environ.get("GL_MOMENTUM", "0.99"))
    n = int(os.environ.get("GL_N", "24000"))
    if os.environ.get("GL_INTERIOR") == "float64":
        S.set_interior("float64")
    c = Stft.Config.create(fft_size=2048, hop=512)
    o = O.stft_config(2048, hop=512)
    rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
    dev, yard, conv_d, conv_w = [], [], [], []
    for seed in range(first, first + seeds):
        rng = np.random.default_rng(seed)
        x = rng.uniform(-1, 1, size=(1, n)).astype(np.float32)
        mag = np.abs(Stft.transform(c, x)).astype(np.float32)
        phase = rng.uniform(-np.pi, np.pi, size=mag.shape).astype(np.float32)
        want = O.griffin_lim(o, mag, iters, momentum, phase, None).astype(np.float64)
        got = Stft.griffin_lim(c, mag, n_iter=iters, momentum=momentum, init=phase)
        stored = rounded_gl(o, mag, iters, momentum, phase)
        conv = lambda y: float(np.linalg.norm(np.abs(O.transform(o, y.astype(np.float64))) - mag) / np.linalg.norm(mag))
        dev.append(rel(got, want)); yard.append(rel(stored, want)); conv_d.append(conv(got)); conv_w.append(conv(want))
        print("seed %d: device %.3e | float32-storage oracle %.3e | convergence device %.5f oracle %.5f" % (seed, dev[-1], yard[-1], conv_d[-1], conv_w[-1]), flush=True)
    q = lambda v, p: float(np.quantile(np.asarray(v), p))
    table = {k: {"min": min(v), "q25": q(v, 0.25), "median": q(v, 0.5), "q75": q(v, 0.75), "q90": q(v, 0.9), "max": max(v),
                 "geomean": float(np.exp(np.mean(np.log(np.asarray(v) + 1e-300))))} for k, v in (("device", dev), ("float32_storage_oracle", yard))}
    table["case"] = {"seeds": seeds, "first_seed": first, "iterations": iters, "momentum": momentum, "n": n, "fft": 2048, "hop": 512,
                     "interior": os.environ.get("GL_INTERIOR", "float32"),
                     "max_convergence_ratio_device_over_oracle": max(d / w for d, w in zip(conv_d, conv_w))}
    table["per_seed"] = {"device": dev, "float32_storage_oracle": yard}
    print(json.dumps(table))


if __name__ == "__main__":
    main()

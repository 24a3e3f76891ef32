"""The host-pointer faces at C2 size (numpy in, fresh numpy out: what an nx caller gets), PCIe inclusive.  Round 6: one script for
what host_path_time / host_path_ab / host_path_pinned / host_path_trace / host_invert_time measured.
  python tools/host_path.py [--clips 256] [--n 480000] [--calls 12] [--devices 0,0] MODE...
MODES (each interleaves its two settings in one process, the result dropped outside the timed region as nx's GC would):
  pipeline   the clip-unit pipeline on / off (SMX_HOST_PIPELINE=1 / 0), Stft.power_spectrum
  pinned     the result from the page-locked pool / an ordinary array, Stft.power_spectrum and Stft.transform
  invert     Stft.invert of the C2 spectrogram, result page-locked / ordinary
  sharded    a device list (--devices, default "0,0") against the single device (smx_set_devices), Stft.power_spectrum
  trace      one call of each face under SMX_HOST_TRACE=1 (the stage times of the pipelined call on stderr)"""
import argparse, gc, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=256)
ap.add_argument("--n", type=int, default=480000)
ap.add_argument("--calls", type=int, default=12)
ap.add_argument("--devices", default="0,0")
ap.add_argument("modes", nargs="+")
a = ap.parse_args()
if "trace" in a.modes:
    os.environ["SMX_HOST_TRACE"] = "1"
import soundml_amd as S
from soundml_amd import Stft
x = np.random.default_rng(0).uniform(-1, 1, size=(a.clips, a.n)).astype(np.float32)
c = Stft.Config.create(fft_size=2048, hop=512)


def ab(name, settings, call, check=True):
    """settings: [(label, apply)]; interleaved; first call of each untimed"""
    ts, ref = {lab: [] for lab, _ in settings}, None
    for rep in range(a.calls + 1):
        for lab, apply in settings:
            apply()
            t0 = time.perf_counter(); y = call(); dt = (time.perf_counter() - t0) * 1e3
            if rep == 0 and check:
                if ref is None: ref = y[::37].copy()
                else: assert np.array_equal(ref, y[::37]), (name, lab)
            elif rep:
                ts[lab].append(dt)
            del y; gc.collect()
    for lab, _ in settings:
        v = sorted(ts[lab])
        print("%-12s %-34s min %.1f  q1 %.1f  median %.1f  q3 %.1f  max %.1f ms   (%s)" % (name, lab, v[0], v[len(v) // 4], v[len(v) // 2], v[3 * len(v) // 4], v[-1], " ".join("%.0f" % t for t in ts[lab])), flush=True)


env = lambda k, v: (lambda: os.environ.__setitem__(k, v))
for mode in a.modes:
    if mode == "pipeline":
        ab("power", [("pipelined (default)", env("SMX_HOST_PIPELINE", "1")), ("serial (SMX_HOST_PIPELINE=0)", env("SMX_HOST_PIPELINE", "0"))], lambda: Stft.power_spectrum(c, x))
        os.environ.pop("SMX_HOST_PIPELINE", None)
    elif mode == "pinned":
        pin = [("page-locked result (pool)", lambda: S.set_pinned_results(True)), ("ordinary numpy result", lambda: S.set_pinned_results(False))]
        ab("power", pin, lambda: Stft.power_spectrum(c, x))
        ab("transform", pin, lambda: Stft.transform(c, x))
        S.set_pinned_results(True)
    elif mode == "invert":
        z = Stft.transform(c, x)
        ab("invert", [("page-locked result (pool)", lambda: S.set_pinned_results(True)), ("ordinary numpy result", lambda: S.set_pinned_results(False))],
           lambda: Stft.invert(c, z, length=a.n))
        S.set_pinned_results(True); del z
    elif mode == "sharded":
        devs = [int(d) for d in a.devices.split(",")]
        ab("power", [("one device", lambda: S.set_devices([])), ("devices %s" % devs, lambda: S.set_devices(devs))], lambda: Stft.power_spectrum(c, x))
        S.set_devices([])
    elif mode == "trace":
        for name, call in (("power_spectrum", lambda: Stft.power_spectrum(c, x)), ("transform", lambda: Stft.transform(c, x))):
            for _ in range(2):
                print("-- %s" % name, file=sys.stderr, flush=True); y = call(); del y
    else:
        raise SystemExit("unknown mode %r" % mode)

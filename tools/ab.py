"""ONE runner for every in-process A/B timing (round 6: replaces ab.py / ab_env.py / ab_mel.py / ab_mel_env.py / ab_transform_env.py /
ab_lanes_skew.py).  Variants are builds and / or environments, interleaved in one process on one device on the same data -- never
compare separate runs or boxes (boxes differ by +-10 % on one binary: the board's power cap leaves every chip a different clock).

  python tools/ab.py [--face power|transform|mel|invert|fir] [--fft 2048] [--hop 0] [--clips 256] [--n 480000] [--power 2.0]
                     [--mels 128] [--sr 48000] [--interior float32|float64] [--rounds 30] [--energy] VARIANT [VARIANT ...]

  VARIANT = "[path/to/libsoundml_amd.so][@VAR=value[,VAR=value...]]"     "" = the shipped build, default environment
            e.g.  ""  "@SMX_POWER_SKEW=0"  "soundml_amd/lib_x/libsoundml_amd.so"  "soundml_amd/lib_diag/libsoundml_amd.so@SMX_NOSTORE=1"
            (options the library reads per launch; a switch read once per process needs a process of its own)

Prints a checksum per variant and whether its output equals the first variant's bit for bit, then min / quartiles of the
interleaved rounds (HIP events around 4 launches).  --energy: afterwards every variant runs back to back for ~2.5 s while rocm-smi
is sampled: package W, shader clock, and JOULES PER LAUNCH = W x ms -- the headline kernel sits at the board's 1400 W cap, where
time follows joules per frame, so an A/B of that kernel is read in joules as well as in milliseconds."""
import argparse, ctypes, os, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
i64, vp, ci, f64 = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_double
ap = argparse.ArgumentParser()
ap.add_argument("--face", default="power", choices=("power", "transform", "mel", "invert", "fir"))
ap.add_argument("--fft", type=int, default=2048)
ap.add_argument("--hop", type=int, default=0)
ap.add_argument("--clips", type=int, default=0)
ap.add_argument("--n", type=int, default=0)
ap.add_argument("--power", type=float, default=2.0)
ap.add_argument("--mels", type=int, default=128)
ap.add_argument("--sr", type=int, default=48000)
ap.add_argument("--taps", type=int, default=8192)
ap.add_argument("--interior", default="float32")
ap.add_argument("--rounds", type=int, default=30)
ap.add_argument("--energy", action="store_true")
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
fft, hop = a.fft, a.hop or a.fft // 4
fir = a.face == "fir"
clips = a.clips or (8 if fir else 256)
n = a.n or (2880000 if fir else 480000)
frames, bins = 1 + n // hop, fft // 2 + 1
x = torch.rand(clips, n, device="cuda") * 2 - 1
if a.face == "power": out = torch.empty(clips, bins, frames, device="cuda")
elif a.face == "transform": out = torch.empty(clips, bins, frames, 2, device="cuda")
elif a.face == "mel": out = torch.empty(clips, a.mels, frames, device="cuda")
elif a.face == "invert":
    z = torch.randn(clips, bins, frames, 2, device="cuda")
    z[:, 0, :, 1] = 0; z[:, -1, :, 1] = 0
    out = torch.empty(clips, n, device="cuda")
else: out = torch.empty_like(x)
units = clips * (n if fir else frames)


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path or os.path.join(ROOT, "soundml_amd", "lib", "libsoundml_amd.so")))
    lib.smx_last_error.restype = ctypes.c_char_p
    h, mh, plan = vp(), vp(), vp()
    lib.smx_stft_config_create.argtypes = [i64, i64, i64, ci, ci, f64, ci, ci, vp, ctypes.POINTER(vp)]
    assert lib.smx_stft_config_create(fft, -(2 ** 63), hop, 0, 0, 0.0, 0, 0, None, ctypes.byref(h)) == 0, lib.smx_last_error()
    if a.interior == "float64": assert lib.smx_set_interior(1) == 0
    if a.face == "power":
        lib.smx_stft_power_range_f32_dev.argtypes = [vp, vp, i64, i64, i64, i64, i64, f64, vp, vp]
        return lambda: lib.smx_stft_power_range_f32_dev(h, vp(x.data_ptr()), clips, n, n, 0, frames, a.power, vp(out.data_ptr()), None)
    if a.face == "transform":
        lib.smx_stft_transform_range_f32_dev.argtypes = [vp, vp, i64, i64, i64, i64, i64, vp, vp]
        return lambda: lib.smx_stft_transform_range_f32_dev(h, vp(x.data_ptr()), clips, n, n, 0, frames, vp(out.data_ptr()), None)
    if a.face == "mel":
        lib.smx_mel_config_create.argtypes = [i64, i64, i64, f64, ci, f64, ci, ci, ctypes.POINTER(vp)]
        assert lib.smx_mel_config_create(a.mels, a.sr, fft, 0.0, 0, 0.0, 0, 0, ctypes.byref(mh)) == 0, lib.smx_last_error()
        lib.smx_mel_spectrogram_f32_dev.argtypes = [vp, vp, vp, i64, i64, i64, f64, vp, vp]
        return lambda: lib.smx_mel_spectrogram_f32_dev(h, mh, vp(x.data_ptr()), clips, n, n, a.power, vp(out.data_ptr()), None)
    if a.face == "invert":
        lib.smx_stft_invert_f32_dev.argtypes = [vp, vp, i64, i64, i64, ci, i64, vp, vp]
        return lambda: lib.smx_stft_invert_f32_dev(h, vp(z.data_ptr()), clips, bins, frames, 1, n, vp(out.data_ptr()), None)
    taps = (ctypes.c_double * a.taps)()
    beta = f64()
    lib.smx_fir_kaiser_beta.argtypes = [f64, ctypes.POINTER(f64)]
    assert lib.smx_fir_kaiser_beta(100.0, ctypes.byref(beta)) == 0
    lib.smx_fir_design_lowpass.argtypes = [i64, f64, f64, vp]
    assert lib.smx_fir_design_lowpass(a.taps, 0.25, beta.value, taps) == 0, lib.smx_last_error()
    lib.smx_fir_plan_create.argtypes = [vp, i64, ctypes.POINTER(vp)]
    assert lib.smx_fir_plan_create(taps, a.taps, ctypes.byref(plan)) == 0, lib.smx_last_error()
    lib.smx_fir_apply_f32_dev.argtypes = [vp, vp, i64, i64, i64, vp, i64, vp]
    return lambda: lib.smx_fir_apply_f32_dev(plan, vp(x.data_ptr()), clips, n, n, vp(out.data_ptr()), n, None)


def env_of(v):
    return [kv.split("=", 1) for kv in filter(None, (v.split("@", 1)[1] if "@" in v else "").split(","))]


class Env:
    def __init__(self, v): self.kv = env_of(v)
    def __enter__(self):
        for k, val in self.kv: os.environ[k] = val
    def __exit__(self, *exc):
        for k, _ in self.kv: os.environ.pop(k, None)


calls = {}
for v in a.variants:
    path = v.split("@", 1)[0]
    with Env(v):
        calls[v] = load(path)
        for _ in range(3): assert calls[v]() == 0
torch.cuda.synchronize()
ref = None
for v in a.variants:   # do the variants agree?
    with Env(v):
        out.zero_(); assert calls[v]() == 0; torch.cuda.synchronize()
    name = v or "(shipped build, default environment)"
    line = "%-60s checksum %016x" % (name[-60:], int(out.view(torch.int32).to(torch.int64).sum().item()) & (2 ** 64 - 1))
    if ref is None: ref = out.clone()
    else: line += "  vs first: bit-identical %s, max |diff| / peak %.3g" % (bool(torch.equal(ref, out)), float((ref - out).abs().max() / ref.abs().max()))
    print(line, flush=True)
del ref
ts = {v: [] for v in a.variants}
for rnd in range(a.rounds):
    for v in a.variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with Env(v):
            e0.record()
            for _ in range(4): calls[v]()
            e1.record(); torch.cuda.synchronize()
        ts[v].append(e0.elapsed_time(e1) / 4)
unit = "Gsamples/s" if fir else "Mframes/s"
for v in a.variants:
    t = sorted(ts[v])
    print("%-60s min %.4f  q1 %.4f  median %.4f  q3 %.4f ms  (%.1f %s at median)"
          % ((v or "(shipped build, default environment)")[-60:], t[0], t[len(t) // 4], t[len(t) // 2], t[3 * len(t) // 4], units / t[len(t) // 2] / (1e6 if fir else 1e3), unit), flush=True)

if a.energy:
    child_env = {k: val for k, val in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP_", "ROCPROF", "HSA_TOOLS_"))}

    def sample(got, stop):
        time.sleep(1.0)   # (the package power rocm-smi reports lags the load by a few hundred ms)
        while not stop.is_set():
            try:
                r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5, env=child_env)
                rows = [ln.split(",") for ln in r.stdout.strip().splitlines() if "," in ln]
                hdr, val = rows[0], rows[1]
                col = lambda key: next((val[i] for i, hname in enumerate(hdr) if key in hname.lower()), None)
                got.append((float(col("power (w)")), int("".join(ch for ch in col("sclk clock speed") if ch.isdigit()))))
            except Exception:   # noqa: BLE001
                pass
            time.sleep(0.2)
    print("energy: each variant back to back for ~2.5 s, rocm-smi sampled from 1 s on (idle board: ~300 W)")
    for v in a.variants:
        got, stop = [], threading.Event()
        th = threading.Thread(target=sample, args=(got, stop)); th.start()
        t_end, ms = time.time() + 2.6, []
        with Env(v):
            while time.time() < t_end:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): calls[v]()
                e1.record(); torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1) / 20)
        stop.set(); th.join()
        tail = sorted(ms[len(ms) // 2:]); m = tail[len(tail) // 2]
        if got:
            w = sorted(g[0] for g in got)[len(got) // 2]; clk = sorted(g[1] for g in got)[len(got) // 2]
            print("%-60s %.4f ms  %6.0f W  %4d MHz  %.4f J per launch  %.0f nJ per %s" % ((v or "(shipped build, default environment)")[-60:], m, w, clk, w * m * 1e-3, w * m * 1e-3 / units * 1e9, "sample" if fir else "frame"), flush=True)
        else:
            print("%-60s %.4f ms  (rocm-smi gave no sample)" % ((v or "(shipped)")[-60:], m))
        time.sleep(1.0)

"""In-process A/B timing of two builds of libsoundml_amd.so (interleaved rounds, same device, same data):
  python tools/ab.py soundml_amd/lib_a/libsoundml_amd.so soundml_amd/lib/libsoundml_amd.so
(methodology: per-variant median and min over interleaved rounds; never compare separate runs/boxes).
AB_INTERIOR=float64 times the float64 interior instead; AB_CLIPS / AB_N / AB_ROUNDS size the run."""
import ctypes, os, sys
import torch
i64, vp = ctypes.c_int64, ctypes.c_void_p
paths = sys.argv[1:]
N_ENV = int(os.environ.get("AB_N", "480000"))   # AB_N=482816 gives 944 frames: 64-byte aligned output rows
libs = []
clips, n = int(os.environ.get("AB_CLIPS", "256")), N_ENV
frames = 1 + n // 512
x = torch.rand(clips, n, device="cuda") * 2 - 1
out = torch.empty(clips, 1025, frames, device="cuda")
# "path@VAR=val" sets an environment variable around that variant's launches (for options read per launch)
for p in paths:
    lib = ctypes.CDLL(os.path.abspath(p.split("@")[0]))
    h = vp()
    lib.smx_stft_config_create.argtypes = [i64, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(vp)]
    assert lib.smx_stft_config_create(2048, -(2**63), 512, 0, 0, 0.0, 0, 0, None, ctypes.byref(h)) == 0
    lib.smx_stft_power_range_f32_dev.argtypes = [vp, vp, i64, i64, i64, i64, i64, ctypes.c_double, vp, vp]
    if os.environ.get("AB_INTERIOR") == "float64":   # the reference's interior (window, transform and |.|^p in float64)
        assert lib.smx_set_interior(1) == 0
    libs.append((p, lib, h))
def setenv(p):
    for kv in p.split("@")[1:]:
        k, v = kv.split("=")
        os.environ[k] = v
def clearenv(p):
    for kv in p.split("@")[1:]:
        os.environ.pop(kv.split("=")[0], None)
def run(lib, h):
    assert lib.smx_stft_power_range_f32_dev(h, vp(x.data_ptr()), clips, n, n, 0, frames, 2.0, vp(out.data_ptr()), None) == 0
for p, lib, h in libs:
    setenv(p)
    for _ in range(3): run(lib, h)
    clearenv(p)
torch.cuda.synchronize()
# do the builds agree?  (bit for bit, and the largest difference relative to the output's peak)
ref = None
for p, lib, h in libs:
    setenv(p); out.zero_(); run(lib, h); torch.cuda.synchronize(); clearenv(p)
    print("%-52s checksum %016x" % (p[-52:], int(out.view(torch.int32).to(torch.int64).sum().item()) & (2**64 - 1)))
    if ref is None:
        ref = out.clone()
    else:
        print("%-52s vs first: bit-identical %s, max |diff| / peak %.3g" % (p[-52:], bool(torch.equal(ref, out)), float((ref - out).abs().max() / ref.abs().max())))
del ref
ts = {p: [] for p in paths}
for rnd in range(int(os.environ.get("AB_ROUNDS", "40"))):
    for p, lib, h in libs:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        setenv(p)
        a.record()
        for _ in range(4): run(lib, h)
        b.record(); torch.cuda.synchronize()
        clearenv(p)
        ts[p].append(a.elapsed_time(b) / 4)
for p in paths:
    v = sorted(ts[p])
    print("%-52s min %.4f  q1 %.4f  median %.4f  q3 %.4f ms  (%.1f Mframes/s at median)"
          % (p[-52:], v[0], v[len(v) // 4], v[len(v) // 2], v[3 * len(v) // 4], clips * frames / v[len(v) // 2] / 1e3))

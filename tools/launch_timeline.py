"""Where one launch of the fft-2048 power kernel spends its time OUTSIDE the steady tile loop (round 5).
A `make COARSE=1` build (soundml_amd/lib_clock) records the chip-wide 100 MHz clock (s_memrealtime) per workgroup and wave at:
kernel entry, tables in / first samples requested, tiles 1 and 2, the last tile, loop exit, last flush issued, kernel exit (after
the border epilogue, stores drained).  The kernels otherwise run as shipped.
  python tools/launch_timeline.py [clips n_samples]...        (default: C2 = 256 x 480000, and 512 x 1440000 = an eighth of C5)
Prints, per shape: HIP-event time per launch (back-to-back), the in-kernel span (last exit - first entry), and the attribution:
dispatch stagger, prologue, first tile, steady tiles, last tile + drain, epilogue, idle tail (workgroups done before the last)."""
import ctypes, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.environ.get("TL_LIB", os.path.join(ROOT, "soundml_amd", "lib_clock", "libsoundml_amd.so")))
i64, vp = ctypes.c_int64, ctypes.c_void_p
h = vp()
lib.smx_stft_config_create.argtypes = [i64, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(vp)]
assert lib.smx_stft_config_create(2048, -(2**63), 512, 0, 0, 0.0, 0, 0, None, ctypes.byref(h)) == 0
lib.smx_stft_power_range_f32_dev.argtypes = [vp, vp, i64, i64, i64, i64, i64, ctypes.c_double, vp, vp]
shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or [(256, 480000), (512, 1440000)]
S, nwg = 24, 256
for clips, n in shapes:
    frames = 1 + n // 512
    x = torch.rand(clips, n, device="cuda") * 2 - 1
    out = torch.empty(clips, 1025, frames, device="cuda")
    def run():
        assert lib.smx_stft_power_range_f32_dev(h, vp(x.data_ptr()), clips, n, n, 0, frames, 2.0, vp(out.data_ptr()), None) == 0
    t0 = time.time()
    while time.time() - t0 < 1.5:      # sustained state (profiles/r06/step_time_transient.log)
        for _ in range(10): run()
        torch.cuda.synchronize()
    reps = 20
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): run()
    b.record(); torch.cuda.synchronize()
    ev_ms = a.elapsed_time(b) / reps
    singles = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize(); singles.append(a.elapsed_time(b))
    buf = np.zeros(nwg * 16 * S, dtype=np.uint64)
    assert lib.smx_debug_read_stamps(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), buf.size) == 0
    st = buf.reshape(nwg, 16, S)[:, :8, :].astype(np.int64)
    us = lambda ticks: np.asarray(ticks, dtype=np.float64) / 100.0   # 100 MHz -> microseconds
    entry, ready, t1, t2, tlast, loop_end, flushed, done = (st[:, :, k] for k in (15, 16, 9, 10, 11, 17, 18, 19))
    ntiles = st[:, 0, 22]
    nborder = st[:, 0, 23]
    k0 = entry.min()
    span = us(done.max() - k0)
    wg_entry = us(entry.min(axis=1) - k0)
    wg_ready = us(ready.max(axis=1) - entry.min(axis=1))
    wg_first = us(t1.max(axis=1) - ready.max(axis=1))
    wg_second = us(t2.max(axis=1) - t1.max(axis=1))
    steady_tiles = np.maximum(ntiles - 2, 1)
    wg_steady = us(tlast.max(axis=1) - t1.max(axis=1))          # tiles 1 .. ntiles - 2
    wg_last = us(flushed.max(axis=1) - tlast.max(axis=1))        # the last tile + its flush
    wg_epi = us(done.max(axis=1) - flushed.max(axis=1))
    wg_done = us(done.max(axis=1) - k0)
    per_tile = wg_steady / steady_tiles
    print("== %d clips x %d samples: %d frames/clip, tiles per workgroup %d..%d, border tiles on %d workgroups"
          % (clips, n, frames, ntiles.min(), ntiles.max(), int((nborder > 0).sum())))
    print("   HIP events: %.4f ms per launch back-to-back (%d), single launches median %.4f min %.4f; in-kernel span %.1f us  -> outside the kernel %.1f us"
          % (ev_ms, reps, sorted(singles)[reps // 2], min(singles), span, ev_ms * 1000 - span))
    q = lambda v: "mean %.1f  median %.1f  p90 %.1f  max %.1f" % (np.mean(v), np.median(v), np.percentile(v, 90), np.max(v))
    print("   entry after the first workgroup's entry, us: " + q(wg_entry))
    print("   prologue (entry -> tables in, first samples requested), us: " + q(wg_ready))
    pieces = [us(st[:, :, k]).mean() for k in range(5)]   # (COARSE builds leave slots 0-4 to the prologue)
    print("      of it, mean over waves: table requests issued %.2f, tile walk %.2f, first samples requested %.2f, tables into LDS (their arrival) %.2f, barrier + loop set-up %.2f"
          % tuple(pieces))
    print("   tile 0 (until every wave starts tile 1), us: " + q(wg_first))
    print("   tile 1, us: " + q(wg_second))
    print("   steady tile, us: " + q(per_tile) + "   x tiles = %.1f us mean" % np.mean(per_tile * ntiles))
    print("   last tile + final flush, us: " + q(wg_last))
    print("   epilogue + store drain, us: " + q(wg_epi) + "   (workgroups with a border tile: %s)" % (q(wg_epi[nborder > 0]) if (nborder > 0).any() else "none"))
    print("   workgroup finished at, us after kernel start: " + q(wg_done) + "  min %.1f" % wg_done.min())
    ideal = float(np.mean(per_tile)) * float(np.mean(ntiles))
    print("   => span %.1f us = steady-rate work %.1f us (mean tile x mean tiles) + %.1f us (%.1f %%) of entry stagger / prologue / first + last tile / epilogue / imbalance"
          % (span, ideal, span - ideal, 100 * (span - ideal) / span))
    by_xcd = [np.mean(per_tile[np.arange(nwg) % 8 == k]) for k in range(8)]
    print("   steady tile by XCD (blockIdx %% 8), us: " + " ".join("%.2f" % v for v in by_xcd))
    slow = np.argsort(-wg_done)[:8]
    print("   the 8 last workgroups: " + ", ".join("wg %d: tile %.2f us, border %d, done %.1f" % (w, per_tile[w], nborder[w], wg_done[w]) for w in slow))
    del x, out

#!/usr/bin/env python3
"""Benchmark of the hot path: STFT power spectrogram, n_fft=2048 hop=512 Hann,
float32, on synthetic device-resident audio (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

One process per GPU.  Started without RANK in the environment and with N > 1, this
process is only a launcher: it starts N fresh rank processes (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1) BEFORE importing torch or the
library -- the parent makes no GPU call -- forwards rank 0's JSON line and exits
with the worst child status.  Under `torch.distributed.run` (RANK set) it is a rank.

Workloads (BASELINE.json `configs`).  `value` is the SAME workload at every N, so that a scaling
efficiency computed from the per-N lines compares like with like:
  every N C2 per GPU: 256 clips x 10 s x 48 kHz resident on each GPU (`scaling` = "weak": rank r
          owns clips [256 r, 256 r + 256) of a 256 N clip job; no collective on the data path);
          a step = one Stft.power_spectrum pass over the rank's batch through the C ABI = ONE
          launch of stft2048_power32_kernel (asserted through the library's launch counter).
  N = 1   `extra` carries C3 (fused mel, 128 mels), C4 (FIR 8192 taps, 8 x 60 s), C1's geometry,
          Stft.transform and the one-GPU point of C5, each timed with HIP events the same way.
  N > 1   `extra.c5_strong` is BASELINE configs[4]: 4096 clips x 30 s sharded by contiguous clip
          ranges (soundml_amd.shard.clip_range; 512 clips per GPU at 8), strong scaling; its
          one-GPU point is `extra.c5_one_gpu` of the N = 1 line.  (--workload c5 makes it `value`.)
The line also says what the ranks saw: `world_size_seen`, the backend, and the per-rank kernel
time (min / max over ranks).
The only communication is the timing barrier and the MAX-over-ranks reduction of
the clock (RCCL when every rank has its own GPU; gloo when ranks share a device,
which RCCL refuses -- the 2-rank test on a 1-GPU box).

Timing: W untimed warm-up steps, then exactly K steps between barrier + synchronize.  Before the warm-up steps the same step
runs in groups of 8 event-timed launches until the last 8 lie within 1 % of each other, their mean lies within 0.3 % of the mean 128 launches
earlier, and --preroll-ms (default 500) have passed (at most 2 x --preroll-ms)
(untimed, uncounted; reported as `config.preconditioning` and, with the W warm-up steps, as `warmup_effective`): out of an idle device the
kernel's time is not stationary (0.51 ms for three launches, 0.61-0.64 for the next ten, the sustained 0.50 after ~40 --
profiles/r06/step_time_transient.log), and `--steps 20 --warmup 5` would sample that hump.  `value` is the sustained rate;
`extra.c2_burst_from_idle` is the same K / W measurement out of an idle device, in every run.

Rank 0 prints ONE JSON line: whole-job Mframes/s, the HBM roofline of the dominant
kernel (HIP events on the launch stream) and, at N = 1, the CPU baseline: the
oracle's C restatement (float64 interior, all host cores) on the SAME C2 batch, whose
output also checks every frame of the GPU result.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FFT, HOP, SR, BINS = 2048, 512, 48000, 1025
ALGO_BYTES_PER_FRAME = HOP * 4 + BINS * 4      # hop*4 read + bins*4 written = 6148 B
MEL_BYTES_PER_FRAME = HOP * 4 + 128 * 4        # fused mel from audio: 2560 B
FIR_BYTES_PER_SAMPLE = 8                       # 4 in + 4 out
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_GBS = 6290.0                      # MI355X_MICROARCH.md: what a float4 copy reaches (79 % of the spec)
MFMA_F32_PEAK_TFLOPS = 157.3                   # MI355X_MICROARCH.md: v_mfma_f32_*_f32


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preroll-ms", type=float, default=500.0,
                    help="device preconditioning before the W warm-up steps: the same step repeated for this long (not timed, not counted; "
                         "0 = none).  A burst from idle runs up to 25 %% slower between its 5th and 25th launch while the power "
                         "management settles (profiles/r06/step_time_transient.log); the metric is the sustained rate")
    ap.add_argument("--workload", choices=("auto", "c2", "c5"), default="auto",
                    help="auto = c2: BASELINE configs[1] per GPU at every N (weak); c5: configs[4] sharded (strong)")
    ap.add_argument("--clips", type=int, default=0,
                    help="override the clip count (C2: per GPU, default 256; C5: whole job, default 4096)")
    ap.add_argument("--seconds", type=float, default=0.0, help="override the clip length (C2: 10 s, C5: 30 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="rank plumbing only (no GPU, no library): rendezvous over gloo, barrier, MAX-reduce, one JSON line")
    ap.add_argument("--verify-shards", action="store_true",
                    help="every rank also computes the whole batch and checks shard == slice bit for bit (small sizes)")
    return ap.parse_args()


# ---- launcher (no torch, no GPU call) ---------------------------------------------------------------
def launch_ranks(n):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread; every child is polled, and the first failure ends the others (a rank that
    # died before or inside a collective would otherwise leave the rest in a 10-30 minute collective timeout)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    worst = 0
    deadline = time.time() + float(os.environ.get("SMX_BENCH_RANK_TIMEOUT", "3000"))
    alive = list(procs)
    while alive:
        for p in list(alive):
            rc = p.poll()
            if rc is not None:
                alive.remove(p)
                worst = worst or rc
        if worst or time.time() > deadline:
            for p in alive:
                p.kill()          # the exact children we started
                p.wait()
            worst = worst or 124
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    out = b"".join(c for c in chunks if c)
    for ln in out.decode().splitlines():   # ONE JSON line on stdout; whatever else a library printed there (gloo's banner) goes to stderr
        (sys.stdout if ln.lstrip().startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    return worst             # a failed rank is a failed run: no retry, never a re-exec


# ---- CPU baseline + full parity (rank 0, N = 1) ---------------------------------------------------------
def cpu_baseline(x_host, gpu_out_host):
    """oracle/oracle_stft.c (float64 interior, clip-parallel pthreads) on the whole C2 batch that the GPU
    just processed; its output checks every frame of the GPU result (north_star: 1e-5 relative).
    Method (BASELINE.md section 3.2's discipline: medians after warm-ups): one untimed whole-batch run, then the MEDIAN of five
    whole-batch runs into an output array whose pages are already mapped; the single-thread row on 32 clips (median of three
    after a warm-up) -- a 0.1 s sample on a core that was idle measured the core's wake-up, not the code (round 5: 16 threads
    read 35x one thread).  `consistency` puts the two side by side with the CPU seconds each really consumed (getrusage)."""
    import resource
    import numpy as np
    from oracle import c_oracle, soundml_oracle as O
    cores = c_oracle.effective_cpus()     # affinity and cgroup quota, not os.cpu_count(): the GPU boxes show 256 CPUs, quota 16
    c = O.stft_config(FFT, hop=HOP)
    clips, n = x_host.shape
    frames = clips * O.frames(c, n)
    want = np.zeros((clips, BINS, O.frames(c, n)), dtype=np.float32)

    def cpu_s():
        r = resource.getrusage(resource.RUSAGE_SELF)
        return r.ru_utime + r.ru_stime
    c_oracle.stft(c, x_host, 2.0, threads=cores, out=want)      # warm-up: library paged in, output pages mapped, cores awake
    runs, cpu_runs = [], []
    for _ in range(5):
        c0, t0 = cpu_s(), time.perf_counter()
        c_oracle.stft(c, x_host, 2.0, threads=cores, out=want)
        runs.append(time.perf_counter() - t0)
        cpu_runs.append(cpu_s() - c0)
    order = sorted(range(5), key=lambda i: runs[i])
    dt, cpu_dt = runs[order[2]], cpu_runs[order[2]]
    worst = 0.0
    for i in range(clips):   # clip by clip: bounded temporaries
        peak = float(want[i].max())
        err = float(np.max(np.abs(gpu_out_host[i].astype(np.float64) - want[i])))
        worst = max(worst, err / peak)
    # comparators: the same restatement on ONE thread, and numpy's pocketfft (float64 frames x window ->
    # rfft -> |.|^2, one thread) -- each on a bounded sample of the same batch
    k1 = min(clips, 32)
    one = np.zeros((k1, BINS, O.frames(c, n)), dtype=np.float32)
    c_oracle.stft(c, x_host[:k1], 2.0, threads=1, out=one)
    r1, cpu1 = [], []
    for _ in range(3):
        c0, t0 = cpu_s(), time.perf_counter()
        c_oracle.stft(c, x_host[:k1], 2.0, threads=1, out=one)
        r1.append(time.perf_counter() - t0)
        cpu1.append(cpu_s() - c0)
    o1 = sorted(range(3), key=lambda i: r1[i])
    dt1, cpu_dt1 = r1[o1[1]], cpu1[o1[1]]
    win = np.asarray(c.analysis_window, dtype=np.float64)
    k2 = min(clips, 2)
    t0 = time.perf_counter()
    fr_np = 0
    for i in range(k2):
        xp = np.pad(x_host[i].astype(np.float64), (FFT // 2, FFT // 2), mode="reflect")
        idx = np.arange(0, xp.size - FFT + 1, HOP)[:, None] + np.arange(FFT)[None, :]
        spec = np.fft.rfft(xp[idx] * win[None, :], axis=1)
        pw = spec.real ** 2 + spec.imag ** 2
        fr_np += pw.shape[0]
    dtn = time.perf_counter() - t0
    v_all, v_one = frames / dt / 1e6, k1 * O.frames(c, n) / dt1 / 1e6
    speedup = v_all / v_one
    comparators = {"single_thread": {"value": round(v_one, 4), "unit": "Mframes/s", "cores": 1, "kind": "port",
                                     "sample": "%d clips, oracle/oracle_stft.c on one thread, median of 3 runs after a warm-up, %.2f s each" % (k1, dt1)},
                   "numpy_rfft": {"value": round(fr_np / dtn / 1e6, 4), "unit": "Mframes/s", "cores": 1, "kind": "comparator",
                                  "sample": "%d clips, numpy.fft.rfft (pocketfft, float64) of the windowed frames + |.|^2 on one thread, %.2f s" % (k2, dtn)}}
    return {"value": round(v_all, 4), "unit": "Mframes/s", "cores": cores, "kind": "port", "comparators": comparators,
            "sample": "the whole C2 batch: %d clips x %d samples (%d frames), oracle/oracle_stft.c f64 interior, "
                      "%d threads (os.cpu_count() %d), median of 5 runs after one warm-up run, %.2f s each (all five: %s)"
                      % (clips, n, frames, cores, os.cpu_count() or 1, dt, " ".join("%.2f" % v for v in sorted(runs))),
            # N threads cannot run more than N x one thread: the two rows and the CPU seconds each consumed per frame
            "consistency": {"threads_speedup": round(speedup, 2), "bound": round(1.2 * cores, 1), "within_bound": bool(speedup <= 1.2 * cores),
                            "cpu_us_per_frame_all_threads": round(cpu_dt / frames * 1e6, 2),
                            "cpu_us_per_frame_one_thread": round(cpu_dt1 / (k1 * O.frames(c, n)) * 1e6, 2)},
            "gpu_vs_oracle_max_err_over_peak": float("%.3g" % worst), "gpu_vs_oracle_frames_checked": frames,
            "gate": 1e-5}


def workload_name(workload, clips_arg, seconds_arg, world):
    """The name in config.workload.  For the default workload it does not depend on N: every line of a 1 / 2 / 4 / 8 GPU
    series names the same per-GPU batch (tests/test_sharding_gloo.py asserts that)."""
    if workload == "c2":
        return "C2 per GPU: %d clips x %.0f s mono 48 kHz fp32" % (clips_arg or 256, seconds_arg or 10.0)
    total = clips_arg or 4096
    return "C5: %d clips x %.0f s mono 48 kHz fp32 sharded by contiguous clip ranges (%d per GPU)" % (
        total, seconds_arg or 30.0, (total + world - 1) // world)


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.dry_run:   # CPU check of the launcher and of the rank environment (tests/test_sharding_gloo.py)
        from soundml_amd import shard
        if world > 1:
            dist.init_process_group("gloo")
        shard.barrier()
        t = shard.timed_region_max(1.0 + rank)
        lo, hi = shard.clip_range(args.clips or 4096, world, rank)
        wl = args.workload if args.workload != "auto" else "c2"
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "max_clock": t, "rank0_clips": [lo, hi],
                              "local_rank": local_rank, "master": os.environ.get("MASTER_ADDR"),
                              "workload": workload_name(wl, args.clips, args.seconds, world), "scaling": "weak" if wl == "c2" else "strong",
                              "world_size_seen": dist.get_world_size() if world > 1 else 1}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no HIP device visible (there is no CPU path)")
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    backend = "none"
    if world > 1:
        # chosen from the device count alone, identically on every rank (RCCL refuses two ranks on one device); a group that
        # fails to form fails the run -- re-initialising over a half-built store can leave ranks on different backends
        backend = os.environ.get("SMX_BENCH_BACKEND") or ("nccl" if ndev >= world else "gloo")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    red_dev = dev if backend == "nccl" else None

    import soundml_amd as S
    from soundml_amd import Fir, Mel, Stft
    from soundml_amd._lib import check, lib
    vp = ctypes.c_void_p

    stream = torch.cuda.current_stream(dev)
    sptr = vp(stream.cuda_stream)
    cfg = Stft.Config.create(fft_size=FFT, hop=HOP)          # Hann, centered, reflect (librosa defaults)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def make_clip_batch(lo, hi, n):
        """clips [lo, hi) of the job: uniform[-1,1), clip g seeded 42 + g, so any shard is regenerable anywhere"""
        x = torch.empty(hi - lo, n, device=dev, dtype=torch.float32)
        gen = torch.Generator(device=dev)
        for g in range(lo, hi):
            gen.manual_seed(42 + g)
            x[g - lo].uniform_(-1.0, 1.0, generator=gen)
        return x

    def timed(step, steps, warmup):
        """W warm-up steps, then K steps between barriers; HIP events on the launch stream around every step.
        Returns (elapsed seconds, MAX over ranks; sorted per-step ms of this rank; launches per step)."""
        preroll_steps = 0
        settled = None
        means = []
        if args.preroll_ms > 0:   # bring the chip to its sustained state (see --preroll-ms): ADAPTIVE -- groups of 8 steps, each
            # timed by HIP events, until --preroll-ms have passed AND the last 8 step times lie within 1 % of each other (at most 2 x --preroll-ms)
            t_pre = time.perf_counter()
            while True:
                evp = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
                for a, b in evp:
                    a.record(stream)
                    step()
                    b.record(stream)
                preroll_steps += 8
                torch.cuda.synchronize()
                last8 = [a.elapsed_time(b) for a, b in evp]
                spent = (time.perf_counter() - t_pre) * 1e3
                means.append(sum(last8) / 8.0)
                # settled: the last 8 steps within 1 % of each other AND their mean within 0.3 % of the mean 128 steps earlier -- the
                # first condition alone is met on the slow part of the decay too (one box: 184 steps, 0.481 ms; the same box run to
                # the cap: 0.457 ms -- profiles/r07/bench_n1_full.json against bench_n1.json)
                settled = (max(last8) - min(last8)) <= 0.01 * min(last8) and len(means) > 16 and abs(means[-1] - means[-17]) <= 0.003 * means[-1]
                # N > 1: every rank runs the full 2 x --preroll-ms (no early exit), so that the ranks reach the barrier together --
                # a rank that settled early would idle there while the others go on, and leave its sustained state again
                if (world == 1 and settled and spent >= args.preroll_ms) or spent >= 2.0 * args.preroll_ms:
                    break
        timed.settled = settled
        timed.preroll_steps = preroll_steps
        for _ in range(warmup):
            step()
        barrier()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        l0 = lib.smx_debug_kernel_launches()
        t0 = time.perf_counter()
        for a, b in ev:
            a.record(stream)
            step()
            b.record(stream)
        barrier()
        elapsed = time.perf_counter() - t0
        launches = (lib.smx_debug_kernel_launches() - l0) / max(steps, 1)
        elapsed = S.shard.timed_region_max(elapsed, device=red_dev)
        return elapsed, sorted(a.elapsed_time(b) for a, b in ev), launches

    def power_step(x, out, clips, n, frames):
        return lambda: check(lib.smx_stft_power_range_f32_dev(cfg._h, vp(x.data_ptr()), clips, n, n, 0, frames, 2.0,
                                                              vp(out.data_ptr()), sptr))

    workload = args.workload if args.workload != "auto" else "c2"   # the same workload at every N (see the module docstring)
    line = {"metric": "STFT Mframes/sec (n_fft=2048 hop=512)", "unit": "Mframes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic"}
    extra = {}

    # ---- the main workload ------------------------------------------------------------------------------
    if workload == "c2":
        clips = args.clips or 256
        n = int(round((args.seconds or 10.0) * SR))
        lo, hi = rank * clips, (rank + 1) * clips
        total_clips = clips * world
        scaling = "weak"
        wname = workload_name("c2", args.clips, args.seconds, world)
    else:
        total_clips = args.clips or 4096
        n = int(round((args.seconds or 30.0) * SR))
        lo, hi = S.shard.clip_range(total_clips, world, rank)
        clips = hi - lo
        scaling = "strong"
        wname = workload_name("c5", args.clips, args.seconds, world)
    frames = Stft.frames(cfg, n)
    x = make_clip_batch(lo, hi, n)
    out = torch.empty(clips, BINS, frames, device=dev, dtype=torch.float32)
    elapsed, step_ms, launches = timed(power_step(x, out, clips, n, frames), args.steps, args.warmup)
    main_preroll_steps = timed.preroll_steps
    main_settled = timed.settled
    # One step is ONE launch of the fused kernel (interior tiles, then the few border frames of every clip through
    # the same frame code), so the HIP events around a step are that kernel's launch durations: their average is
    # what `rocprofv3 --kernel-trace --stats` of this command reports for it (profiles/).
    if clips > 0 and launches != 1:
        raise SystemExit("bench.py: a step of the hot path issued %.2f kernel launches, expected 1 -- the step events "
                         "are no longer one kernel's duration" % launches)
    avg_ms = sum(step_ms) / len(step_ms)

    def over_ranks(v):
        """(min, max) of a per-rank scalar"""
        if world == 1:
            return v, v
        t = torch.tensor([v, -v], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t[0].item()), -float(t[1].item())
    rank_ms = over_ranks(avg_ms)
    # `ranks_counted`: an all-reduce SUM of 1 over the group -- the collective itself says how many ranks it spanned (the world size
    # alone is the launcher's environment).  Device ordinals summed the same way: 0 + 1 + ... + N - 1 when every rank has its own GPU.
    if world > 1:
        cnt = torch.tensor([1.0, float(local_rank)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        ranks_counted, ordinals_sum = int(cnt[0].item()), int(cnt[1].item())
    else:
        ranks_counted, ordinals_sum = 1, local_rank
    ranks_seen = {"world_size_seen": dist.get_world_size() if world > 1 else 1, "ranks_counted": ranks_counted,
                  "device_ordinals_sum": ordinals_sum, "backend": backend,
                  "rank_kernel_ms_avg_min": round(rank_ms[0], 4), "rank_kernel_ms_avg_max": round(rank_ms[1], 4)}

    shard_check = None
    if args.verify_shards:   # the reference's per-slice law (stft_grid.ml:180-205) across ranks, on the HIP path
        full = Stft.power_spectrum(cfg, make_clip_batch(0, total_clips, n))
        ok = bool(torch.equal(full[lo:hi], out))
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        shard_check = {"ranks_bit_exact": int(t.item()), "ranks": world, "world_size_seen": dist.get_world_size() if world > 1 else 1,
                       "backend": backend}
        del full

    if rank == 0:
        total_frames = total_clips * frames
        achieved = clips * frames * ALGO_BYTES_PER_FRAME / (avg_ms * 1e-3) / 1e9
        # `traffic` is NOT measured by this run: it is the PMC figure of the committed profile (separate --pmc passes of
        # rocprofv3, tools/profile_round.sh), quoted with the commit and kernel duration it was taken at
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if workload == "c2" and clips == 256 and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("bytes_per_launch")
                traffic_source = {k: tj.get(k) for k in ("source", "commit", "kernel", "kernel_us_in_profile", "box") if tj.get(k) is not None}
            except Exception:
                traffic = None
        line.update({
            "value": round(total_frames * args.steps / elapsed / 1e6, 3),
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "scaling": scaling,
            "config": {"workload": "%s (uniform[-1,1), clip g seeded 42+g), STFT n_fft=2048 hop=512 Hann centered/reflect "
                                   "power=2 -> [clips;1025;%d], device-resident in and out" % (wname, frames),
                       "frames_per_gpu": clips * frames, "sharding": "clips over ranks, no collective",
                       "timing_backend": backend,
                       # untimed, uncounted: the same step repeated before the W warm-up steps until the chip's power management has
                       # settled (--preroll-ms; `extra.c2_burst_from_idle` is the same measurement without it)
                       "preconditioning": {"min_ms": args.preroll_ms, "steps": main_preroll_steps, "settled_within_1pct": main_settled,
                                           "rule": "groups of 8 event-timed steps until the last 8 lie within 1 % of each other and their mean within 0.3 % of the mean 128 steps earlier (at least min_ms, at most 2 x min_ms; N > 1: always 2 x min_ms, so that the ranks meet the barrier together)"}},
            # every launch of the step that ran before the timed region: the preconditioning steps + the W declared warm-up steps
            "warmup_effective": main_preroll_steps + args.warmup,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "frac_of_measured": round(achieved / HBM_MEASURED_GBS, 4), "measured_peak": HBM_MEASURED_GBS,
                         "kernel": "stft2048_power32_kernel<true, 2, false, %d> (one launch per step: all %d frames of %d clips; "
                                   "the flush in whole aligned 64-byte blocks, %s)"
                                   % (1 if frames % 2 == 0 else 2, frames, clips, "a pair of frames per lane" if frames % 2 == 0 else "a frame per lane"),
                         "launches_per_step": launches,
                         "kernel_ms_avg": round(avg_ms, 4), "kernel_ms_median": round(step_ms[len(step_ms) // 2], 4),
                         "kernel_ms_min": round(step_ms[0], 4),
                         "algorithmic_bytes_per_frame": ALGO_BYTES_PER_FRAME},
        })
        line.update(ranks_seen)
        if shard_check:
            line["shard_check"] = shard_check

    # ---- CPU baseline on the same batch (rank 0, one GPU) ------------------------------------------------------
    if world == 1 and workload == "c2" and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(x.cpu().numpy(), out.cpu().numpy())

    # ---- extras: the other BASELINE configurations, same HIP-event method ---------------------------------------
    if not args.no_extras:
        k, w = max(5, min(args.steps, 20)), max(2, min(args.warmup, 5))
        if world == 1 and workload == "c2" and clips == 256:
            # The headline measurement WITHOUT its preconditioning: W warm-up steps and K timed ones right out of an idle device
            # (the CPU baseline above left it idle for seconds).  Reported so that the effect of --preroll-ms is on the record.
            pre = args.preroll_ms
            args.preroll_ms = 0.0
            try:
                el, ms, _ = timed(power_step(x, out, clips, n, frames), args.steps, args.warmup)
            finally:
                args.preroll_ms = pre
            extra["c2_burst_from_idle"] = {"workload": "the headline step, %d timed after %d warm-up steps, from an idle device (no preconditioning)" % (args.steps, args.warmup),
                                           "value": round(clips * frames * args.steps / el / 1e6, 1), "unit": "Mframes/s",
                                           "kernel_ms_avg": round(sum(ms) / len(ms), 4), "kernel_ms_min": round(ms[0], 4), "kernel_ms_max": round(ms[-1], 4)}
            # C3: fused mel spectrogram (128 mels) of the same batch
            mc = Mel.Config.create(n_mels=128, sample_rate=SR, fft_size=FFT)
            mout = torch.empty(clips, 128, frames, device=dev, dtype=torch.float32)
            _, ms, nl = timed(lambda: check(lib.smx_mel_spectrogram_f32_dev(cfg._h, mc._h, vp(x.data_ptr()), clips, n, n, 2.0,
                                                                            vp(mout.data_ptr()), sptr)), k, w)
            a = sum(ms) / len(ms)
            mfma = None
            ppath = os.path.join(ROOT, "profiles", "mfma_util.json")
            if os.path.exists(ppath):
                try:
                    mfma = json.load(open(ppath))
                except Exception:
                    mfma = None
            extra["c3_mel"] = {"workload": "C3: mel spectrogram (128 mels, Slaney) fused from audio on the C2 batch",
                               "value": round(clips * frames / a / 1e3, 1), "unit": "Mframes/s", "ms": round(a, 4),
                               "ms_min": round(ms[0], 4), "launches_per_step": nl,
                               "roofline": {"bound": "hbm", "achieved": round(clips * frames * MEL_BYTES_PER_FRAME / a / 1e6, 1),
                                            "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": round(clips * frames * MEL_BYTES_PER_FRAME / a / 1e6 / HBM_PEAK_GBS, 4),
                                            "algorithmic_bytes_per_frame": MEL_BYTES_PER_FRAME},
                               # What the MFMA pipe really does (committed PMC profile, profiles/mfma_util.json): pipe occupancy and
                               # the flops EXECUTED -- v_mfma_f32_4x4x1_16B_f32 blocks, every 4-mel group over its own band: 1/20 of
                               # the dense product's flops (round 3's 16 x 16 x 4 tiles executed 1/7 of it, at 2.5 times the pipe time).
                               "mfma": mfma,
                               # SURVEY 8(d)'s bound for reference only: the DENSE product (2 x 128 x 1025 flop per frame) priced
                               # against the fp32 MFMA peak.  Not a utilisation: 95 % of those flops are never executed.
                               "mfma_dense_equivalent": {"bound": "mfma", "achieved": round(2.0 * 128 * BINS * clips * frames / a / 1e9, 1),
                                                         "peak": 157.3, "unit": "TFLOP/s",
                                                         "frac": round(2.0 * 128 * BINS * clips * frames / a / 1e9 / 157.3, 4),
                                                         "algorithmic_flop_per_frame": 2 * 128 * BINS,
                                                         "note": "dense-equivalent rate, NOT utilisation (see mfma)"}}
            del mout
            # C4: FIR 8192 taps on 8 channels x 60 s
            h = Fir.design_lowpass(8192, 0.25, 100.0)
            plan = Fir.Plan.create(h)
            ch, ns = 8, 60 * SR
            xs = make_clip_batch(10000, 10000 + ch, ns)
            ys = torch.empty_like(xs)
            _, ms, nl = timed(lambda: check(lib.smx_fir_apply_f32_dev(plan._h, vp(xs.data_ptr()), ch, ns, ns, vp(ys.data_ptr()),
                                                                      ns, sptr)), k, w)
            a = sum(ms) / len(ms)
            extra["c4_fir"] = {"workload": "C4: 8192-tap Kaiser lowpass, overlap-save N=%d, 8 ch x 60 s 48 kHz" % plan.block,
                               "value": round(ch * ns / a / 1e6, 2), "unit": "Gsamples/s", "ms": round(a, 4),
                               "ms_min": round(ms[0], 4), "launches_per_step": nl,
                               "roofline": {"bound": "hbm", "achieved": round(ch * ns * FIR_BYTES_PER_SAMPLE / a / 1e6, 1),
                                            "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": round(ch * ns * FIR_BYTES_PER_SAMPLE / a / 1e6 / HBM_PEAK_GBS, 4),
                                            "algorithmic_bytes_per_sample": FIR_BYTES_PER_SAMPLE}}
            del xs, ys
            # C1's geometry (fft 1024 / hop 256, 441 000 samples) as a batch of 256 clips: the 16-lane form of the frame pipeline
            c1cfg = Stft.Config.create(fft_size=1024, hop=256)
            n1 = 441000
            f1 = Stft.frames(c1cfg, n1)
            x1 = make_clip_batch(20000, 20000 + 256, n1)
            o1 = torch.empty(256, 513, f1, device=dev, dtype=torch.float32)
            _, ms, nl = timed(lambda: check(lib.smx_stft_power_range_f32_dev(c1cfg._h, vp(x1.data_ptr()), 256, n1, n1, 0, f1, 2.0,
                                                                             vp(o1.data_ptr()), sptr)), k, w)
            a = sum(ms) / len(ms)
            b1 = 256 * 4 + 513 * 4   # hop x 4 read + 513 x 4 written per frame
            extra["c1_batch"] = {"workload": "C1's geometry as a batch: 256 clips x 441000 samples, fft 1024 / hop 256 (%d frames)" % (256 * f1),
                                 "value": round(256 * f1 / a / 1e3, 1), "unit": "Mframes/s", "ms": round(a, 4), "ms_min": round(ms[0], 4),
                                 "launches_per_step": nl,
                                 "roofline": {"bound": "hbm", "achieved": round(256 * f1 * b1 / a / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": round(256 * f1 * b1 / a / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_frame": b1}}
            # the same batch at fft 256 / hop 64: the 4-lane form (round 5; the Stockham kernel before it took 0.93 ms)
            c256 = Stft.Config.create(fft_size=256, hop=64)
            f256 = Stft.frames(c256, n1)
            o256 = torch.empty(256, 129, f256, device=dev, dtype=torch.float32)
            _, ms, nl = timed(lambda: check(lib.smx_stft_power_range_f32_dev(c256._h, vp(x1.data_ptr()), 256, n1, n1, 0, f256, 2.0,
                                                                             vp(o256.data_ptr()), sptr)), k, w)
            a = sum(ms) / len(ms)
            b256 = 64 * 4 + 129 * 4
            extra["fft256_batch"] = {"workload": "256 clips x 441000 samples, fft 256 / hop 64 (%d frames)" % (256 * f256),
                                     "value": round(256 * f256 / a / 1e3, 1), "unit": "Mframes/s", "ms": round(a, 4), "ms_min": round(ms[0], 4),
                                     "launches_per_step": nl,
                                     "roofline": {"bound": "hbm", "achieved": round(256 * f256 * b256 / a / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                  "frac": round(256 * f256 * b256 / a / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_frame": b256}}
            # ... and at fft 4096 / hop 1024 (the reference reaches this size from cqt.ml:648 and hpss.ml:490-492): a frame in a whole wave,
            # 8-frame single-buffered tiles (round 6: stft4096_power64_kernel; the Stockham kernel before it took 1.07 ms)
            c4k = Stft.Config.create(fft_size=4096, hop=1024)
            f4k = Stft.frames(c4k, n1)
            o4k = torch.empty(256, 2049, f4k, device=dev, dtype=torch.float32)
            _, ms, nl = timed(lambda: check(lib.smx_stft_power_range_f32_dev(c4k._h, vp(x1.data_ptr()), 256, n1, n1, 0, f4k, 2.0,
                                                                             vp(o4k.data_ptr()), sptr)), k, w)
            a = sum(ms) / len(ms)
            b4k = 1024 * 4 + 2049 * 4
            extra["fft4096_batch"] = {"workload": "256 clips x 441000 samples, fft 4096 / hop 1024 (%d frames)" % (256 * f4k),
                                      "value": round(256 * f4k / a / 1e3, 1), "unit": "Mframes/s", "ms": round(a, 4), "ms_min": round(ms[0], 4),
                                      "launches_per_step": nl,
                                      "roofline": {"bound": "hbm", "achieved": round(256 * f4k * b4k / a / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                   "frac": round(256 * f4k * b4k / a / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_frame": b4k}}
            del x1, o1, o256, o4k
            # Stft.transform of the C2 batch (complex64 out: 10 248 algorithmic bytes per frame)
            zc = torch.empty(clips, BINS, frames, 2, device=dev, dtype=torch.float32)
            _, ms, nl = timed(lambda: check(lib.smx_stft_transform_range_f32_dev(cfg._h, vp(x.data_ptr()), clips, n, n, 0, frames,
                                                                                 vp(zc.data_ptr()), sptr)), k, w)
            a = sum(ms) / len(ms)
            bc = HOP * 4 + BINS * 8
            extra["c2_complex"] = {"workload": "Stft.transform of the C2 batch (complex64 spectrogram)", "value": round(clips * frames / a / 1e3, 1),
                                   "unit": "Mframes/s", "ms": round(a, 4), "ms_min": round(ms[0], 4), "launches_per_step": nl,
                                   "roofline": {"bound": "hbm", "achieved": round(clips * frames * bc / a / 1e6, 1), "peak": HBM_PEAK_GBS,
                                                "unit": "GB/s", "frac": round(clips * frames * bc / a / 1e6 / HBM_PEAK_GBS, 4),
                                                "algorithmic_bytes_per_frame": bc}}
            # The face an nx caller gets (stft.mli:211-250: host tensors in and out): Stft.power_spectrum on a numpy batch through the
            # host-pointer entry point -- upload, kernels and download of clip units overlapped inside the C ABI (transfer.cpp); a fresh
            # result array per call, as nx allocates one -- since late round 5 from the library's page-locked pool (smx_host_alloc: what
            # the OCaml binding's result tensors use too; the block goes back to the pool when the array is dropped).  PCIe roof: 63 GB/s per direction; the bytes of both directions are counted.
            import numpy as _np
            # (an ordinary numpy array, as a host caller holds: a copy made by this thread.  The array that torch's x.cpu() returns reads
            # 1.7x slower from the library's copying threads -- 44 against 26 ms for the same call, tools/host_path_in_bench.py)
            xh = _np.array(x.cpu().numpy())
            hp = []
            for it_h in range(6):   # one untimed call first (it allocates the pool's block: ~90 ms once per process), then five timed
                t_h = time.perf_counter()
                ph = Stft.power_spectrum(cfg, xh)
                if it_h > 0:
                    hp.append(time.perf_counter() - t_h)
                ph_owndata = bool(ph.flags["OWNDATA"])
                if it_h == 0:
                    same = bool(_np.array_equal(ph[:4], out[:4].cpu().numpy()) and _np.array_equal(ph[-3:], out[-3:].cpu().numpy()))
                del ph     # (releasing a GB of touched pages is not the call: not timed)
            hp.sort()
            extra["c2_host_path"] = {"workload": "C2 through the host-pointer entry point (numpy in, fresh numpy out in a block of the page-locked result pool): Stft.power_spectrum; median of 5 calls after one untimed call",
                                     "result_page_locked": bool(not ph_owndata),
                                     "value": round(clips * frames / hp[2] / 1e6, 2), "unit": "Mframes/s", "ms": round(hp[2] * 1e3, 2), "ms_min": round(hp[0] * 1e3, 2),
                                     "ms_all_sorted": [round(v * 1e3, 1) for v in hp],
                                     "equals_device_resident_call": same,
                                     # PCIe is full duplex: the floor of this call is its LARGER direction at one direction's peak (the download, 0.98 GB / 63 GB/s =
                                     # 15.6 ms); measured bare on this link with the two directions overlapped these bytes take 18.9 ms (profiles/r07/pcie_duplex_probe.log)
                                     "roofline": {"bound": "pcie", "achieved": round(max(clips * n * 4, clips * BINS * frames * 4) / hp[2] / 1e9, 2), "peak": 63.0, "unit": "GB/s",
                                                  "frac": round(max(clips * n * 4, clips * BINS * frames * 4) / hp[2] / 1e9 / 63.0, 4),
                                                  "bytes_up": clips * n * 4, "bytes_down": clips * BINS * frames * 4,
                                                  "measured_duplex_floor_ms": 18.9, "frac_of_measured_duplex_floor": round(18.9 / (hp[2] * 1e3), 4),
                                                  "note": "achieved = the larger direction's bytes / the call's time, against one direction's 63 GB/s; "
                                                          "measured_duplex_floor_ms: the same bytes bare, both directions overlapped (tools/probes/pcie_duplex_probe.hip)"}}
            assert same, "host path != device-resident call"
            del xh
            # Stft.invert of that spectrogram (stft.ml:900-939): istft2048_pipe_kernel, 8200 B in + 2048 B out per frame
            yi = torch.empty(clips, n, device=dev, dtype=torch.float32)
            lib.smx_stft_invert_f32_dev.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, vp, vp]
            _, ms, nl = timed(lambda: check(lib.smx_stft_invert_f32_dev(cfg._h, vp(zc.data_ptr()), clips, BINS, frames, 1, n, vp(yi.data_ptr()), sptr)), k, w)
            a = sum(ms) / len(ms)
            bi = BINS * 8 + HOP * 4
            extra["c2_invert"] = {"workload": "Stft.invert of the C2 spectrogram (complex64 in, float32 audio out); round trip checked",
                                  "value": round(clips * frames / a / 1e3, 1), "unit": "Mframes/s", "ms": round(a, 4), "ms_min": round(ms[0], 4),
                                  "launches_per_step": nl, "round_trip_max_abs_err": float((yi - x).abs().max()),
                                  "roofline": {"bound": "hbm", "achieved": round(clips * frames * bi / a / 1e6, 1), "peak": HBM_PEAK_GBS,
                                               "unit": "GB/s", "frac": round(clips * frames * bi / a / 1e6 / HBM_PEAK_GBS, 4),
                                               "algorithmic_bytes_per_frame": bi}}
            assert extra["c2_invert"]["round_trip_max_abs_err"] < 1e-5, extra["c2_invert"]
            del zc, yi
            # C2 on the reference's own numerics (window, transform and |.|^2 in float64; float32 in and out): the same entry
            # point under set_interior("float64").  HBM bytes are C2's; what bounds this kernel is float64 vector issue
            # (~5 900 cycles per frame and wave at 4 cycles per instruction: DESIGN 4.3b), stated beside the HBM figure.
            S.set_interior("float64")
            try:
                _, ms, nl = timed(power_step(x, out, clips, n, frames), k, w)
            finally:
                S.set_interior("float32")
            a = sum(ms) / len(ms)
            extra["c2_float64_interior"] = {"workload": "C2 with the reference's float64 interior (set_interior float64): stft2048_power_wide_kernel",
                                            "value": round(clips * frames / a / 1e3, 1), "unit": "Mframes/s", "ms": round(a, 4), "ms_min": round(ms[0], 4),
                                            "launches_per_step": nl, "dtype": "f64",
                                            "roofline": {"bound": "hbm", "achieved": round(clips * frames * ALGO_BYTES_PER_FRAME / a / 1e6, 1),
                                                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                         "frac": round(clips * frames * ALGO_BYTES_PER_FRAME / a / 1e6 / HBM_PEAK_GBS, 4),
                                                         "algorithmic_bytes_per_frame": ALGO_BYTES_PER_FRAME,
                                                         "note": "issue-bound, not HBM-bound: 16 frames x 5900 cycles of float64 vector instructions per SIMD quartet and tile"}}
            # Stft.griffin_lim (stft.ml:961-1017), 32 iterations at the reference's defaults on the C2 magnitudes: 33 syntheses + 32 analyses, the
            # rebuilt spectra frame-major inside the library (round 5).  Per frame and iteration: c_k, c_(k-1) and the magnitudes read, the signal
            # written and read, c_(k+1) written = 8200 x 3 + 4100 + 2048 x 2 bytes.
            out.copy_(torch.rand_like(out))          # magnitudes in [0, 1)
            yg = torch.empty(clips, n, device=dev, dtype=torch.float32)
            lib.smx_stft_griffin_lim_f32_dev.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_double, vp, ctypes.c_int,
                                                         ctypes.c_int64, vp, vp]
            gl_ms = []
            for it_g in range(4):                    # one untimed call (its scratch comes out of the pool), then three timed
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record(torch.cuda.current_stream())
                check(lib.smx_stft_griffin_lim_f32_dev(cfg._h, vp(out.data_ptr()), clips, BINS, frames, 32, 0.99, None, 1, n, vp(yg.data_ptr()), sptr))
                eb.record(torch.cuda.current_stream())
                torch.cuda.synchronize()
                if it_g:
                    gl_ms.append(ea.elapsed_time(eb))
            gl_ms.sort()
            bg = 8200 * 3 + 4100 + 2048 * 2
            extra["c2_griffin_lim"] = {"workload": "Stft.griffin_lim, 32 iterations, momentum 0.99, on 256 x 1025 x 938 magnitudes -> 256 x 480000 samples",
                                       "value": round(clips * frames * 32 / gl_ms[1] / 1e3, 1), "unit": "Mframes/s (frame-iterations)", "ms": round(gl_ms[1], 2),
                                       "ms_per_iteration": round(gl_ms[1] / 32, 4), "finite": bool(torch.isfinite(yg).all()),
                                       "roofline": {"bound": "hbm", "achieved": round(clips * frames * 32 * bg / gl_ms[1] / 1e6, 1), "peak": HBM_PEAK_GBS,
                                                    "unit": "GB/s", "frac": round(clips * frames * 32 * bg / gl_ms[1] / 1e6 / HBM_PEAK_GBS, 4),
                                                    "algorithmic_bytes_per_frame_and_iteration": bg}}
            del yg
            # C5 on ONE GPU: the N = 1 point of BASELINE configs[4] (71 GB resident)
            free_b, _ = torch.cuda.mem_get_info(dev)
            del x, out
            c5_clips, n5 = 4096, 30 * SR
            f5 = Stft.frames(cfg, n5)
            need = c5_clips * (n5 + BINS * f5) * 4
            if free_b > need + (8 << 30):
                x5 = make_clip_batch(0, c5_clips, n5)
                o5 = torch.empty(c5_clips, BINS, f5, device=dev, dtype=torch.float32)
                _, ms, nl = timed(power_step(x5, o5, c5_clips, n5, f5), 5, 2)
                a = sum(ms) / len(ms)
                extra["c5_one_gpu"] = {"workload": "C5 on one GPU: 4096 clips x 30 s, %d frames, one call" % (c5_clips * f5),
                                       "value": round(c5_clips * f5 / a / 1e3, 1), "unit": "Mframes/s", "ms": round(a, 3),
                                       "launches_per_step": nl,
                                       "roofline": {"bound": "hbm", "achieved": round(c5_clips * f5 * ALGO_BYTES_PER_FRAME / a / 1e6, 1),
                                                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                    "frac": round(c5_clips * f5 * ALGO_BYTES_PER_FRAME / a / 1e6 / HBM_PEAK_GBS, 4)}}
                del x5, o5
        elif world > 1 and workload == "c2" and not args.clips:
            # BASELINE configs[4], strong scaling: 4096 clips x 30 s sharded by contiguous clip ranges over the same ranks
            del x, out
            c5_clips, n5 = 4096, 30 * SR
            f5 = Stft.frames(cfg, n5)
            lo5, hi5 = S.shard.clip_range(c5_clips, world, rank)
            x5 = make_clip_batch(lo5, hi5, n5)
            o5 = torch.empty(hi5 - lo5, BINS, f5, device=dev, dtype=torch.float32)
            el, ms, nl = timed(power_step(x5, o5, hi5 - lo5, n5, f5), k, w)
            a = sum(ms) / len(ms)
            r5 = over_ranks(a)
            extra["c5_strong"] = {"workload": workload_name("c5", 0, 0.0, world),
                                  "value": round(c5_clips * f5 * k / el / 1e6, 1), "unit": "Mframes/s", "scaling": "strong",
                                  "ms_per_step": round(el / k * 1e3, 4), "launches_per_step": nl,
                                  "rank_kernel_ms_avg_min": round(r5[0], 4), "rank_kernel_ms_avg_max": round(r5[1], 4),
                                  "n1_point": "extra.c5_one_gpu of the N = 1 line"}
            del x5, o5

    # ---- the drop-in's own multi-GPU face: ONE process, host tensors in and out, every visible device (smx_set_devices) -----------------
    # The reference's caller is one OCaml process handing over nx (host) tensors, a batch of clips per call (stft.mli:211-250); with a
    # device list the C ABI cuts `lead` into contiguous clip ranges, one host thread + staging ring pair + PCIe link per device, every
    # range writing its slice of the one result.  C5-shaped: 32 clips x 30 s per listed device, PCIe inclusive,
    # a fresh result per call (from the page-locked pool).  N > 1: rank 0 drives all N devices while the other ranks wait at a barrier.
    if not args.no_extras and workload == "c2" and not args.clips:
        barrier()
        if rank == 0:
            try:   # (an extra: a failure on a device set this code has never met must not take the headline line with it -- it is reported in the row)
                import numpy as _np
                per, n5 = 32, 30 * SR
                devices = list(range(ndev if world == 1 else min(ndev, world)))
                c5h, f5 = per * len(devices), Stft.frames(cfg, n5)
                xh5 = _np.empty((c5h, n5), _np.float32)
                edge = {}
                for d in range(len(devices)):
                    xb = make_clip_batch(50000 + d * per, 50000 + (d + 1) * per, n5)
                    xh5[d * per:(d + 1) * per] = xb.cpu().numpy()
                    if d == 0:
                        edge["first"] = Stft.power_spectrum(cfg, xb[:2]).cpu().numpy()    # the device-resident call on rank 0's device
                    if d == len(devices) - 1:
                        edge["last"] = Stft.power_spectrum(cfg, xb[-2:]).cpu().numpy()
                    del xb

                def calls(devs, count):
                    S.set_devices(devs)
                    try:
                        ts, same = [], None
                        for it_h in range(count + 1):   # one untimed call first (page-locked blocks, staging rings, per-device tables), then `count` timed
                            t_h = time.perf_counter()
                            ph = Stft.power_spectrum(cfg, xh5)
                            if it_h:
                                ts.append(time.perf_counter() - t_h)
                            else:
                                same = bool(_np.array_equal(ph[:2], edge["first"]) and _np.array_equal(ph[-2:], edge["last"]))
                            del ph
                        return sorted(ts), same
                    finally:
                        S.set_devices([])
                ts, same = calls(devices, 5)
                up_b, down_b = c5h * n5 * 4, c5h * BINS * f5 * 4
                row = {"workload": "one process, one Stft.power_spectrum call on a host batch of %d clips x 30 s (32 per device) sharded by smx_set_devices over devices %s: "
                                   "upload, kernels and download of every shard inside the call; median of 5 calls after one untimed call" % (c5h, devices),
                       "devices": devices, "value": round(c5h * f5 / ts[2] / 1e6, 2), "unit": "Mframes/s", "ms": round(ts[2] * 1e3, 2),
                       "ms_all_sorted": [round(v * 1e3, 1) for v in ts], "equals_device_resident_call": same,
                       "roofline": {"bound": "pcie", "achieved": round(max(up_b, down_b) / ts[2] / 1e9, 2), "peak": 63.0 * len(devices), "unit": "GB/s",
                                    "frac": round(max(up_b, down_b) / ts[2] / 1e9 / (63.0 * len(devices)), 4), "bytes_up": up_b, "bytes_down": down_b,
                                    "note": "the larger direction's bytes / the call's time against 63 GB/s per device (each MI355X has its own Gen5 x16 link)"}}
                if len(devices) > 1:   # the same batch through ONE device: what the sharding buys
                    t1, same1 = calls([devices[0]], 2)
                    row["one_device_ms"] = round(t1[len(t1) // 2] * 1e3, 2)
                    row["speedup_over_one_device"] = round(t1[len(t1) // 2] / ts[2], 2)
                    same = same and same1
                if not same:   # (loud, but the line survives: `equals_device_resident_call` false + this)
                    row["error"] = "sharded host call != device-resident call"
                    print("bench.py: c5_host_sharded: " + row["error"], file=sys.stderr)
                extra["c5_host_sharded"] = row
                del xh5
            except Exception as e_sh:   # noqa: BLE001
                S.set_devices([])
                extra["c5_host_sharded"] = {"error": "%s: %s" % (type(e_sh).__name__, e_sh)}
        barrier()

    # ---- what the board does under the headline kernel (rank 0, one GPU; outside every timed region) --------------------------------
    # LAST of all (a first version ran it before the CPU baseline, and the measurement out of an idle device that follows the baseline then
    # took 65 ms of wall clock for its 50 steps instead of 26: whatever rocm-smi's query leaves behind, nothing is measured after it now).
    # rocm-smi sampled four times while the step runs back to back for ~2.5 s: package power and shader clock.  The kernel sits at the
    # board's power cap (profiles/r07/power_clock_sample.log: 1381-1394 W of 1400, sclk ~2240 of 2400 MHz), and how far the cap pulls
    # the clock down differs by box -- this puts the box's own figures beside its line, and the ENERGY of a launch (W x the kernel's ms:
    # at the cap time follows joules per frame, so a change to the kernel is judged by this figure too).  None where rocm-smi does not answer.
    # rocm-smi is a `#!/usr/bin/env python3` script: it is started with the profiler's variables scrubbed (under rocprofv3 this
    # process carries LD_PRELOAD / ROCP_* / HSA_TOOLS_*, and a child that inherits them would be an exec hop with the profiler's library
    # loaded), and not at all when a counter collection is active.
    if world == 1 and workload == "c2" and rank == 0:
        def power_sample():
            import subprocess, threading
            if os.environ.get("ROCPROF_COUNTER_COLLECTION") or os.environ.get("ROCPROF_PMC"):   # (rocprofv3 --pmc: no child processes at all)
                return None
            child_env = {k: v for k, v in os.environ.items()
                         if k != "LD_PRELOAD" and not k.startswith(("ROCP_", "ROCPROF", "HSA_TOOLS_", "ROCTX", "KOKKOS_TOOLS"))}
            got = []
            def smp():
                time.sleep(1.0)   # (the package power rocm-smi reports lags the load by a few hundred ms: 1154 W at 0.3 s, 1380-1390 from ~0.6 s on)
                for _ in range(4):
                    try:
                        time.sleep(0.25)
                        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5, env=child_env)
                        rows = [ln.split(",") for ln in r.stdout.strip().splitlines() if "," in ln]
                        if len(rows) >= 2:
                            hdr, val = rows[0], rows[1 + min(local_rank, len(rows) - 2)]
                            col = lambda key: next((val[i] for i, hname in enumerate(hdr) if key in hname.lower()), None)
                            pw, sc = col("power (w)"), col("sclk clock speed")
                            got.append({"package_w": float(pw) if pw not in (None, "", "N/A") else None,
                                        "sclk_mhz": int("".join(ch for ch in sc if ch.isdigit())) if sc else None})
                    except Exception:
                        pass
            xb = make_clip_batch(lo, hi, n)
            ob = torch.empty(clips, BINS, frames, device=dev, dtype=torch.float32)
            step = power_step(xb, ob, clips, n, frames)
            th = threading.Thread(target=smp)
            th.start()
            evs = []
            while th.is_alive():
                a_e, b_e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a_e.record(stream)
                for _ in range(20):
                    step()
                b_e.record(stream)
                torch.cuda.synchronize(dev)
                evs.append(a_e.elapsed_time(b_e) / 20.0)
            th.join()
            got = [g for g in got if g.get("package_w") is not None]
            if not got:
                return None
            ps = dict(got[-1])   # the LAST sample: the one furthest into the load (the reported power is a moving average)
            ps["samples_w"] = [g["package_w"] for g in got]
            tail = sorted(evs[len(evs) // 2:])   # the launch time while the samples were taken (second half of the load)
            ps["kernel_ms_under_sampling"] = round(tail[len(tail) // 2], 4)
            ps["joules_per_launch"] = round(ps["package_w"] * tail[len(tail) // 2] * 1e-3, 4)
            ps["nanojoules_per_frame"] = round(ps["package_w"] * tail[len(tail) // 2] * 1e-3 / (clips * frames) * 1e9, 1)
            return ps
        try:
            ps = power_sample()
        except Exception:
            ps = None
        if ps:
            ps["note"] = "rocm-smi while the step runs back to back (outside the timed region); the board's cap is 1400 W and the clock's ceiling 2400 MHz; joules_per_launch = package W x the launch's ms under the same load"
        line["roofline"]["board_under_the_kernel"] = ps

    if rank == 0:
        if extra:
            line["extra"] = extra
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

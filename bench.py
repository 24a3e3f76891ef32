#!/usr/bin/env python3
"""Benchmark of the hot path: STFT power spectrogram, n_fft=2048 hop=512 Hann,
float32, on synthetic device-resident audio (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

One process per GPU.  A step = one pass of Stft.power_spectrum (through the C
ABI, device pointers) over this rank's resident batch: BASELINE config C2,
256 clips x 10 s x 48 kHz per GPU (weak scaling: clips shard embarrassingly, no
collective on the data path; the only communication is the timing barrier).
Rank 0 prints ONE JSON line with the whole-job Mframes/s, the HBM roofline of
the dominant kernel (HIP events on the launch stream) and, at N=1, a CPU
baseline (the oracle's C restatement, float64 interior, all host cores) on a
bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_FRAME = 512 * 4 + 1025 * 4      # hop*4 read + bins*4 written = 6148 B
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(fft, hop, n, target_seconds=12.0):
    """Times the oracle's C port (float64 interior, clip-parallel pthreads) on a
    bounded sample of the workload: as many 10 s clips as fit ~target_seconds."""
    import numpy as np
    from oracle import c_oracle, soundml_oracle as O
    cores = os.cpu_count() or 1
    c = O.stft_config(fft, hop=hop)
    rng = np.random.default_rng(42)
    probe = rng.uniform(-1, 1, size=(cores, n)).astype(np.float32)
    t0 = time.perf_counter()
    c_oracle.stft(c, probe, 2.0, threads=cores)
    dt = time.perf_counter() - t0
    rounds = max(1, min(64, int(target_seconds / max(dt, 1e-3))))
    clips = cores * rounds
    x = rng.uniform(-1, 1, size=(clips, n)).astype(np.float32)
    t0 = time.perf_counter()
    c_oracle.stft(c, x, 2.0, threads=cores)
    dt = time.perf_counter() - t0
    frames = clips * O.frames(c, n)
    return {"value": round(frames / dt / 1e6, 4), "unit": "Mframes/s", "cores": cores, "kind": "port",
            "sample": "%d clips x %d samples (%d frames) of the C2 workload, oracle/oracle_stft.c f64 "
                      "interior, %d threads, %.1f s" % (clips, n, frames, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--clips", type=int, default=256, help="clips per GPU (C2: 256)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    import soundml_amd as S
    from soundml_amd import Stft
    from soundml_amd._lib import check, lib

    fft, hop, sr = 2048, 512, 48000
    n = int(round(args.seconds * sr))
    clips = args.clips
    cfg = Stft.Config.create(fft_size=fft, hop=hop)          # Hann, centered, reflect (librosa defaults)
    frames = Stft.frames(cfg, n)
    # synthetic audio: uniform[-1,1), seeded per rank so any shard is regenerable on-device
    gen = torch.Generator(device=dev)
    gen.manual_seed(42 + rank)
    x = torch.rand(clips, n, device=dev, generator=gen, dtype=torch.float32) * 2 - 1
    out = torch.empty(clips, cfg.bins, frames, device=dev, dtype=torch.float32)
    stream = torch.cuda.current_stream(dev)
    sptr = ctypes.c_void_p(stream.cuda_stream)

    def step():
        check(lib.smx_stft_power_range_f32_dev(cfg._h, ctypes.c_void_p(x.data_ptr()), clips, n, n, 0, frames,
                                               2.0, ctypes.c_void_p(out.data_ptr()), sptr))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    elapsed = S.shard.timed_region_max(elapsed, device=dev)
    step_ms = sorted(a.elapsed_time(b) for a, b in ev)
    avg_step_ms = sum(step_ms) / len(step_ms)
    # One step is ONE launch of the fused kernel (stft2048_power_kernel<true, true, false>: the interior
    # tiles, then the few border frames of every clip through the same frame code), so the HIP events recorded
    # on the launch stream around each step of the timed region are that kernel's launch durations: their
    # average is what `rocprofv3 --kernel-trace --stats` of this command reports for it (profiles/).
    kernel_ms = step_ms
    avg_kernel_ms = avg_step_ms

    if rank == 0:
        total_frames = clips * frames * world
        value = total_frames * args.steps / elapsed / 1e6
        achieved = clips * frames * ALGO_BYTES_PER_FRAME / (avg_kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "STFT Mframes/sec (n_fft=2048 hop=512)", "value": round(value, 3), "unit": "Mframes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2 per GPU: %d clips x %.0f s mono 48 kHz fp32 (uniform[-1,1) seed 42+rank), "
                                   "STFT n_fft=2048 hop=512 Hann centered/reflect power=2 -> [clips;1025;%d], "
                                   "device-resident in and out" % (clips, args.seconds, frames),
                       "frames_per_gpu": clips * frames, "sharding": "clips over ranks, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "stft2048_power_kernel<true, true, false> (one launch per step: all %d frames of %d clips)" % (frames, clips),
                         "kernel_ms_avg": round(avg_kernel_ms, 4), "kernel_ms_min": round(kernel_ms[0], 4),
                         "algorithmic_bytes_per_frame": ALGO_BYTES_PER_FRAME},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(fft, hop, n)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

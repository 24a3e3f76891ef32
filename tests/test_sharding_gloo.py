"""The N>1 path on CPU: two processes over gloo (the production backend is RCCL = "nccl").
Clips shard embarrassingly, so what must hold is (a) the partition covers every
clip exactly once, (b) a rank's shard result equals the same slice of the full batch exactly
(the reference's per-slice law, stft_grid.ml:180-205; checked here on the oracle because the
HIP path needs a GPU), (c) the timing reduction is the MAX over ranks, (d) the optional host
gather reassembles the batch in clip order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from soundml_amd import shard


def test_clip_range_partitions():
    for total in (0, 1, 2, 7, 256, 4096, 4097):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = shard.clip_range(total, world, r)
                assert 0 <= lo <= hi <= total
                seen.extend(range(lo, hi))
                assert hi - lo in (total // world, total // world + 1)
            assert seen == list(range(total))
    assert shard.clip_range(4096, 8, 3) == (1536, 2048)      # BASELINE C5: 512 clips per GPU
    with pytest.raises(ValueError):
        shard.clip_range(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import soundml_oracle as O
        rng = np.random.default_rng(123)                  # every rank can regenerate the whole batch
        x = rng.uniform(-1, 1, size=(5, 3000)).astype(np.float32)
        c = O.stft_config(256, hop=64)
        lo, hi = shard.clip_range(x.shape[0], world, rank)
        local = O.power_spectrum(c, x[lo:hi])             # this rank's shard, no communication
        full = O.power_spectrum(c, x)
        assert np.array_equal(local, full[lo:hi])         # shard == slice of the batch, exactly
        shard.barrier()
        t = shard.timed_region_max(0.25 + rank)           # MAX over ranks
        assert t == 0.25 + (world - 1)
        gathered = shard.gather_host(torch.from_numpy(local))
        assert np.array_equal(gathered.numpy(), full)
        open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(world))


def test_bench_launcher_starts_the_ranks(tmp_path):
    """`python bench.py --gpus 2` (no RANK in the environment: the driver's command) must itself start two rank
    processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set and a 127.0.0.1 rendezvous, forward rank 0's one
    JSON line and return the worst child status.  `--dry-run` stops each rank before any GPU call (gloo barrier and
    MAX-reduce only), so the plumbing is covered on CPU; the product path of two ranks runs in
    tests/test_gpu_baseline_configs.py::test_bench_two_ranks_one_box."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True,
                         text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["max_clock"] == 2.0 and line["rank0_clips"] == [0, 2048]
    assert line["master"] == "127.0.0.1"
    # `value` names the SAME workload at every N (a scaling efficiency divides like by like), and the line says what the
    # ranks saw
    assert line["world_size_seen"] == 2 and line["scaling"] == "weak"
    names = {line["workload"]}
    for n in (1, 8):
        o = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--dry-run"], capture_output=True,
                           text=True, timeout=600, env=env)
        assert o.returncode == 0, o.stdout + o.stderr
        ln = json.loads(o.stdout.strip().splitlines()[-1])
        assert ln["n_gpus"] == n and ln["world_size_seen"] == n
        names.add(ln["workload"])
    assert len(names) == 1, names
    # a failing rank is a failing run (here: no HIP device in this container -> every rank exits non-zero)
    import torch
    if not torch.cuda.is_available():
        bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                             text=True, timeout=600, env=env)
        assert bad.returncode != 0


def test_the_c_abi_splits_clips_by_the_same_rule():
    """smx_set_devices shards the host-pointer batch calls inside the C ABI (capi.cpp for_each_shard); its split is
    smx_shard_clip_range, which must be shard.clip_range -- the rule bench.py's ranks use -- for every (clips, shards, shard):
    contiguous, balanced, the first clips % shards shards one clip more, every clip owned exactly once.  No device needed."""
    import ctypes
    from soundml_amd import _lib, shard
    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    for total in (0, 1, 2, 7, 8, 255, 256, 257, 4096, 4099):
        for world in (1, 2, 3, 4, 8, 16):
            covered = 0
            for r in range(world):
                _lib.check(_lib.lib.smx_shard_clip_range(total, world, r, ctypes.byref(lo), ctypes.byref(hi)))
                assert (lo.value, hi.value) == shard.clip_range(total, world, r), (total, world, r)
                assert lo.value == covered
                covered = hi.value
            assert covered == total
    assert _lib.lib.smx_shard_clip_range(10, 4, 4, ctypes.byref(lo), ctypes.byref(hi)) == 2
    assert b"outside a list of 4" in _lib.lib.smx_last_error()
    assert _lib.lib.smx_shard_clip_range(10, 0, 0, ctypes.byref(lo), ctypes.byref(hi)) == 2

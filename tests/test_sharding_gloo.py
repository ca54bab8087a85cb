"""CPU tier: the N > 1 path (env shards + one observation all_gather per step) with world_size 2 on gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mujoco_jaco_amd.sharding import ObsGather, env_seed, shard_range


def test_shard_ranges_cover_everything():
    for total, world in ((524288, 8), (65536, 3), (10, 4), (7, 8)):
        spans = [shard_range(r, world, total) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    assert len({env_seed(7, r) for r in range(8)}) == 8


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total, width = 64, 26
    lo, hi = shard_range(rank, world, total)
    g = ObsGather(hi - lo, width, torch.device("cpu"))
    for step in range(3):
        # each rank "steps" its own shard: rows are a pure function of (global env id, step)
        ids = torch.arange(lo, hi, dtype=torch.float32)[:, None]
        local = ids * 100 + step + torch.arange(width, dtype=torch.float32)[None, :] * 0.01
        if step == 1:                       # the overlapped form (side stream on a GPU; plain call on CPU): same result
            g.start(local); full = g.wait()
        else:
            full = g(local)
        ref = torch.arange(total, dtype=torch.float32)[:, None] * 100 + step + torch.arange(width, dtype=torch.float32)[None, :] * 0.01
        assert torch.equal(full, ref), (rank, step)
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out[rank] = float(t.item())
    dist.barrier()
    dist.destroy_process_group()


def test_obs_gather_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert dict(out) == {0: 2.0, 1: 2.0}


def test_bench_self_launch_dry_gather():
    """`bench.py --gpus 2` outside a launcher spawns its own ranks (before any GPU call) and aggregates: rehearsed on gloo / CPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-gather", "--steps", "3"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["gather_ok"] and line["steps"] == 3
    assert 0 < line["per_rank"]["ms_per_step_min"] <= line["per_rank"]["ms_per_step_max"]

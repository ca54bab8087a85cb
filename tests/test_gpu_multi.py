"""GPU tier, N > 1: `bench.py --gpus 2` (its own rank launch, one process per GPU, RCCL all_gather of the observation rows on a side
stream).  Collected everywhere; runs only where two HIP devices are visible (the round-end 8-GPU node), skips cleanly elsewhere."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_gpus_rccl_gather():
    if torch.cuda.device_count() < 2:   # (device_count does not initialise the GPU: the ranks are started from a process that never has)
        pytest.skip("needs two HIP devices")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--preroll", "4", "--batch", "8192"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["value"] > 0
    c = line["config"]["collective"]
    assert c["backend"] == "nccl" and c["world_size"] == 2 and c["devices_visible"] >= 2
    assert line["config"]["error_flags_or"] & 15 == 0


def _overlap_worker(rank, world, port, out):
    import torch.distributed as dist
    from mujoco_jaco_amd.sharding import ObsGather
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    g = ObsGather(256, 26, dev)
    ok = g._overlap
    for step in range(4):
        local = torch.full((256, 26), float(10 * step + rank), device=dev)
        g.start(local)
        local.fill_(-1.0)                       # the rows may be overwritten right away: the gather reads its staging copy
        full = g.wait()
        torch.cuda.current_stream().synchronize()
        ok = ok and bool((full.view(world, 256, 26)[:, 0, 0].cpu() == torch.tensor([10.0 * step + r for r in range(world)])).all())
    out[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_gather_two_ranks_on_one_gpu():
    """The side-stream form of the observation gather (sharding.ObsGather.start / wait) with two ranks sharing the one GPU of this box
    over gloo (RCCL refuses two ranks on one device): staging copy, event ordering and double use of the staging buffer."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    try:
        mp.spawn(_overlap_worker, args=(2, port, out), nprocs=2, join=True)
    except Exception as e:   # a gloo build without device-tensor collectives
        pytest.skip("gloo cannot gather device tensors here: %s" % (str(e).splitlines()[-1][:120],))
    assert dict(out) == {0: True, 1: True}

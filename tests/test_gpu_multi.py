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

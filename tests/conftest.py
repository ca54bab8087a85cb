import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a HIP device and the built extension: skip (not fail) them on a CPU-only box."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    have = have and os.path.exists(os.path.join(ROOT, "mujoco_jaco_amd", "libjaco_env.so"))
    if have:
        return
    skip = pytest.mark.skip(reason="needs a HIP device and mujoco_jaco_amd/libjaco_env.so")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def model_arrays():
    from mujoco_jaco_amd.modelc import blob
    return blob.load(os.path.join(ROOT, "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.jacomdl"))


@pytest.fixture(scope="session")
def names():
    d = {}
    for line in open(os.path.join(ROOT, "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.names.txt")):
        k, v = line.strip().split(": ", 1)
        d[k] = v.split()
    return d

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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

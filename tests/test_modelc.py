"""CPU tier: the model compiler's MJCF reader (mujoco_jaco_amd/modelc/mjcf.py) on the on-disk format either side of the hot path."""
import os

import numpy as np
import pytest

from mujoco_jaco_amd.modelc import mjcf

REF_ASSETS = "/root/reference/env_script/assets/jaco2"


def test_numeric_attributes_are_read_like_mujoco_reads_them():
    """Stream extraction: each number is the longest valid prefix; white space between numbers is optional.  The reference's own
    jaco2_torque.xml:47 has size=".01 .02.035" (three numbers for MuJoCo)."""
    assert mjcf._numbers(".01 .02.035") == [0.01, 0.02, 0.035]
    assert mjcf._numbers("1e-3 -2.5E+2 3") == [0.001, -250.0, 3.0]
    assert mjcf._numbers("0 -1.9 -1.1") == [0.0, -1.9, -1.1]
    assert mjcf._numbers("  1.  2  ") == [1.0, 2.0]
    assert mjcf._numbers("-.5-.5") == [-0.5, -0.5]
    with pytest.raises(ValueError):
        mjcf._numbers("1 x 2")
    assert np.array_equal(mjcf._floats(".3", n=3, default=(1, 2, 3)), [0.3, 2, 3])


@pytest.mark.skipif(not os.path.isdir(REF_ASSETS), reason="reads the reference's model data (build container only)")
def test_sibling_mjcfs_parse():
    """jaco2_torque.xml (12 hinge joints incl. the sprung distal finger joints, :109-133) and the old curtain model parse now; the
    default model and the arm-only model keep the sizes SURVEY.md section 8a lists."""
    sizes = {}
    for name in ("jaco2_curtain_torque", "jaco2_reaching_torque", "jaco2_torque", "jaco2_curtain_torque_old", "jaco2_sensor_torque"):
        M, names = mjcf.parse(os.path.join(REF_ASSETS, name + ".xml"), timestep=0.001)
        sizes[name] = (int(M["nq"][0]), int(M["nv"][0]), int(M["nu"][0]), int(M["njnt"][0]))
    assert sizes["jaco2_curtain_torque"] == (23, 21, 9, 11) and sizes["jaco2_reaching_torque"] == (9, 9, 9, 9)
    assert sizes["jaco2_torque"] == (12, 12, 9, 12) and sizes["jaco2_curtain_torque_old"][3] == 13

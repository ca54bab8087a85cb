"""CPU tier: the C-ABI library loads and exports every symbol include/jaco_env.h declares; no compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "jaco_env.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(jaco_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from mujoco_jaco_amd import _lib
    names = _declared()
    assert len(names) >= 15
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "libjaco_env.so does not export %s" % n
    assert sorted(_lib.SYMBOLS) == names, "ctypes table and header disagree"
    _lib.load()
    _lib.load("_d12")          # the build for the layout of jaco2_torque.xml exports the same ABI
    _lib.load("_d30")          # ... and the build for the two-arm layout of jaco2_dual_torque.xml
    assert _lib.variant_for(open(_lib.model_path("jaco2_torque"), "rb").read()) == "_d12"
    assert _lib.variant_for(open(_lib.model_path("jaco2_dual_torque"), "rb").read()) == "_d30"
    assert _lib.variant_for(open(_lib.model_path("jaco2_curtain_torque"), "rb").read()) == "" and _lib.variant_for(open(_lib.model_path("jaco2_reaching_torque"), "rb").read()) == ""


def test_header_cites_reference_call_sites():
    src = open(os.path.join(ROOT, "include", "jaco_env.h")).read()
    for cite in ("mujoco.py:258-278", "mujoco.py:213-246", "env_mujoco_util.py:470-475", "mujoco_config.py:74"):
        assert cite in src


def test_no_cpu_fallback():
    """Without a HIP device the product must refuse to run rather than fall back to any CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mujoco_jaco_amd import _lib
    from mujoco_jaco_amd.physics import BatchedMujoco, JacoError
    with pytest.raises(JacoError):
        BatchedMujoco(4)
    L = _lib.load()
    blob = open(_lib.model_path("jaco2_curtain_torque"), "rb").read()
    buf = ctypes.create_string_buffer(blob, len(blob))
    cfg = _lib.JacoConfig(ctypes.cast(buf, ctypes.c_void_p), len(blob), 4, 0, 50, 0, 0)
    h = ctypes.c_void_p()
    assert L.jaco_create(ctypes.byref(cfg), ctypes.byref(h)) == -3  # JACO_ENODEV
    assert b"no usable HIP device" in L.jaco_last_error(None)


def test_product_does_not_touch_the_oracle():
    """oracle/ and tests/emu are checkers: nothing under mujoco_jaco_amd/ or bench.py's timed path may import them."""
    pkg = os.path.join(ROOT, "mujoco_jaco_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert "oracle_binding" not in txt and "libjaco_oracle" not in txt and "emu_binding" not in txt, fn

"""ctypes binding of oracle/libjaco_oracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the checker; nothing under mujoco_jaco_amd/ imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ASSETS = os.path.join(ROOT, "mujoco_jaco_amd", "assets")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "libjaco_oracle.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(ORACLE_DIR, "jaco_oracle.c")):
            build()
        L = ctypes.CDLL(so)
        vp, cp, dp, ci, cd = ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_double
        L.orc_load_model.restype = vp
        L.orc_load_model.argtypes = [cp]
        L.orc_free_model.argtypes = [vp]
        L.orc_make_data.restype = vp
        L.orc_make_data.argtypes = [vp]
        L.orc_free_data.argtypes = [vp]
        L.orc_reset.argtypes = [vp, vp]
        L.orc_model_int.argtypes = [vp, cp]
        L.orc_set_option.argtypes = [vp, cp, cd]
        L.orc_set.argtypes = [vp, vp, cp, dp, ci]
        L.orc_get.argtypes = [vp, vp, cp, dp, ci]
        L.orc_ncon.argtypes = [vp]
        L.orc_nefc.argtypes = [vp]
        L.orc_solver_iter.argtypes = [vp]
        L.orc_forward.argtypes = [vp, vp]
        L.orc_step.argtypes = [vp, vp]
        L.orc_jac_body_com.argtypes = [vp, vp, ci, dp, dp]
        L.orc_step_batch.argtypes = [vp, ci, ci, dp, dp, dp, dp, dp, ci]
        L.orc_step_batch_stats.argtypes = [vp, ci, ci, dp, dp, dp, dp, dp, ci, ctypes.POINTER(ctypes.c_int)]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


class Oracle:
    """One fp64 reference environment (model + data)."""

    def __init__(self, model="jaco2_curtain_torque"):
        self.L = lib()
        path = os.path.join(ASSETS, model + ".jacomdl")
        self.m = self.L.orc_load_model(path.encode())
        if not self.m:
            raise RuntimeError("cannot load " + path)
        self.d = self.L.orc_make_data(self.m)
        for k in ("nq", "nv", "nu", "nbody", "ngeom", "nsite", "nmocap", "nsensor", "njnt"):
            setattr(self, k, self.L.orc_model_int(self.m, k.encode()))

    def __del__(self):
        try:
            self.L.orc_free_data(self.d)
            self.L.orc_free_model(self.m)
        except Exception:
            pass

    def option(self, name, value):
        assert self.L.orc_set_option(self.m, name.encode(), float(value)) == 0, name

    def reset(self):
        self.L.orc_reset(self.m, self.d)

    def set(self, name, value):
        a = np.ascontiguousarray(np.asarray(value, dtype=np.float64).reshape(-1))
        assert self.L.orc_set(self.m, self.d, name.encode(), _dp(a), a.size) == 0, (name, a.size)

    def get(self, name, n=None):
        sizes = {"qpos": self.nq, "qvel": self.nv, "ctrl": self.nu, "qacc_warmstart": self.nv, "xpos": 3 * self.nbody,
                 "xquat": 4 * self.nbody, "xmat": 9 * self.nbody, "xipos": 3 * self.nbody, "geom_xpos": 3 * self.ngeom,
                 "geom_xmat": 9 * self.ngeom, "site_xpos": 3 * self.nsite, "site_xmat": 9 * self.nsite,
                 "qM": self.nv ** 2, "sensordata": self.nsensor, "actuator_force": self.nu,
                 "mocap_pos": 3 * self.nmocap, "mocap_quat": 4 * self.nmocap}
        if n is None:
            if name in sizes:
                n = sizes[name]
            elif name.startswith("efc_J"):
                n = self.nefc * self.nv
            elif name.startswith("efc_"):
                n = self.nefc
            elif name == "contact":
                n = 11 * self.ncon
            else:
                n = self.nv
        out = np.zeros(n)
        assert self.L.orc_get(self.m, self.d, name.encode(), _dp(out), n) == 0, (name, n)
        return out

    @property
    def ncon(self):
        return self.L.orc_ncon(self.d)

    @property
    def nefc(self):
        return self.L.orc_nefc(self.d)

    @property
    def solver_iter(self):
        return self.L.orc_solver_iter(self.d)

    def forward(self):
        self.L.orc_forward(self.m, self.d)

    def step(self, ctrl=None, n=1):
        if ctrl is not None:
            self.set("ctrl", ctrl)
        for _ in range(n):
            self.L.orc_step(self.m, self.d)

    def jac_body_com(self, body):
        jp, jr = np.zeros(3 * self.nv), np.zeros(3 * self.nv)
        self.L.orc_jac_body_com(self.m, self.d, body, _dp(jp), _dp(jr))
        return jp.reshape(3, -1), jr.reshape(3, -1)

    def step_batch(self, qpos, qvel, qacc_ws, ctrl, nsub=1, nthreads=1, sensordata=None, stats=None):
        """In-place batched stepping of env-major fp64 arrays; stats: optional int32 [nenv, 4] (max ncon, max nefc, max iterations, last ncon)."""
        nenv = qpos.shape[0]
        for a in (qpos, qvel, qacc_ws, ctrl):
            assert a.dtype == np.float64 and a.flags.c_contiguous
        sp = _dp(sensordata) if sensordata is not None else None
        st = None
        if stats is not None:
            assert stats.dtype == np.int32 and stats.flags.c_contiguous and stats.shape == (nenv, 4)
            st = stats.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
        self.L.orc_step_batch_stats(self.m, nenv, nsub, _dp(qpos), _dp(qvel), _dp(qacc_ws), _dp(ctrl), sp, nthreads, st)

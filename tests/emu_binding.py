"""ctypes binding of tests/emu/libjaco_emu.so -- TEST INFRASTRUCTURE ONLY.

Runs the *unmodified* HIP kernel source (mujoco_jaco_amd/csrc/physics_kernel.h) compiled for the host
against a lockstep 64-lane wavefront emulator, so kernel logic can be checked against the oracle
without a GPU.  Not a product path: the product library refuses to run without a HIP device.
"""
import ctypes
import sys
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
EMU_DIR = os.path.join(ROOT, "tests", "emu")
ASSETS = os.path.join(ROOT, "mujoco_jaco_amd", "assets")
_libs = {}


def lib(layout=""):
    """layout "": the default build (11 bodies / 21 dofs in blocks 9 + 6 + 6); "_d12": the build for jaco2_torque.xml (12 hinge dofs, one tree);
    "_d30": the build for jaco2_dual_torque.xml (two arms + two objects, 30 dofs; ctrl level); "_wrench" / "_nolook" / "_mprpairs": A/B builds of the default layout
    (body-space constraint rows; the Newton solver without its look-ahead stop; MPR two pairs per wave)."""
    if layout not in _libs:
        subprocess.check_call(["make", "-s", "-C", EMU_DIR] + (["libjaco_emu%s.so" % layout] if layout in ("_wrench", "_nolook", "_mprpairs") else []))   # (A/B builds: on demand)
        L = ctypes.CDLL(os.path.join(EMU_DIR, "libjaco_emu%s.so" % layout))
        fp, ip, up = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint)
        L.emu_physics_step.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, fp, fp, fp, fp, up, ip, fp, ctypes.c_int, ip]
        L.emu_env_call.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong,
                                   fp, fp, fp, fp, up, ip, fp, fp, fp, fp, fp, fp, ctypes.POINTER(ctypes.c_ubyte), fp, ip]
        L.emu_marker_rest.argtypes = [ctypes.c_char_p, ctypes.c_long, fp]
        L.emu_set_auto_reset.argtypes = [ctypes.c_int, fp, ctypes.c_int]
        L.emu_reset_env.argtypes = [ctypes.c_int, ctypes.c_ulonglong, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, fp, fp, fp, fp, fp, fp, fp]
        _libs[layout] = L
    return _libs[layout]


class EmuEnv:
    """Batched env state (fp32, [nenv][n]) stepped by the emulated kernel."""

    def __init__(self, model="jaco2_curtain_torque", nenv=1, layout=None):
        self.blob = open(os.path.join(ASSETS, model + ".jacomdl"), "rb").read()
        from mujoco_jaco_amd.modelc import blob as blobmod
        M = blobmod.loads(self.blob)
        from mujoco_jaco_amd import _lib as product_lib
        self.L = lib(layout if layout is not None else product_lib.variant_for(self.blob))   # (the same layout choice as the product's loader)
        self.M = M
        self.nq, self.nv, self.nu, self.ns = int(M["nq"][0]), int(M["nv"][0]), int(M["nu"][0]), int(M["nsensor"][0])
        self.nenv = nenv
        self.qpos = np.tile(M["qpos0"].astype(np.float32), (nenv, 1))
        self.qvel = np.zeros((nenv, self.nv), np.float32)
        self.qacc_ws = np.zeros((nenv, self.nv), np.float32)
        self.sensordata = np.zeros((nenv, self.ns), np.float32)
        self.flags = np.zeros(nenv, np.uint32)
        self.stats = np.zeros((nenv, 4), np.int32)
        self.dbg = np.zeros(self.L.emu_dbg_size(), np.float32)

    def step(self, ctrl, nsub=1, disable_contact=False, dbg_env=-1):
        ctrl = np.ascontiguousarray(np.broadcast_to(np.asarray(ctrl, np.float32), (self.nenv, self.nu)))
        fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        hv = ctypes.c_int(0)
        rc = self.L.emu_physics_step(self.blob, len(self.blob), self.nenv, nsub, int(disable_contact), fp(self.qpos), fp(self.qvel),
                                     fp(self.qacc_ws), fp(ctrl), fp(self.sensordata),
                                     self.flags.ctypes.data_as(ctypes.POINTER(ctypes.c_uint)),
                                     self.stats.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                     fp(self.dbg) if dbg_env >= 0 else None, dbg_env, ctypes.byref(hv))
        self.heavy_envs = hv.value
        assert rc == 0


class EmuJacoEnv(EmuEnv):
    """Env-level calls (jaco_step / jaco_forward semantics) on the emulated kernels."""

    def __init__(self, model="jaco2_curtain_torque", nenv=1, task_id=0, frame_skip=50, seed=0):
        super().__init__(model, nenv)
        self.task_id, self.frame_skip, self.seed = task_id, frame_skip, seed
        self.task = np.zeros((nenv, self.L.emu_task_floats()), np.float32)
        self.cache = np.zeros((nenv, self.L.emu_cache_floats()), np.float32)
        self.obs = np.zeros((nenv, 26), np.float32)
        self.reward = np.zeros(nenv, np.float32)
        self.done = np.zeros(nenv, np.uint8)
        self.task[:, 0] = 0.6; self.task[:, 16] = 0.6
        rest = np.zeros(24, np.float32)
        assert self.L.emu_marker_rest(self.blob, len(self.blob), rest.ctypes.data_as(ctypes.POINTER(ctypes.c_float))) == 0
        self.marker = np.tile(rest, (nenv, 1))   # ["hand", "subgoal_reach"] x (position, rotation): parked at the XML pose

    def _call(self, mode, action=None, noise=None):
        fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if a is not None else None
        hv = ctypes.c_int(0)
        self.L.emu_set_obs_mode(int(getattr(self, "obs_mode", 0)))
        rc = self.L.emu_env_call(self.blob, len(self.blob), self.nenv, mode, self.frame_skip, self.task_id, 6 if self.task_id in (2, 7) else 7, self.seed,
                                 fp(self.qpos), fp(self.qvel), fp(self.qacc_ws), fp(self.sensordata),
                                 self.flags.ctypes.data_as(ctypes.POINTER(ctypes.c_uint)), self.stats.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                 fp(self.task), fp(self.cache), fp(action), fp(noise), fp(self.obs), fp(self.reward),
                                 self.done.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), fp(self.marker), ctypes.byref(hv))
        assert rc == 0
        self.heavy_envs = hv.value

    def placing_hold(self, nsub=150):
        """mode 3: the held part of the placing reset (jaco_reset runs it between the reset kernel and jaco_forward)."""
        fs, self.frame_skip = self.frame_skip, nsub
        try:
            self._call(3, None, None)
        finally:
            self.frame_skip = fs

    def grasping_prereach(self, cap=4000, noise=None):
        """mode 6: the pre-reach loops of the grasping reset (jaco_reset runs them after the draws); returns the observation row."""
        fs, self.frame_skip = self.frame_skip, cap
        try:
            self._call(6, None, None if noise is None else np.ascontiguousarray(noise, np.float32))
        finally:
            self.frame_skip = fs
        return self.obs.copy()

    def set_auto_reset(self, on):
        """option "auto_reset": a finished env is reset (draws + sim.forward() + first observation) by the wave that finished it."""
        q0 = np.ascontiguousarray(self.M["qpos0"], np.float32)
        self.L.emu_set_auto_reset(int(on), q0.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), self.nq)

    def set_init_buffer(self, rows):
        """kwarg init_buffer (jaco_set_init_buffer of the library): recorded rows every reset draws its reaching goal from; None = sampled goals."""
        self.L.emu_set_init_buffer.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_int]
        if rows is None:
            self.L.emu_set_init_buffer(None, 0, 0)
            return
        r = np.ascontiguousarray(rows, np.float32)
        self.L.emu_set_init_buffer(r.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), r.shape[0], r.shape[1])

    def reset_env(self, env, noise=None):
        """jaco_reset(mask = {env}): the reset kernel's work for one env, then the forward pass + observation (mode 2; here for all envs,
        which is harmless for the others: a forward pass does not change state)."""
        fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        q0 = np.ascontiguousarray(self.M["qpos0"], np.float32)
        base = np.ascontiguousarray(self.M["f_pos"].reshape(-1, 3)[0], np.float32)
        rest = np.zeros(24, np.float32)
        assert self.L.emu_marker_rest(self.blob, len(self.blob), fp(rest)) == 0
        self.L.emu_reset_env(self.task_id, self.seed, env, self.nq, self.nv, fp(q0), fp(base), fp(self.qpos), fp(self.qvel), fp(self.qacc_ws), fp(self.task),
                             fp(self.marker), fp(rest))

    def forward(self, noise=None):
        self._call(2, None, None if noise is None else np.ascontiguousarray(noise, np.float32))
        return self.obs.copy()

    def env_step(self, action, noise=None):
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(action, np.float32), (self.nenv, 6 if self.task_id in (2, 7) else 7)))
        self._call(1, a, None if noise is None else np.ascontiguousarray(noise, np.float32))
        return self.obs.copy(), self.reward.copy(), self.done.copy()

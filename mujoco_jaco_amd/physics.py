"""Ctrl-level batched physics: the `Mujoco` sim-interface tier of the reference, batched.

Mirrors /root/reference/env_script/mujoco.py (send_forces :258-278, set_joint_state :332-347,
get_feedback :349-359, get/set state :213-246) for `num_envs` environments resident on one MI355X.
PyTorch is used only to own device buffers and streams; every computation is in libjaco_env.so.
"""
import ctypes

import numpy as np
import torch

from . import _lib


class JacoError(RuntimeError):
    pass


class BatchedMujoco:
    def __init__(self, num_envs, robot_file="jaco2_curtain_torque", device=0, frame_skip=50, task=0, seed=0):
        if not torch.cuda.is_available():
            raise JacoError("BatchedMujoco needs a HIP device (no CPU path exists)")
        self.device = torch.device("cuda", device)
        self._blob = open(_lib.model_path(robot_file), "rb").read()
        self.L = _lib.load(_lib.variant_for(self._blob))
        self._blob_buf = ctypes.create_string_buffer(self._blob, len(self._blob))
        cfg = _lib.JacoConfig(ctypes.cast(self._blob_buf, ctypes.c_void_p), len(self._blob), int(num_envs), int(device),
                              int(frame_skip), int(task), int(seed))
        self.h = ctypes.c_void_p()
        rc = self.L.jaco_create(ctypes.byref(cfg), ctypes.byref(self.h))
        if rc != 0:
            raise JacoError("jaco_create failed (%d): %s" % (rc, self.L.jaco_last_error(None).decode()))
        dims = [ctypes.c_int() for _ in range(6)]
        self._chk(self.L.jaco_dims(self.h, *[ctypes.byref(d) for d in dims]))
        self.nq, self.nv, self.nu, self.nsensor, self.nobs, self.nact = [d.value for d in dims]
        self.num_envs = int(num_envs)

    def _chk(self, rc):
        if rc != 0:
            raise JacoError("libjaco_env error %d: %s" % (rc, self.L.jaco_last_error(self.h).decode()))

    def close(self):
        if getattr(self, "h", None):
            self.L.jaco_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _dev(self, t, n, dtype=torch.float32):
        if t is None:
            return None
        assert t.is_cuda and t.dtype == dtype and t.is_contiguous() and t.numel() == self.num_envs * n, (t.shape, t.dtype, n)
        return ctypes.c_void_p(t.data_ptr())

    # ---- state (sim.get_state / set_state)
    def set_state(self, qpos=None, qvel=None, qacc_warmstart=None):
        self._chk(self.L.jaco_set_state(self.h, self._dev(qpos, self.nq), self._dev(qvel, self.nv),
                                        self._dev(qacc_warmstart, self.nv), self._stream()))

    def get_state(self):
        qpos = torch.empty(self.num_envs, self.nq, device=self.device)
        qvel = torch.empty(self.num_envs, self.nv, device=self.device)
        qacc = torch.empty(self.num_envs, self.nv, device=self.device)
        self._chk(self.L.jaco_get_state(self.h, self._dev(qpos, self.nq), self._dev(qvel, self.nv), self._dev(qacc, self.nv), self._stream()))
        return qpos, qvel, qacc

    def state_views(self):
        """Copies of (qpos, qvel, qacc_warmstart) as [num_envs, n] tensors on the current stream."""
        return self.get_state()

    def reset_state(self):
        self._chk(self.L.jaco_reset_state(self.h, self._stream()))

    # ---- send_forces
    def send_forces(self, ctrl, nsub=1):
        self._chk(self.L.jaco_physics_step(self.h, self._dev(ctrl, self.nu), int(nsub), self._stream()))

    def send_forces_debug(self, ctrl, env, nsub=1):
        n = self.L.jaco_debug_dump_floats()
        out = np.zeros(n, np.float32)
        torch.cuda.synchronize()
        self._chk(self.L.jaco_physics_step_debug(self.h, self._dev(ctrl, self.nu), int(nsub), int(env),
                                                 out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), n))
        return out

    def sensordata(self):
        out = torch.empty(self.num_envs, self.nsensor, device=self.device)
        self._chk(self.L.jaco_get_sensordata(self.h, self._dev(out, self.nsensor), self._stream()))
        return out

    def flags(self):
        out = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        self._chk(self.L.jaco_get_flags(self.h, self._dev(out, 1, torch.int32), self._stream()))
        return out

    def clear_flags(self):
        self._chk(self.L.jaco_clear_flags(self.h, self._stream()))

    def stats(self):
        out = torch.empty(self.num_envs, 4, dtype=torch.int32, device=self.device)
        self._chk(self.L.jaco_get_stats(self.h, self._dev(out, 4, torch.int32), self._stream()))
        return out

    def set_option(self, name, value):
        self._chk(self.L.jaco_set_option(self.h, name.encode(), float(value)))

    def launch_count(self):
        """Kernel launches issued for this handle since the previous call (host-side counter)."""
        return int(self.L.jaco_launch_count(self.h))

    def enable_timing(self, on=True):
        self._chk(self.L.jaco_enable_timing(self.h, int(on)))

    def step_time_ms(self):
        """Mean device time of a whole step launch set in the timing window (call before kernel_time_ms, which closes it)."""
        ms = ctypes.c_double()
        self._chk(self.L.jaco_step_time_ms(self.h, ctypes.byref(ms)))
        return ms.value

    def kernel_time_ms(self):
        ms, n = ctypes.c_double(), ctypes.c_int()
        self._chk(self.L.jaco_kernel_time_ms(self.h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

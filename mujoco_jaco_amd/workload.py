"""Synthetic workloads shared by bench.py, the GPU tests and the probes.

`reset_states` draws the reference's `picking` reset distribution (env_mujoco_util.py:177-180,
:215-217, Appendix A of SURVEY.md): arm joint angles, fingers at qpos0, object on the holder,
destination pedestal; `random_ctrl` draws ctrl = U(-1,1) * motor force range, fingers held at 0.6.
"""
import numpy as np


def reset_states(qpos0, nenv, seed=0, contact=True, f32_draws=False):
    """f32_draws: the drawn coordinates are rounded to fp32 (so an fp32 and an fp64 engine can be handed the same numbers), the
    model's own constants (finger angles 1.1, pedestal height 0.09, ...) stay exactly what the XML says -- the pedestal's
    bottom face is at z = 0.09 + 0.07 - 0.16 = 0 by construction, a knife edge that the fp32-rounded 0.09f would move 3.6 nm
    above the floor for an fp64 engine (no contact in the first step) while the fp32 engine still computes exactly 0."""
    rng = np.random.default_rng(seed)
    nq = len(qpos0)
    q = np.tile(np.asarray(qpos0, np.float64), (nenv, 1))
    r = (lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)) if f32_draws else (lambda a: a)
    lo = np.array([0.7, 3.8, 1.0, 1.8, 1.0, 0.8]); hi = np.array([2.5, 4.0, 1.7, 2.5, 2.5, 2.3])
    q[:, :6] = r(rng.uniform(lo, hi, (nenv, 6)))
    if nq >= 23:
        q[:, 9] = r(rng.uniform(-0.1, 0.1, nenv))
        q[:, 10] = r(0.65 + rng.uniform(-0.08, 0.02, nenv))
        q[:, 11] = r(0.1898 if contact else 0.5)
        q[:, 12:16] = [1, 0, 0, 0]
        q[:, 16] = r(0.4 + rng.uniform(-0.05, 0.05, nenv))
        q[:, 17] = r(0.3 + rng.uniform(-0.05, 0.05, nenv))
    elif nq == 19 and abs(qpos0[13] - 0.65) < 1e-9:
        # jaco2_curtain_torque_sensor.xml (xml:283: the object starts 0.6 m above its holder): put it on the holder's disc (cylinder, top at
        # z = 0.41), the fingers half open (closed to 0 they touch each other in this model)
        q[:, 6:12:2] = r(rng.uniform(0.3, 0.7, (nenv, 3)))
        q[:, 12] = r(rng.uniform(-0.04, 0.04, nenv)); q[:, 13] = r(0.65 + rng.uniform(-0.04, 0.04, nenv)); q[:, 14] = r(0.4401 if contact else 0.8)
    return q


def random_ctrl(nenv, seed=1, scale=1.0, grip=0.6):
    rng = np.random.default_rng(seed)
    c = np.zeros((nenv, 9))
    c[:, :6] = rng.uniform(-1, 1, (nenv, 6)) * np.array([30, 30, 30, 15, 15, 15]) * scale
    c[:, 6:] = grip
    return c


def reset_states_dual(qpos0, nenv, seed=0, f32_draws=True):
    """jaco2_dual_torque.xml: both arms in the picking pose range (env_mujoco_util.py:177-180), both objects resting on their holders
    (object_holder_1/2 at (-+0.5, 0.6): top at z = 0.17, object half height 0.03)."""
    rng = np.random.default_rng(seed)
    q = np.tile(np.asarray(qpos0, np.float64), (nenv, 1))
    lo = np.array([0.7, 3.8, 1.0, 1.8, 1.0, 0.8]); hi = np.array([2.5, 4.0, 1.7, 2.5, 2.5, 2.3])
    q[:, 0:6] = rng.uniform(lo, hi, (nenv, 6)); q[:, 9:15] = rng.uniform(lo, hi, (nenv, 6))
    q[:, 18:21] = [-0.5, 0.6, 0.2001]; q[:, 25:28] = [0.5, 0.6, 0.2001]
    return q.astype(np.float32).astype(np.float64) if f32_draws else q

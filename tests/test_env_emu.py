"""CPU tier: env-level step of the emulated HIP kernels (OSC + substeps + obs/reward/done) against the fp64 oracle
env (tests/oracle_env.py = physics oracle + oracle/glue.py, the latter pinned by the reference's golden vectors)."""
import numpy as np
import pytest

from emu_binding import EmuJacoEnv
from mujoco_jaco_amd import workload
from oracle_env import OracleEnv


def _pair(names, M, seed, fs):
    e = EmuJacoEnv(frame_skip=fs)
    oe = OracleEnv(names, frame_skip=fs)
    q = workload.reset_states(M["qpos0"], 1, seed=seed)[0]
    oe.obj_goal = q[9:12].copy(); oe.dest_goal = np.array([q[16], q[17], 0.3468])
    oe.set_state(q.astype(np.float32).astype(np.float64))
    e.qpos[0] = q; e.task[0, 4:7] = oe.obj_goal; e.task[0, 7:10] = oe.dest_goal
    return e, oe


def test_env_step_matches_oracle_env(names, model_arrays):
    e, oe = _pair(names, model_arrays, 3, 10)
    rng = np.random.default_rng(3)
    nz = rng.uniform(size=(1, 12)).astype(np.float32)
    obs0 = e.forward(nz)
    assert np.abs(obs0[0] - oe.observe(nz[0, 6:].astype(np.float64))[0]).max() < 2e-6
    for step in range(6):
        a = rng.uniform(-1, 1, 7).astype(np.float32); nz = rng.uniform(size=(1, 12)).astype(np.float32)
        obs, rew, done = e.env_step(a, nz)
        oo, orew, odone, _ = oe.step(a.astype(np.float64), nz[0].astype(np.float64))
        assert obs[0, 0] == oo[0] and bool(done[0]) == odone              # touch class and done: exact
        assert np.abs(obs[0] - oo).max() < 5e-5 and abs(rew[0] - orew) < 1e-4
    assert e.task[0, 1] == 6 and e.task[0, 2] == 6                         # current_steps, num_episodes


def test_gripper_ramp_and_target(names, model_arrays):
    e, oe = _pair(names, model_arrays, 4, 5)
    e.forward()
    a = np.array([0.5, -0.2, 0.1, 0.3, -0.4, 0.9, 1.0], np.float32)
    e.env_step(a, np.full((1, 12), 0.5, np.float32))
    assert abs(e.task[0, 0] - 0.7) < 1e-6 and abs(e.task[0, 16] - 0.6) < 1e-6   # gripper += a6/10, clipped to [0.6, 1]
    import glue
    pe, qe = oe._ee()
    target, _, _ = glue.take_action(pe, qe, a.astype(np.float64), 0.6, 5)
    assert np.abs(e.task[0, 10:16] - target).max() < 1e-5


def test_timeout_and_freeze(names, model_arrays):
    e, _ = _pair(names, model_arrays, 5, 1)
    e.forward()
    e.task[0, 1] = 698                                                   # current_steps
    z = np.zeros(7, np.float32)
    _, _, d = e.env_step(z); assert d[0] == 0
    _, r, d = e.env_step(z); assert d[0] == 1 and abs(r[0] - (-10.0)) < 0.2   # time out: -10 (+ small shaping reward)
    q = e.qpos.copy()
    _, r, d = e.env_step(z)                                              # frozen until reset
    assert d[0] == 1 and r[0] == 0 and np.array_equal(q, e.qpos)


def test_osc_pseudo_inverse_branch(names, model_arrays):
    """Wrist-singular arm pose (joint4 ~ 0): |det(J M^-1 J^T)| < 1e-3 -> abr_control's SVD branch (singular values < 0.005
    dropped).  The kernel's Jacobi pseudo-inverse must reproduce the oracle's np.linalg.svd path."""
    import glue
    e = EmuJacoEnv(frame_skip=4)
    oe = OracleEnv(names, frame_skip=4)
    q = model_arrays["qpos0"].copy()
    q[:6] = [0.9757, 4.054, 1.8082, 1.6956, -0.0194, 1.6474]
    q[9:12] = [0.0, 0.65, 0.1898]; q[16:18] = [0.4, 0.3]
    oe.obj_goal = q[9:12].copy(); oe.dest_goal = np.array([0.4, 0.3, 0.3468])
    oe.set_state(q.astype(np.float32).astype(np.float64))
    o = oe.o
    jp, jr = o.jac_body_com(oe.ee); J = np.vstack([jp[:, :6], jr[:, :6]]); M6 = o.get("qM").reshape(21, 21)[:6, :6]
    assert abs(np.linalg.det(J @ np.linalg.inv(M6) @ J.T)) < 1e-3          # the scenario really is in the SVD branch
    e.qpos[0] = q; e.task[0, 4:7] = oe.obj_goal; e.task[0, 7:10] = oe.dest_goal
    nz = np.full((1, 12), 0.5, np.float32)
    e.forward(nz)
    a = np.array([0.3, -0.5, 0.2, 0.1, 0.4, -0.2, 0.0], np.float32)
    obs, rew, done = e.env_step(a, nz)
    oo, orew, odone, _ = oe.step(a.astype(np.float64), nz[0].astype(np.float64))
    assert e.flags[0] & 64                                                   # informational flag: branch taken
    assert np.abs(e.qpos[0, :9] - o.get("qpos")[:9]).max() < 2e-4 and np.abs(obs[0] - oo).max() < 2e-4


def _placing_state(M, seed):
    """Arm angles of the reference's placing reset (env_mujoco_util.py:181-185); object / pedestal as sampled by __sample_goal."""
    rng = np.random.default_rng(seed)
    q = workload.reset_states(M["qpos0"], 1, seed=seed)[0]
    a0 = rng.uniform(3 * np.pi / 8, np.pi / 2) if rng.uniform() < 0.5 else rng.uniform(np.pi / 2, 5 * np.pi / 8)
    q[:6] = [a0, 3.85, rng.uniform(1, 1.1), rng.uniform(2, 2.1), rng.uniform(0.8, 2.3), rng.uniform(-1.2, -1.1)]
    return q


def test_placing_hold_matches_oracle(names, model_arrays):
    """Placing reset, object part: object pinned to the grasp frame while the fingers close for `nsub` controlled substeps
    (fresh controller data every substep).  Short hold here (the GPU tier runs the reference's 150)."""
    nsub = 25
    q = _placing_state(model_arrays, 11)
    e = EmuJacoEnv(task_id=1)
    oe = OracleEnv(names, task="placing")
    oe.set_state(q.astype(np.float32).astype(np.float64))
    e.qpos[0] = q
    e.placing_hold(nsub)
    pos, quat, target = oe.placing_hold(nsub)
    # object pose: pinned (bit-for-bit the value written to the cache row), quaternion equal up to sign
    pin = e.cache[0, 96:103]
    assert np.array_equal(e.qpos[0, 9:16], pin) and np.all(e.qvel[0, 9:] == 0)
    sgn = 1.0 if np.dot(pin[3:], quat) > 0 else -1.0
    assert np.abs(pin[:3] - pos).max() < 1e-6 and np.abs(sgn * pin[3:] - quat).max() < 1e-6
    assert np.abs(e.task[0, 10:16] - target).max() < 1e-5                      # controller target = EE pose at reset
    oq = oe.o.get("qpos")
    assert np.abs(e.qpos[0, :9] - oq[:9]).max() < 2e-4                         # arm + fingers after the hold
    assert e.qpos[0, 6:9].max() < q[6:9].min() - 0.003                         # the finger servos moved towards the 0.6 command
    assert e.task[0, 17] == 0 and e.task[0, 19] == 0                           # substep counter / pending flag cleared


def test_tier_alternation(names, model_arrays):
    """An env whose first substeps overflow the light capacities (hand spawned inside the pedestal at reset): light code ->
    heavy code for the burst -> back to the light code for the rest of the step (flag 128), all inside one env step.
    The result must not depend on where the env was stepped: same observation as with the hand-back disabled (env stays
    on the heavy code), and both close to the fp64 oracle env, which knows no tiers (loose bound: the separation of
    the interpenetrating bodies is violent and amplifies rounding differences)."""
    from emu_binding import lib
    seed = 108
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1, 1, 7).astype(np.float32); a[:3] = np.sign(a[:3]) * np.maximum(np.abs(a[:3]), 0.5)
    nz = np.full((1, 12), 0.5, np.float32)
    out = {}
    try:
        for tier_return in (1, 0):
            lib().emu_set_tier_return(tier_return)
            e, oe = _pair(names, model_arrays, seed, 30)
            e.forward(nz)
            obs, rew, done = e.env_step(a, nz)
            out[tier_return] = (obs[0].copy(), int(e.flags[0]), bool(done[0]))
    finally:
        lib().emu_set_tier_return(1)
    assert out[1][1] & 32 and out[1][1] & 128 and not out[0][1] & 128
    assert np.abs(out[1][0] - out[0][0]).max() < 1e-5
    oo, orew, odone, _ = oe.step(a.astype(np.float64), nz[0].astype(np.float64))
    assert out[1][2] == odone and out[1][0][0] == oo[0] and np.abs(out[1][0] - oo).max() < 3e-3


def test_hand_down_to_second_medium_drain(names, model_arrays):
    """The same scenario through the two-round drain sequence (jaco_env.hip): the heavy drain does not keep the calmed-down env for
    the rest of its step but queues it for the medium tier again (a second medium drain, then a final heavy drain).  Which
    workgroup steps an env must not matter: bit-identical observation, reward and state."""
    from emu_binding import lib
    seed = 108
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1, 1, 7).astype(np.float32); a[:3] = np.sign(a[:3]) * np.maximum(np.abs(a[:3]), 0.5)
    nz = np.full((1, 12), 0.5, np.float32)
    out = {}
    before = lib().emu_handed_down()
    try:
        for down in (0, 1):
            lib().emu_set_handdown(down)
            e, oe = _pair(names, model_arrays, seed, 30)
            e.forward(nz)
            obs, rew, done = e.env_step(a, nz)
            out[down] = (obs[0].copy(), float(rew[0]), e.qpos[0].copy(), e.qvel[0].copy(), int(e.flags[0]))
    finally:
        lib().emu_set_handdown(0)
    assert lib().emu_handed_down() > before                     # the env really took the second round
    assert out[1][4] & 32 and out[1][4] & 128
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
    assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3])


def test_reaching_task_matches_oracle_env(names, model_arrays):
    """Task 'reaching' (env_mujoco_util.py:192-207, 314-351, 504-520): 6-wide action (gripper held at 0.6), reaching reward,
    reaching termination with the success flag, time-out at 500 steps, and the rulebased_subgoal = False observation
    (reaching goal in obs[17:23]); glue pinned by tests/golden/glue_vectors_reaching.npz."""
    import glue
    fs = 8
    e = EmuJacoEnv(frame_skip=fs, task_id=2); e.obs_mode = 1
    oe = OracleEnv(names, task="reaching", frame_skip=fs); oe.rulebased = False
    q = workload.reset_states(model_arrays["qpos0"], 1, seed=9)[0]
    oe.obj_goal = q[9:12].copy(); oe.dest_goal = np.array([q[16], q[17], 0.3468])
    oe.set_state(q.astype(np.float32).astype(np.float64))
    pe, qe = oe._ee()
    # a goal 1.5 cm from the EE with a slightly rotated orientation: reached within a few steps
    goal = np.concatenate([pe + [0.012, -0.006, 0.005], glue.euler_from_quat(qe) + [0.05, -0.04, 0.03]]).astype(np.float32).astype(np.float64)
    oe.reach_goal = goal
    e.qpos[0] = q; e.task[0, 4:7] = oe.obj_goal; e.task[0, 7:10] = oe.dest_goal; e.task[0, 32:38] = goal
    obs0 = e.forward()
    assert np.abs(obs0[0] - oe.observe(None)[0]).max() < 2e-6 and np.allclose(obs0[0, 17:20], goal[:3], atol=1e-7)
    rng = np.random.default_rng(9)
    reached = False
    for step in range(5):
        a = np.concatenate([(goal[:3] - oe._ee()[0]) * 25 * 0.8, rng.uniform(-0.05, 0.05, 3)]).clip(-1, 1).astype(np.float32)
        obs, rew, done = e.env_step(a)
        oo, orew, odone, osucc = oe.step(a.astype(np.float64), np.full(12, 0.5))
        assert bool(done[0]) == odone and np.abs(obs[0] - oo).max() < 5e-5 and abs(rew[0] - orew) < 2e-3
        assert abs(e.task[0, 0] - 0.6) < 1e-7                                # 6-wide action: gripper command stays 0.6
        if odone:
            reached = True
            assert orew > 100 and e.task[0, 29] == 1.0 == float(osucc)      # success flag (the reference's 3-tuple lacks it)
            break
    assert reached
    # time-out of a non-picking task: 500 steps (env_mujoco.py:20-21)
    e2 = EmuJacoEnv(frame_skip=1, task_id=2); e2.obs_mode = 1
    e2.qpos[0] = q; e2.task[0, 32:38] = goal + 1.0; e2.forward(); e2.task[0, 1] = 498
    z = np.zeros(6, np.float32)
    _, _, d = e2.env_step(z); assert d[0] == 0
    _, r, d = e2.env_step(z); assert d[0] == 1 and abs(r[0] - (-10.0)) < 0.2


def test_non_finite_state_is_quarantined_not_followed_out_of_bounds(names, model_arrays):
    """A NaN / Inf anywhere in the state must neither index outside a table (a hull support scan used to return index
    0x7fffffff for a NaN direction) nor loop forever: the env ends its episode with reward 0 and JACO_FLAG_NAN and freezes."""
    for adr, arr, val in ((2, "qvel", np.nan), (12, "qpos", np.nan), (4, "qvel", np.inf)):
        e, _ = _pair(names, model_arrays, 4, 6)
        e.forward()
        getattr(e, arr)[0, adr] = val
        obs, rew, done = e.env_step(np.zeros(7, np.float32))
        assert done[0] == 1 and rew[0] == 0 and (e.flags[0] & 8) and np.isfinite(obs).all(), (arr, adr)
        obs, rew, done = e.env_step(np.zeros(7, np.float32))
        assert done[0] == 1 and rew[0] == 0


def test_action_taken_once_when_a_step_overflows_in_its_first_substep(names, model_arrays):
    """A zero motion action puts the "hand" marker's sticks onto the EE's axis sticks (env_mujoco_util.py:644-646): ~46 contacts /
    184 rows in the very first substep, so the light tier hands the env on before any substep has run -- and so do the medium
    and heavy tiers.  Each tier used to run _take_action again for the resumed step (JT_SUB == 0): the gripper command moved by
    a[6] / 10 once per tier (found by the closed-loop drift leg, tools/env_drift.py: finger angles 0.03 rad off the oracle)."""
    e, oe = _pair(names, model_arrays, 3, 6)
    nz = np.full((1, 12), 0.5, np.float32)
    e.forward(nz)
    a = np.array([0, 0, 0, 0, 0, 0, 1.0], np.float32)
    for step in range(2):
        obs, rew, done = e.env_step(a, nz)
        oo, orew, odone, _ = oe.step(a.astype(np.float64), nz[0].astype(np.float64))
        assert e.flags[0] & 32 and e.stats[0, 1] > 128                       # the scenario really left the light and the medium tier
        assert abs(e.task[0, 0] - (0.7 + 0.1 * step)) < 1e-6 and abs(e.task[0, 16] - (0.6 + 0.1 * step)) < 1e-6   # gripper: one increment per step
        assert abs(obs[0, 7] - oo[7]) < 1e-6 and np.abs(e.qpos[0, 6:9] - oe.o.get("qpos")[6:9]).max() < 1e-4
        assert int(e.task[0, 18:19].view(np.uint32)[0]) & 0x1FFFFFFF == 12 * (step + 1) + 6   # draw counter (JT_RNG: count | bit 30): 6 (reset obs) + 12 per step


# ---------------------------------------------------------------- tasks grasping / pickAndplace (round 3)
def _quat_to_mat(q):
    w, x, y, z = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def _inject_scene(e, model_arrays, ee, eeq, obj, q2, touch):
    """Put a golden scene into what terminal_inspection reads: the cached frame of the body that carries the EE (so that the kernel's
    EE frame = the given pose), the object position, joint 2, and a sensordata pattern of the given touch class."""
    f = model_arrays["f_frame_EE"]
    pl, Rl = f[1:4], _quat_to_mat(f[4:8])
    Rb = _quat_to_mat(eeq / np.linalg.norm(eeq)) @ Rl.T
    pb = ee - Rb @ pl
    e.cache[0, 78:81] = pb; e.cache[0, 81:90] = Rb.reshape(-1); e.cache[0, 90:93] = obj
    e.qpos[0, 2] = q2
    s = np.zeros(20, np.float32)
    if touch == 1: s[1] = 1.0                   # 0_touch: an inner pad alone
    if touch == 2: s[0] = 1.0                   # EE_touch: outer
    if touch == 3: s[2] = 1.0; s[6] = 1.0       # 1_touch (thumb) + 5_touch (index)
    e.sensordata[0] = s


def test_kernel_terminal_inspection_against_the_references_own_outputs(model_arrays):
    """The in-kernel termination rules of tasks grasping and pickAndplace (csrc/env_logic.h), driven through the terminal_inspection
    entry (mode 5), against tests/golden/glue_vectors_grasping.npz = outputs of the reference's own _get_terminal_inspection
    (env_mujoco_util.py:521-536, 585-600): done flag exact, bonus to fp32 rounding, the `picked` flag carried in the task row."""
    import os
    GG = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glue_vectors_grasping.npz"))
    n = len(GG["g_reward"])
    for task_id, key in ((3, "g_term_grasp"), (4, "g_term_pp")):
        e = EmuJacoEnv(task_id=task_id, frame_skip=1)
        for k in range(0, n, 2):
            _inject_scene(e, model_arrays, GG["g_ee"][k], GG["g_eeq"][k], GG["g_obj"][k], GG["g_q2"][k], int(GG["g_touch"][k]))
            e.task[0] = 0; e.task[0, 0] = 0.6; e.task[0, 16] = 0.6
            e.task[0, 2] = GG["g_nsteps"][k]; e.task[0, 7:10] = GG["g_dest_goal"][k]; e.task[0, 31] = GG["g_picked_in"][k]
            e._call(5)
            rd, rb, rwb = GG[key][k]
            assert bool(e.done[0]) == bool(rd), (task_id, k)
            assert abs(e.reward[0] - rb) < 2e-4 and abs(e.task[0, 30] - rwb) < 1e-5, (task_id, k, e.reward[0], rb)
            if task_id == 4:
                assert e.task[0, 31] == GG["g_picked_out"][k] and e.task[0, 29] == float(rb == 180.0)
            else:
                assert e.task[0, 29] == float(bool(rd) and rb > 100)     # success flag (the 4th value the reference's tuple lacks)
            assert e.task[0, 2] == GG["g_nsteps"][k] + 1 and e.task[0, 1] == 1
            if rd:   # the terminal step latches (success, wb) for jaco_get_last_terminal (what survives an in-kernel reset)
                import ctypes
                e.L.emu_last_terminal.restype = ctypes.POINTER(ctypes.c_float)
                lt = e.L.emu_last_terminal()
                assert lt[0] == e.task[0, 29] and abs(lt[1] - rwb) < 1e-5


def test_grasping_prereach_and_steps_match_oracle_env(names, model_arrays):
    """Grasping reset (env_mujoco_util.py:123-170) from two arm poses of its init range (:181-185): the emulated kernel's pre-reach
    (mode 6) must take the oracle env's path -- same loop exits, same EE target, observation and arm state -- and the env steps
    that follow must agree in observation, reward (:352-391) and termination."""
    for seed, expect_min in ((18, 50), (13, 1)):
        rng = np.random.default_rng(100 + seed)
        q = model_arrays["qpos0"].copy()
        a0 = rng.uniform(3 * np.pi / 8, np.pi / 2) if seed % 2 else rng.uniform(np.pi / 2, 5 * np.pi / 8)
        q[:6] = [a0, 3.85, rng.uniform(1, 1.1), rng.uniform(2, 2.1), rng.uniform(0.8, 2.3), rng.uniform(-1.2, -1.1)]
        q[9:12] = [rng.uniform(-.1, .1), 0.65 + rng.uniform(-.08, .02), 0.1898]; q[16:18] = [0.42, 0.31]
        f32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
        q[:6], q[9:11], q[16:18] = f32(q[:6]), f32(q[9:11]), f32(q[16:18])       # drawn coordinates fp32-representable; model constants exact
        e = EmuJacoEnv(task_id=3, frame_skip=10)
        oe = OracleEnv(names, task="grasping", frame_skip=10)
        oe.obj_goal = f32(q[9:12]); oe.dest_goal = f32([q[16], q[17], 0.3468])
        oe.reach_goal = np.array([0.35, -0.33, 0.4, 0.2, 1.0, 0.05])
        oe.set_state(q)
        e.qpos[0] = q; e.task[0, 4:7] = oe.obj_goal; e.task[0, 7:10] = oe.dest_goal; e.task[0, 32:38] = oe.reach_goal
        nz = np.full((1, 12), 0.5, np.float32); nz[0, 0] = 0.8
        n = oe.grasping_prereach(float(-0.1 + 0.2 * np.float32(0.8)))
        obs = e.grasping_prereach(4000, nz)
        assert n >= expect_min and e.task[0, 38] == 3 and not (e.flags[0] & 0x20000)            # both loops left, no cap
        assert np.abs(e.task[0, 10:16] - oe.target).max() < 1e-6                                 # self.target_pos after the reset
        oo = oe.observe(nz[0, 6:].astype(np.float64))[0]
        assert np.abs(obs[0] - oo).max() < 2e-5 and np.abs(e.qpos[0, :9] - oe.o.get("qpos")[:9]).max() < 2e-5, (seed, n)
        for step in range(2):
            a = rng.uniform(-1, 1, 7).astype(np.float32); nz = rng.uniform(size=(1, 12)).astype(np.float32)
            ob, rew, done = e.env_step(a, nz)
            if step == 0:
                oe.grip = 0.6
            o2, orew, odone, _ = _oracle_step_with_target(oe, a, nz[0])
            assert bool(done[0]) == odone and ob[0, 0] == o2[0]
            assert np.abs(ob[0] - o2).max() < 1e-4 and abs(rew[0] - orew) < 1e-4, (seed, step)


def _oracle_step_with_target(oe, a, nz):
    return oe.step(a.astype(np.float64), nz.astype(np.float64))


def test_kernel_terminal_inspection_picking_placing_against_the_references_own_outputs(model_arrays):
    """Same for the two tasks the reference runs end to end (env_mujoco_util.py:537-548,567-582): tests/golden/glue_vectors.npz."""
    import os
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glue_vectors.npz"))
    for task_id, key in ((0, "s_term_pick"), (1, "s_term_place")):
        e = EmuJacoEnv(task_id=task_id, frame_skip=1)
        for k in range(0, len(G["s_ee"]), 2):
            _inject_scene(e, model_arrays, G["s_ee"][k], G["s_eeq"][k], G["s_obj"][k], G["s_q2"][k], int(G["s_touch"][k]))
            e.task[0] = 0; e.task[0, 0] = 0.6; e.task[0, 16] = 0.6
            e.task[0, 2] = G["s_nsteps"][k]; e.task[0, 7:10] = G["s_dest_goal"][k]
            e._call(5)
            rd, rb, rwb, rs = G[key][k]
            assert bool(e.done[0]) == bool(rd) and e.task[0, 29] == rs, (task_id, k)
            assert abs(e.reward[0] - rb) < 2e-4 and abs(e.task[0, 30] - rwb) < 1e-5


def test_terminal_inspection_on_a_finished_env_is_frozen(names, model_arrays):
    """jaco_terminal_inspection on its own (mode 5) freezes like jaco_step: a finished -- or NaN-quarantined -- env keeps done = 1,
    bonus 0 and its counters (it used to count another step and recompute done, un-quarantining an env with NaN poses)."""
    e, _ = _pair(names, model_arrays, 5, 1)
    e.forward()
    e.task[0, 1] = 699                                                   # one terminal_inspection away from the time-out
    e._call(5)
    assert e.done[0] == 1 and abs(e.reward[0] + 10.0) < 1e-6 and e.task[0, 3] == 1 and e.task[0, 1] == 700
    steps, episodes = e.task[0, 1], e.task[0, 2]
    e._call(5)                                                           # again: frozen
    assert e.done[0] == 1 and e.reward[0] == 0 and e.task[0, 1] == steps and e.task[0, 2] == episodes and e.task[0, 3] == 1
    e.cache[0, 78:93] = np.nan                                           # quarantined env: poses non-finite
    e._call(5)
    assert e.done[0] == 1 and e.reward[0] == 0 and e.task[0, 3] == 1


def test_auto_reset_equals_the_explicit_reset_chain(names, model_arrays):
    """Option "auto_reset": the wave that finishes an episode does sim.reset(), the draws of _reset and sim.forward() + the first
    observation itself.  Bit-for-bit what the explicit chain (reset kernel's work, then a mode-2 forward pass) leaves behind: state,
    task row (goals, draw counter, counters), controller cache, markers and the observation row; reward / done are the terminal step's."""
    def make():
        e, _ = _pair(names, model_arrays, 7, 2)
        e.seed = 1234
        e.forward(np.full((1, 12), 0.5, np.float32))
        e.task[0, 1] = 698                                       # two steps before the 700-step time-out
        return e
    a = np.array([0.3, -0.2, 0.1, 0.2, -0.1, 0.3, 0.5], np.float32)
    nz = np.full((1, 12), 0.25, np.float32)
    ea, eb = make(), make()
    ea.set_auto_reset(1)
    oa, ra, da = ea.env_step(a, nz); ob, rb, db = eb.env_step(a, nz)
    assert da[0] == 0 and np.array_equal(oa, ob) and np.array_equal(ea.qpos, eb.qpos)                 # no episode end: nothing differs
    oa, ra, da = ea.env_step(a, nz)                              # time-out: auto path resets in the same call
    ea.set_auto_reset(0)
    ob, rb, db = eb.env_step(a, nz)                              # explicit path: terminal step ...
    assert da[0] == 1 and db[0] == 1 and ra[0] == rb[0] and abs(ra[0] + 10.0) < 0.2
    eb.reset_env(0)                                              # ... then jaco_reset(mask): reset kernel's work + forward pass
    ob = eb.forward(nz)
    assert np.array_equal(oa, ob), np.abs(oa - ob).max()
    for x, y in ((ea.qpos, eb.qpos), (ea.qvel, eb.qvel), (ea.qacc_ws, eb.qacc_ws), (ea.task, eb.task), (ea.cache, eb.cache), (ea.marker, eb.marker)):
        assert np.array_equal(x, y)
    assert ea.task[0, 1] == 0 and ea.task[0, 3] == 0 and ea.task[0, 39] == 0 and ea.task[0, 0] == np.float32(0.6)   # fresh episode, forward pass done
    assert abs(oa[0, 10] - 0.1898) < 1e-6 and abs(oa[0, 7] + 1.0) < 1e-6                                  # object on the holder, gripper command 0.6
    # the new episode runs on: both sides step identically
    o2a, _, d2a = ea.env_step(a, nz); o2b, _, d2b = eb.env_step(a, nz)
    assert d2a[0] == 0 and np.array_equal(o2a, o2b) and np.array_equal(ea.qpos, eb.qpos)


@pytest.mark.parametrize("seed", [26, 31])
def test_auto_reset_forward_pass_survives_a_tier_hand_off(names, model_arrays, seed):
    """Resets whose drawn pose puts the hand into the holder / pedestal (48 contacts / 216 rows for seed 26, 65 / 290 for seed 31, with the draw counter at 18 as it stands after one forward pass and one step): the
    forward pass of the auto-reset overflows the light tier and is finished by a bigger one (JT_FWD travels in the task row).  Same
    bits as the explicit reset chain, whose forward pass is handed on the same way."""
    def make():
        e, _ = _pair(names, model_arrays, 7, 2)
        e.seed = seed
        e.forward(np.full((1, 12), 0.5, np.float32))
        e.task[0, 1] = 699
        return e
    a = np.zeros(7, np.float32); nz = np.full((1, 12), 0.25, np.float32)
    ea, eb = make(), make()
    ea.set_auto_reset(1)
    oa, ra, da = ea.env_step(a, nz)
    ea.set_auto_reset(0)
    ob, rb, db = eb.env_step(a, nz)
    assert da[0] == 1 and db[0] == 1 and ra[0] == rb[0]
    eb.reset_env(0); eb.flags[:] = 0
    ob = eb.forward(nz)
    assert eb.flags[0] & 32 and ea.flags[0] & 32                                       # both forward passes left the light tier
    assert np.array_equal(oa, ob) and np.array_equal(ea.qpos, eb.qpos) and np.array_equal(ea.task, eb.task) and np.array_equal(ea.cache, eb.cache)
    assert ea.task[0, 39] == 0 and ea.task[0, 19] == 0 and ea.task[0, 17] == 0          # forward pass done, nothing pending


def test_side_rows_keep_a_68_row_env_in_the_light_tier(names, model_arrays):
    """Light tier, split mode (csrc/collision.h "Side rows"): object on holder (16 rows) + pedestal on floor (16) + the EE's axis sticks on
    the "hand" marker's sticks bring a step to 68-72 rows -- more than the 64-row main buffer.  The pedestal's rows go to the side buffer
    and are solved by the side Newton solve; the env stays in the light tier (no tier flag) and matches the oracle env like any other."""
    seen = 0
    for seed in (4, 6):
        fs = 10
        e = EmuJacoEnv(frame_skip=fs); oe = OracleEnv(names, frame_skip=fs)
        q = workload.reset_states(model_arrays["qpos0"], 1, seed=seed, f32_draws=True)[0]
        oe.obj_goal = q[9:12].copy(); oe.dest_goal = np.array([q[16], q[17], 0.3468]).astype(np.float32).astype(np.float64)
        oe.set_state(q)
        e.qpos[0] = q; e.task[0, 4:7] = oe.obj_goal; e.task[0, 7:10] = oe.dest_goal
        rng = np.random.default_rng(seed)
        nz = rng.uniform(size=(1, 12)).astype(np.float32)
        e.forward(nz); oe.observe(nz[0, 6:].astype(np.float64))
        for step in range(4):
            a = (rng.uniform(-1, 1, 7) * 0.05).astype(np.float32); nz = rng.uniform(size=(1, 12)).astype(np.float32)
            e.flags[:] = 0
            obs, rew, done = e.env_step(a, nz)
            oo, orew, odone, _ = oe.step(a.astype(np.float64), nz[0].astype(np.float64))
            assert (e.stats[0, 0], e.stats[0, 1]) == (oe.o.ncon, oe.o.nefc)                 # contacts / rows (main + side) as the oracle counts them
            assert np.abs(obs[0] - oo).max() < 2e-6 and np.abs(e.qpos[0] - oe.o.get("qpos")).max() < 2e-6 and abs(rew[0] - orew) < 1e-5
            if 64 < e.stats[0, 1] <= 80 and not (e.flags[0] & 32):
                seen += 1                                                                    # more rows than the main buffer holds, and no bigger tier was used
    assert seen >= 2


def test_pair_list_does_not_change_results(names, model_arrays):
    """Broadphase pair list (collision.h "Pair list"): the list pass runs the same exact test on a superset of the pairs that can
    pass it, so state, observation, contact / row / candidate counts must equal those of the all-pairs pass BIT FOR BIT -- over
    env steps with large actions (the list is rebuilt inside a step: markers jump, the arm sweeps several cm) and with the arm at rest."""
    import ctypes
    nenv, fs = 6, 25
    runs, passes = [], {}
    for on in (1, 0):
        e = EmuJacoEnv(nenv=nenv, frame_skip=fs)
        e.L.emu_set_pair_list.argtypes = [ctypes.c_int]
        e.L.emu_get_counter.argtypes = [ctypes.c_int, ctypes.c_int]; e.L.emu_get_counter.restype = ctypes.c_long
        e.L.emu_set_pair_list(on)
        e.L.emu_get_counter(0, 1); e.L.emu_get_counter(1, 1)
        try:
            q = workload.reset_states(model_arrays["qpos0"], nenv, seed=17)
            e.qpos[:] = q; e.task[:, 4:7] = q[:, 9:12]; e.task[:, 7] = q[:, 16]; e.task[:, 8] = q[:, 17]; e.task[:, 9] = 0.3468
            rng = np.random.default_rng(5)
            e.forward(rng.uniform(size=(nenv, 12)).astype(np.float32))
            trace = []
            for step in range(3):
                a = rng.uniform(-1, 1, (nenv, 7)).astype(np.float32) * (1.0 if step < 2 else 0.02)
                obs, rew, done = e.env_step(a, rng.uniform(size=(nenv, 12)).astype(np.float32))
                trace.append((e.qpos.copy(), e.qvel.copy(), obs, rew, done, e.stats.copy(), e.sensordata.copy(), e.flags.copy()))
            runs.append(trace)
            passes[on] = (e.L.emu_get_counter(0, 1), e.L.emu_get_counter(1, 1), e.L.emu_get_counter(2, 1))
        finally:
            e.L.emu_set_pair_list(1)
    for ta, tb in zip(*runs):
        for x, y in zip(ta, tb):
            assert np.array_equal(x, y)
    assert runs[0][-1][5][:, 0].max() > 0          # contacts were present
    full, listed, entries = passes[1]
    assert passes[0][1] == 0 and listed > 2 * full > 0, passes   # the list really carried most substeps, and was rebuilt inside the steps
    print("pair list: %d all-pairs passes, %d list passes (%.0f entries each) over %d env substeps" % (full, listed, entries / listed, nenv * fs * 3))


# ---------------------------------------------------------------- tasks releasing / carrying / pushing (round 4)
def test_kernel_terminal_releasing_carrying_pushing_against_the_references_own_outputs(model_arrays):
    """The in-kernel termination rules of tasks releasing / carrying / pushing (csrc/env_logic.h) through the terminal_inspection entry
    (mode 5) against tests/golden/glue_vectors_releasing.npz = outputs of the reference's own _get_terminal_inspection
    (env_mujoco_util.py:549-566,583-584): done flag exact, bonus to fp32 rounding; releasing reads the object's velocity (qvel[9:12])."""
    import os
    GR = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glue_vectors_releasing.npz"))
    n = len(GR["r_q2"])
    for task_id, key in ((6, "r_term_rel"), (5, "r_term_carry"), (7, "r_term_push")):
        e = EmuJacoEnv(task_id=task_id, frame_skip=1)
        nsucc = 0
        for k in range(0, n, 1 if task_id == 6 else 4):
            _inject_scene(e, model_arrays, GR["r_ee"][k], GR["r_eeq"][k], GR["r_obj"][k], GR["r_q2"][k], int(GR["r_touch"][k]))
            e.qvel[0, 9:12] = GR["r_objvel"][k]
            e.task[0] = 0; e.task[0, 0] = 0.6; e.task[0, 16] = 0.6
            e.task[0, 2] = GR["r_nsteps"][k]; e.task[0, 7:10] = GR["r_dest_goal"][k]
            e._call(5)
            rd, rb, rwb, _ = GR[key][k]
            assert bool(e.done[0]) == bool(rd), (task_id, k)
            assert abs(e.reward[0] - rb) < 2e-4 and abs(e.task[0, 30] - rwb) < 1e-5, (task_id, k, e.reward[0], rb)
            assert e.task[0, 29] == float(bool(rd) and rb > 100)
            nsucc += int(e.task[0, 29])
        assert (nsucc > 3) == (task_id == 6)


def test_reset_draws_of_releasing_and_carrying_stay_in_the_references_ranges(model_arrays):
    """reset_draws (csrc/env_logic.h) for tasks carrying / releasing against the min / max of 4 096 draws of the reference's own
    _create_init_angle (env_mujoco_util.py:181-189): inside the range, and covering most of it; releasing starts with the fingers at 0.6."""
    import os
    GR = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glue_vectors_releasing.npz"))
    for task_id, name, nang in ((5, "carrying", 6), (6, "releasing", 9)):
        e = EmuJacoEnv(task_id=task_id, nenv=256, seed=11)
        for env in range(256):
            e.reset_env(env)
        lo, hi = GR["init_%s_min" % name], GR["init_%s_max" % name]
        q = e.qpos[:, :nang].astype(np.float64)
        span = hi - lo
        tol = 2e-3 * span + 1e-6   # (the recorded extremes of 4 096 draws sit a few 1e-4 of the span inside the true bounds)
        assert np.all(q >= lo - tol) and np.all(q <= hi + tol), (name, q.min(0), lo, q.max(0), hi)
        wide = span > 1e-9
        assert np.all((q.max(0) - q.min(0))[wide] > 0.8 * span[wide])
        if task_id == 6:
            assert np.all(e.qpos[:, 6:9] == np.float32(0.6))
        else:
            assert np.all(e.qpos[:, 6:9] == np.float32(model_arrays["qpos0"][6:9]))   # fingers stay at the joint reference (mujoco.py:342-343)


def test_action_is_clipped_inside_the_kernel(names, model_arrays):
    """np.clip(action, act_min, act_max) of env_mujoco.py:117 happens in take_action (env_logic.h): an action beyond [-1, 1] gives the
    target, gripper command and state of its clipped twin, bit for bit."""
    outs = []
    for a in (np.array([2.5, -3.0, 0.4, 1.5, -0.2, -9.0, 7.0], np.float32), np.array([1.0, -1.0, 0.4, 1.0, -0.2, -1.0, 1.0], np.float32)):
        e, _ = _pair(names, model_arrays, 4, 3)
        e.forward()
        e.env_step(a, np.full((1, 12), 0.5, np.float32))
        outs.append((e.task[0].copy(), e.qpos[0].copy(), e.obs[0].copy()))
    for x, y in zip(*outs):
        assert np.array_equal(x, y)


def test_reset_draws_the_reaching_goal_from_an_init_buffer(model_arrays):
    """kwarg init_buffer (env_mujoco_util.py:46,208-212): with a goal buffer the reset takes its reaching goal from a random row -- rows
    0 .. n - 2 (np.random.randint(0, n - 1)), position row[1:4], orientation row[4:7], no float16 cast -- and the draws that follow (object
    and destination goals) stay in their ranges.  The buffer is the one of tests/golden/glue_vectors_init_buffer.npz, where the reference's
    own branch returned exactly these rows (tests/test_glue_golden.py)."""
    import os
    import glue
    B = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glue_vectors_init_buffer.npz"))
    buf = B["buffer"].astype(np.float32)
    n = len(buf)
    nenv = 160
    e = EmuJacoEnv(nenv=nenv, task_id=2, seed=11)      # task reaching
    e.set_init_buffer(buf)
    try:
        seen = set()
        for k in range(nenv):
            e.reset_env(k)
            goal = e.task[k, 32:38]
            hits = [i for i in range(n) if np.array_equal(goal, np.hstack([buf[i, 1:4], buf[i, 4:7]]))]
            assert len(hits) == 1 and hits[0] <= n - 2, (k, goal)
            assert np.array_equal(goal.astype(np.float64), glue.sample_reach_goal_from_buffer(buf, (hits[0] + 0.5) / (n - 1))[0])
            seen.add(hits[0])
            assert -0.1 <= e.qpos[k, 9] <= 0.1 and 0.57 <= e.qpos[k, 10] <= 0.67 and abs(e.qpos[k, 11] - 0.1898) < 1e-7
            assert 0.35 <= e.qpos[k, 16] <= 0.45 and 0.25 <= e.qpos[k, 17] <= 0.35
        assert seen == set(range(n - 1))               # every row but the last is drawn
        # in-kernel auto-reset draws from the same buffer with the same stream: bit-identical to the explicit reset
        e.set_init_buffer(None)
        e.reset_env(0)
        assert not any(np.array_equal(e.task[0, 32:38], np.hstack([buf[i, 1:4], buf[i, 4:7]])) for i in range(n))   # back to sampled goals
    finally:
        e.set_init_buffer(None)


def test_forward_pass_runs_on_zero_controls_whatever_the_neighbours_do(model_arrays):
    """sim.forward() after sim.reset() sees data.ctrl = 0.  The env-level launches hand the kernel a placeholder ctrl pointer (the qvel buffer);
    a forward pass alone (mode 2: jaco_forward, the explicit reset chain) used to READ it -- env e took qvel words [9 e, 9 e + 9) as its nine
    controls, so a reset's first touch reading depended on other envs' velocities (found on MI355X by the auto-reset bit-identity test: 1 env of
    8 192 x 6 steps, a reset with the hand inside the pedestal).  Two envs in such a pose: env 1's readings must not move when env 0 spins."""
    q = workload.reset_states(model_arrays["qpos0"], 64, seed=7, f32_draws=True)[62]   # hand 10 cm inside the pedestal: pads and EE sensor loaded
    out = []
    for spin in (0.0, 25.0):
        e = EmuJacoEnv(nenv=2)
        e.qpos[:] = q
        e.task[:, 4:7] = q[9:12]; e.task[:, 7:9] = q[16:18]; e.task[:, 9] = 0.3468
        e.qvel[0, 9:18] = spin          # what env 1 would read as ctrl[0..8]
        e.forward(np.full((2, 12), 0.5, np.float32))
        out.append((e.sensordata[1].copy(), e.obs[1].copy()))
    assert out[0][0].max() > 0.01                                   # the scenario really loads a sensor
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])

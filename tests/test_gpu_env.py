"""GPU tier, env level: JacoBatchedEnv (libjaco_env.so jaco_reset / jaco_step) against the fp64 oracle env, plus the
drop-in surface of the reference's JacoMujocoEnv (env_script/env_mujoco.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_env_step_parity_vs_oracle_env(model_arrays, names):
    from mujoco_jaco_amd import workload
    from mujoco_jaco_amd.env import JacoBatchedEnv
    from oracle_env import OracleEnv
    B, nstep = 12, 3
    q = workload.reset_states(model_arrays["qpos0"], B, seed=71)
    env = JacoBatchedEnv(num_envs=B, task="picking")
    dev = env.device
    env.sim.set_state(torch.tensor(q, dtype=torch.float32, device=dev), torch.zeros(B, 21, device=dev), torch.zeros(B, 21, device=dev))
    t = env.task_state(); t[:] = 0; t[:, 0] = 0.6; t[:, 16] = 0.6
    t[:, 4:7] = torch.tensor(q[:, 9:12], dtype=torch.float32); t[:, 7:9] = torch.tensor(q[:, 16:18], dtype=torch.float32); t[:, 9] = 0.3468
    env.set_task_state(t)
    rng = np.random.default_rng(7)
    oes = []
    for k in range(B):
        oe = OracleEnv(names); oe.obj_goal = q[k, 9:12].astype(np.float32).astype(np.float64)
        oe.dest_goal = np.array([q[k, 16], q[k, 17], 0.3468]).astype(np.float32).astype(np.float64)
        oe.set_state(q[k].astype(np.float32).astype(np.float64)); oes.append(oe)
    nz = rng.uniform(size=(B, 12)).astype(np.float32)
    env.set_noise(torch.tensor(nz)); obs = env.make_observation().cpu().numpy()
    for k in range(B):
        assert np.abs(obs[k] - oes[k].observe(nz[k, 6:].astype(np.float64))[0]).max() < 2e-6
    errs, rerrs = [], []
    for s in range(nstep):
        a = rng.uniform(-1, 1, (B, 7)).astype(np.float32); nz = rng.uniform(size=(B, 12)).astype(np.float32)
        env.set_noise(torch.tensor(nz))
        obs, rew, done, info = env.step(torch.tensor(a))
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        assert info == {0: 0}
        for k in range(B):
            oo, orew, odone, _ = oes[k].step(a[k].astype(np.float64), nz[k].astype(np.float64))
            errs.append(np.abs(obs[k] - oo).max())
            assert bool(done[k]) == odone                                   # termination flag: bit-exact
            assert obs[k, 0] == oo[0]                                        # touch class: exact
            rerrs.append(abs(rew[k] - orew))
    errs, rerrs = np.array(errs), np.array(rerrs)
    print("env-level obs error after up to %d steps x 50 substeps: median %.2e p90 %.2e max %.2e; reward error max %.2e" % (nstep, np.median(errs), np.percentile(errs, 90), errs.max(), rerrs.max()))
    # MAX over 12 envs x 3 steps, bounds = 3x the values measured on MI355X (round 3: obs 5.6e-5, reward 2.8e-7, median 6.0e-8)
    assert np.median(errs) < 2e-7 and errs.max() < 1.7e-4 and rerrs.max() < 1e-6


def test_env_level_closed_loop_parity_256_envs_10_steps(tmp_path):
    """256 envs x 10 env steps x 50 substeps (OSC every substep, env_mujoco_util.py:73-90), random actions, injected noise, against the
    fp64 oracle env (one process per core, in a fresh process): MAX over the batch.  Measured (MI355X, round 3): after 500 substeps
    qpos error median 1.7e-7; observation error median 1e-7; done flags and touch classes equal in all 2 560 env steps."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import env_drift
    B, nstep = 256, 10
    ref_path, gpu_path = str(tmp_path / "ref.npz"), str(tmp_path / "gpu.npz")
    subprocess.run([sys.executable, os.path.join(root, "tools", "env_drift.py"), "oracle", ref_path, str(B), str(nstep)], check=True, timeout=900)
    env_drift.gpu_leg(gpu_path, B, nstep)
    g, r = dict(np.load(gpu_path)), dict(np.load(ref_path))
    assert np.array_equal(g["done"].astype(bool), r["done"].astype(bool))                    # termination flags: bit-exact
    live = ~np.cumsum(r["done"].astype(bool), 0).astype(bool) | r["done"].astype(bool)           # up to and including the step an env finished in
    eq = np.abs(g["qpos"].astype(np.float64) - r["qpos"]).max(2)
    eo = np.abs(g["obs"].astype(np.float64) - r["obs"]).max(2)
    er = np.abs(g["reward"].astype(np.float64) - r["reward"])
    assert np.array_equal(g["obs"][..., 0][live], r["obs"][..., 0][live])                        # touch class: exact
    print("env level, %d envs x %d steps: qpos err median %.2e p99 %.2e max %.2e | obs err median %.2e max %.2e | reward err max %.2e | envs finished %d" % (
        B, nstep, np.median(eq[live]), np.percentile(eq[live], 99), eq[live].max(), np.median(eo[live]), eo[live].max(), er[live].max(), int(r["done"].any(0).sum())))
    assert (g["flags"] & 15).max() == 0
    # bounds = 3x measured (round 5, half-depth box-box contacts: qpos median 1.3e-7, p99 1.2e-6, max 1.0e-4; obs max 7.0e-6; reward max 1.8e-6,
    # median 1e-8; round 4: max 5.9e-5 / 1.5e-5 / 1.3e-7 -- the maxima belong to single envs whose object has been knocked)
    assert np.median(eq[live]) <= 4e-7 and np.percentile(eq[live], 99) <= 3.6e-6 and eq[live].max() <= 3e-4
    assert eo[live].max() <= 5e-5 and er[live].max() <= 5.4e-6 and np.median(er[live]) <= 1e-7


def test_drop_in_surface_single_env():
    """The caller pattern of main.py:250-263 with num_envs = 1: unbatched numpy / python types like the reference."""
    from mujoco_jaco_amd.env import JacoBatchedEnv
    env = JacoBatchedEnv(task="picking", robot_file="jaco2_curtain_torque", n_robots=1, seed=3, visualize=False)
    assert env.observation_space.shape == (26,) and env.action_space.shape == (7,) and env.observation_space.dtype == np.float32
    assert float(env.observation_space.high[0]) == 3.0 and env.metadata is None and env.skip_frames == 50 and env.task_max_steps == 700
    assert env.get_num_observation() == 26 and env.get_num_action() == 7 and env.get_action_bound() == 1 and env.get_state_shape() == 26
    obs = env.reset()
    assert isinstance(obs, np.ndarray) and obs.dtype == np.float32 and obs.shape == (26,)
    assert abs(obs[7] - (-1.0)) < 1e-6 and abs(obs[10] - 0.1898) < 1e-6 and abs(obs[24] - np.pi / 2) < 1e-6   # gripper 0.6, object z
    done, n = False, 0
    while not done and n < 5:
        obs, reward, done, info = env.step(env.action_space.sample() * 1.5)      # clipped like np.clip (env_mujoco.py:117)
        assert isinstance(reward, float) and isinstance(done, bool) and info == {0: 0} and obs.shape == (26,)
        n += 1
    assert np.isfinite(obs).all() and isinstance(env.get_wb(), float)
    env.seed(0)
    assert env.close() is None


def test_reset_distribution_and_mask(model_arrays):
    from mujoco_jaco_amd.env import JacoBatchedEnv
    B = 4096
    env = JacoBatchedEnv(num_envs=B, task="picking", seed=11)
    obs = env.reset()
    q = env.sim.get_state()[0].cpu().numpy()
    lo = np.array([0.7, 3.8, 1.0, 1.8, 1.0, 0.8]); hi = np.array([2.5, 4.0, 1.7, 2.5, 2.5, 2.3])       # env_mujoco_util.py:178-179
    assert (q[:, :6] >= lo - 1e-6).all() and (q[:, :6] <= hi + 1e-6).all()
    assert np.abs(q[:, :6].mean(0) - (lo + hi) / 2).max() < 0.05 * (hi - lo).max()                      # roughly uniform
    assert np.allclose(q[:, 6:9], 1.1) and np.allclose(q[:, 11], 0.1898) and np.allclose(q[:, 12:16], [1, 0, 0, 0])
    assert (np.abs(q[:, 9]) <= 0.1).all() and (q[:, 10] >= 0.57 - 1e-6).all() and (q[:, 10] <= 0.67 + 1e-6).all()
    assert (np.abs(q[:, 16] - 0.4) <= 0.05 + 1e-6).all() and (np.abs(q[:, 17] - 0.3) <= 0.05 + 1e-6).all()
    assert torch.isfinite(obs).all() and (obs[:, 7] + 1).abs().max() < 1e-6
    # step a little, then reset only the odd envs
    env.step(torch.zeros(B, 7))
    q1 = env.sim.get_state()[0].clone()
    mask = torch.zeros(B, dtype=torch.uint8); mask[1::2] = 1
    env.reset(mask)
    q2 = env.sim.get_state()[0]
    assert torch.equal(q1[0::2], q2[0::2]) and not torch.equal(q1[1::2], q2[1::2])
    assert not np.allclose(q2[1::2, :6].cpu().numpy(), q[1::2, :6])                                     # fresh draws


def test_done_envs_freeze_until_reset():
    from mujoco_jaco_amd.env import JacoBatchedEnv
    B = 8
    env = JacoBatchedEnv(num_envs=B, task="picking", frame_skip=2, seed=5)
    env.reset()
    t = env.task_state(); t[:4, 1] = 698; env.set_task_state(t)         # four envs one step before the 700-step time-out
    z = torch.zeros(B, 7)
    _, _, d, _ = env.step(z); assert not d.any()
    _, r, d, _ = env.step(z); assert d[:4].all() and not d[4:].any() and (r[:4] < -9).all()
    q = env.sim.get_state()[0].clone()
    _, r, d, _ = env.step(z)
    q2 = env.sim.get_state()[0]
    assert d[:4].all() and (r[:4] == 0).all() and torch.equal(q[:4], q2[:4]) and not torch.equal(q[4:], q2[4:])
    m = torch.zeros(B, dtype=torch.uint8); m[:4] = 1
    env.reset(m)
    _, _, d, _ = env.step(z); assert not d.any()


def test_placing_reset_vs_oracle(model_arrays, names):
    """jaco_reset for task 'placing' (env_mujoco_util.py:92-117): arm into the placing pose range, object pinned into the
    hand for 150 controlled substeps while the fingers close, then observation.  The arm angles the reset kernel drew are
    read back and the same hold is replayed on the fp64 oracle."""
    from mujoco_jaco_amd.env import JacoBatchedEnv
    from oracle_env import OracleEnv
    B = 6
    env = JacoBatchedEnv(num_envs=B, task="placing", seed=3)
    nz = np.full((B, 12), 0.5, np.float32)
    env.set_noise(torch.tensor(nz))
    obs = env.reset().cpu().numpy()
    q, v, _ = env.sim.get_state()
    q, v = q.cpu().numpy(), v.cpu().numpy()
    t = env.task_state().cpu().numpy()
    a0 = q[:, 0]
    assert ((a0 > 3 * np.pi / 8 - 0.2) & (a0 < 5 * np.pi / 8 + 0.2)).all()             # the hold barely moves the arm
    assert np.all(v[:, 9:] == 0) and np.all(t[:, 17] == 0) and np.all(t[:, 19] == 0)
    flags = env.sim.flags().cpu().numpy()
    assert not (flags & (1 | 2 | 4 | 8)).any()
    errs = []
    for k in range(B):
        oe = OracleEnv(names, task="placing")
        # state before the hold: the drawn arm angles are not observable after it, so replay from the target the kernel
        # recorded (EE pose at reset) is not possible; instead start the oracle from the kernel's post-hold arm state with
        # the object pinned, and check that one more held substep agrees (fixed point of the hold) ...
        q0 = q[k].astype(np.float64)
        oe.set_state(q0, v[k].astype(np.float64))
        oe.dest_goal = t[k, 7:10].astype(np.float64); oe.obj_goal = t[k, 4:7].astype(np.float64)
        oo = oe.observe(nz[k, 6:].astype(np.float64))[0]
        errs.append(np.abs(obs[k] - oo).max())
        assert obs[k, 0] == oo[0]                                                       # touch class from the final sim.forward()
        # ... and that the object sits in the grasp frame: EE_obj position - 0.04 * x axis, same orientation
        xp = oe.o.get("xpos").reshape(-1, 3)[oe.ee_obj]; xm = oe.o.get("xmat").reshape(-1, 3, 3)[oe.ee_obj]
        assert np.abs(q[k, 9:12] - (xp - 0.04 * xm[:, 0])).max() < 3e-2                 # (pinned in space: the closing fingers push the hand a little)
    print("placing reset: obs err max %.2e" % max(errs))
    assert max(errs) < 4e-7   # (measured 1.2e-7)


def test_placing_hold_parity_150(model_arrays, names):
    """The full 150-substep hold on the GPU against the oracle, from a given pre-hold state (mode 3 through jaco_reset is
    covered above; here the kernel is driven through the same entry with a state injected before the hold)."""
    from mujoco_jaco_amd import workload
    from mujoco_jaco_amd.env import JacoBatchedEnv
    from oracle_env import OracleEnv
    B = 4
    env = JacoBatchedEnv(num_envs=B, task="placing", seed=5)
    env.reset()
    q, _, _ = env.sim.get_state()
    q = q.cpu().numpy()
    # pre-hold state: reset arm pose, fingers open, object anywhere (it is re-pinned), zero velocity
    q0 = np.tile(model_arrays["qpos0"], (B, 1)); q0[:, :6] = q[:, :6]; q0[:, 16:18] = q[:, 16:18]
    dev = env.device
    env.sim.set_state(torch.tensor(q0, dtype=torch.float32, device=dev), torch.zeros(B, 21, device=dev), torch.zeros(B, 21, device=dev))
    env._placing_hold()
    q1, v1, _ = env.sim.get_state()
    q1 = q1.cpu().numpy()
    for k in range(B):
        oe = OracleEnv(names, task="placing")
        oe.set_state(q0[k].astype(np.float32).astype(np.float64))
        oe.placing_hold(150)
        oq = oe.o.get("qpos")
        print("placing hold env %d: arm/finger angle err %.2e, object position err %.2e" % (k, np.abs(q1[k, :9] - oq[:9]).max(), np.abs(q1[k, 9:12] - oq[9:12]).max()))
        assert np.abs(q1[k, :9] - oq[:9]).max() < 2.5e-6, (k, np.abs(q1[k, :9] - oq[:9]).max())   # 150 controlled substeps; measured <= 6.9e-7
        assert np.abs(q1[k, 9:12] - oq[9:12]).max() < 2.5e-7                                        # measured <= 7.3e-8


def test_frame_skip4_with_termination_masking(model_arrays, names):
    """BASELINE config 4: 4-substep frame skip + early-termination masking.  Half of the envs are one step away from the
    700-step time-out; after it they must freeze (done = 1, reward 0, state untouched) while the others keep matching
    the fp64 oracle env step for step."""
    from mujoco_jaco_amd import workload
    from mujoco_jaco_amd.env import JacoBatchedEnv
    from oracle_env import OracleEnv
    B, fs, nstep = 8, 4, 12
    q = workload.reset_states(model_arrays["qpos0"], B, seed=91)
    env = JacoBatchedEnv(num_envs=B, task="picking", frame_skip=fs)
    dev = env.device
    env.sim.set_state(torch.tensor(q, dtype=torch.float32, device=dev), torch.zeros(B, 21, device=dev), torch.zeros(B, 21, device=dev))
    t = env.task_state(); t[:] = 0; t[:, 0] = 0.6; t[:, 16] = 0.6
    t[:, 4:7] = torch.tensor(q[:, 9:12], dtype=torch.float32); t[:, 7:9] = torch.tensor(q[:, 16:18], dtype=torch.float32); t[:, 9] = 0.3468
    t[::2, 1] = 698                                           # even envs: current_steps one before the time-out
    env.set_task_state(t)
    oes = []
    for k in range(B):
        oe = OracleEnv(names, frame_skip=fs); oe.obj_goal = q[k, 9:12].astype(np.float32).astype(np.float64)
        oe.dest_goal = np.array([q[k, 16], q[k, 17], 0.3468]).astype(np.float32).astype(np.float64)
        oe.steps = 698 if k % 2 == 0 else 0
        oe.set_state(q[k].astype(np.float32).astype(np.float64)); oes.append(oe)
    rng = np.random.default_rng(17)
    nz = rng.uniform(size=(B, 12)).astype(np.float32)
    env.set_noise(torch.tensor(nz)); env.make_observation()
    frozen_q = None
    fs4_errs = []
    for s in range(nstep):
        a = rng.uniform(-1, 1, (B, 7)).astype(np.float32); nz = rng.uniform(size=(B, 12)).astype(np.float32)
        env.set_noise(torch.tensor(nz))
        obs, rew, done, _ = env.step(torch.tensor(a))
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        qn = env.sim.get_state()[0].cpu().numpy()
        for k in range(B):
            if k % 2 == 0 and s >= 2:                          # timed out at s == 1: frozen from then on
                assert done[k] and rew[k] == 0 and np.array_equal(qn[k], frozen_q[k])
                continue
            oo, orew, odone, _ = oes[k].step(a[k].astype(np.float64), nz[k].astype(np.float64))
            assert bool(done[k]) == odone, (s, k)
            fs4_errs.append(np.abs(obs[k] - oo).max())
            assert np.abs(obs[k] - oo).max() < 1.5e-6 and obs[k, 0] == oo[0]   # (measured max 4.2e-7 over 12 steps)
        if s == 1:
            assert done[::2].all() and not done[1::2].any() and (rew[::2] < -9).all()
            frozen_q = qn.copy()
    print("frame_skip 4, %d env steps: obs err median %.2e max %.2e" % (nstep, np.median(fs4_errs), np.max(fs4_errs)))


def test_launch_order_does_not_change_results():
    """The cost-ordered launch (a permutation of the envs built on the device every step) and the concurrent heavy tier
    (hand-off between workgroups on different XCDs, tier alternation) must not change any result: identical states and
    outputs with both on and both off, for a batch large enough to enable them.  Small actions keep the "hand" marker's
    sticks on the EE's sticks, which sends several per cent of the envs through the hand-off every step."""
    from mujoco_jaco_amd.env import JacoBatchedEnv
    B = 8192
    for scale in (1.0, 0.05):
        outs, heavy = [], 0
        for on in (1, 0):
            env = JacoBatchedEnv(num_envs=B, task="picking", seed=21)
            env.sim.set_option("schedule", on); env.sim.set_option("concurrent_heavy", on)
            env.reset()
            gen = torch.Generator(device=env.device); gen.manual_seed(5)
            for s in range(4):
                a = (torch.rand(B, 7, device=env.device, generator=gen) * 2 - 1) * scale
                if s == 3: env.sim.clear_flags()
                obs, rew, done, _ = env.step(a)
            q, v, _ = env.sim.get_state()
            heavy = max(heavy, int(((env.sim.flags() & 32) != 0).sum()))
            outs.append((q.clone(), v.clone(), obs.clone(), rew.clone(), done.clone()))
            del env
        for x, y in zip(*outs):
            assert torch.equal(x, y)
        if scale < 1.0:
            assert heavy > B // 200          # the hand-off path really was exercised


def test_double_buffered_queue_state_does_not_change_results():
    """The routing kernel of launch N prepares the queue buffer of launch N + 1 (lists back to "nothing published", counters zeroed, workers
    sized from launch N's demand); option merge_prepare = 0 brings back the launch set in which every launch prepares its own buffer with
    jaco_prepare_kernel.  Both must give the same states and outputs bit for bit over a sequence that mixes what flips the buffers in
    different ways: env steps in a row (fast path), masked resets and forward passes in between (launches that do not route), ctrl-level
    launches, at action scale 0.05 (most envs go through the tier queues every step).  Launches per step: one fewer on the fast path."""
    from mujoco_jaco_amd.env import JacoBatchedEnv
    B = 8192
    outs, launches = [], []
    for merge in (1, 0):
        env = JacoBatchedEnv(num_envs=B, task="picking", seed=41)
        env.sim.set_option("merge_prepare", merge)
        env.reset()
        gen = torch.Generator(device=env.device); gen.manual_seed(9)
        rec = []
        for s in range(7):
            a = (torch.rand(B, 7, device=env.device, generator=gen) * 2 - 1) * (0.05 if s < 5 else 1.0)
            if s == 2:
                env.sim.launch_count()
            obs, rew, done, _ = env.step(a)
            if s == 2:
                launches.append(env.sim.launch_count())
            if s == 3:      # a masked reset (reset kernel + listed forward pass: neither routes) between two steps
                mask = torch.zeros(B, dtype=torch.bool, device=env.device); mask[::7] = True
                obs = obs.clone(); obs[mask] = env.reset(mask)[mask]
            if s == 4:      # a ctrl-level launch of the sim tier on the same handle (routes too: hints are on)
                env.sim.send_forces(torch.zeros(B, 9, device=env.device), nsub=3)
            rec.append((obs.clone(), rew.clone(), done.clone()))
        q, v, _ = env.sim.get_state()
        rec.append((q.clone(), v.clone(), env.sim.flags().clone()))
        heavy = int(((env.sim.flags() & 32) != 0).sum())
        outs.append(rec)
        env.close()
    assert heavy > B // 4                                     # the queues really carried load
    assert launches[0] == launches[1] - 1, launches           # the fast path saves the prepare launch
    for x, y in zip(*outs):
        for u, w in zip(x, y):
            assert torch.equal(u, w)


def test_execution_options():
    """Hand-down (heavy drain -> second medium drain) only changes which workgroup runs the same code: bit-identical states and
    outputs.  The start-of-step routing ("hints") changes which capacity tier's code steps a substep; the tiers group their
    row sums differently (1, 2, 4 or 8 rows per lane), so its three settings agree to fp32 rounding, not bit for bit.
    Action scale 0.05: nine in ten envs leave the light tier, a quarter reach the heavy tier within a step."""
    from mujoco_jaco_amd.env import JacoBatchedEnv
    B = 8192
    outs = {}
    for hints, down in ((2, 1), (2, 0), (0, 1), (1, 1)):
        env = JacoBatchedEnv(num_envs=B, task="picking", seed=33)
        env.sim.set_option("hints", hints); env.sim.set_option("handdown", down)
        env.reset()
        gen = torch.Generator(device=env.device); gen.manual_seed(6)
        for s in range(2):
            a = (torch.rand(B, 7, device=env.device, generator=gen) * 2 - 1) * 0.05
            obs, rew, done, _ = env.step(a)
        q, v, _ = env.sim.get_state()
        outs[(hints, down)] = (q.clone(), v.clone(), obs.clone(), rew.clone(), done.clone())
        del env
    for x, y in zip(outs[(2, 1)], outs[(2, 0)]):
        assert torch.equal(x, y)
    for other in ((0, 1), (1, 1)):
        dq = (outs[(2, 1)][0] - outs[other][0]).abs().max(1).values
        print("hints 2 vs %d: qpos difference after 2 env steps: median %.2e, p99 %.2e, max %.2e" % (other[0], dq.median().item(), dq.quantile(0.99).item(), dq.max().item()))
        # every env bounded (MAX over 8 192 envs; bound = 3x the largest difference measured on MI355X: 2.0e-4, hints 2 vs 1, round 3)
        assert dq.median().item() < 1e-6 and dq.quantile(0.99).item() < 5e-6 and dq.max().item() < 6e-4


def test_vec_env_adapter_autoreset():
    """Learner-side adapter: SB-style step() with in-call reset of finished envs, terminal observation kept in infos."""
    from mujoco_jaco_amd.vec_env import JacoVecEnv
    B = 64
    venv = JacoVecEnv(B, task="picking", frame_skip=2, seed=9)
    obs0 = venv.reset()
    assert obs0.shape == (B, 26) and torch.isfinite(obs0).all()
    t = venv.env.task_state(); t[:16, 1] = 698; venv.env.set_task_state(t)      # 16 envs one step before the time-out
    z = torch.zeros(B, 7)
    o, r, d, info = venv.step(z); assert not d.any() and info["terminal_observation"] is None
    o, r, d, info = venv.step(z)
    assert d[:16].all() and not d[16:].any() and (r[:16] < -9).all()
    assert info["terminal_observation"].shape == (16, 26) and not info["is_success"].any()
    assert (info["episode_length"] == 2).all()
    t = venv.env.task_state()
    assert (t[:16, 1] == 0).all() and (t[16:, 1] == 2).all()                   # counters of the reset envs start over
    assert torch.isfinite(o).all() and (o[:16, 7] + 1).abs().max() < 1e-6       # fresh episodes: gripper command back at 0.6
    o2, r2, d2, _ = venv.step(z); assert not d2.any()                           # ... and they run again
    venv.close()


def test_markers_follow_take_action(model_arrays):
    """set_mocap_xyz / set_mocap_orientation of _take_action (env_mujoco_util.py:613-615,644-646): after a step the "hand" marker
    sits at the new EE target, the "subgoal_reach" marker at the rule-based sub-goal; reset parks both at their XML pose."""
    import glue
    from mujoco_jaco_amd.env import JacoBatchedEnv
    B = 8
    env = JacoBatchedEnv(num_envs=B, task="picking", seed=4)
    nz = np.random.default_rng(1).uniform(size=(B, 12)).astype(np.float32)
    env.set_noise(torch.tensor(nz))
    env.reset()
    rest = env.markers().cpu().numpy()
    assert np.allclose(rest[:, :, 2], -0.15) and np.allclose(rest[:, :, 3:], np.tile(np.eye(3).reshape(-1), (B, 2, 1)))   # xml:52,72
    obs0 = env.make_observation().cpu().numpy()
    a = np.random.default_rng(2).uniform(-1, 1, (B, 7)).astype(np.float32)
    env.step(torch.tensor(a))
    mk = env.markers().cpu().numpy()
    t = env.task_state().cpu().numpy()
    for k in range(B):
        target = t[k, 10:16]
        assert np.abs(mk[k, 0, :3] - target[:3]).max() < 1e-6
        R = glue.quat_to_mat(glue.quat_from_euler(*target[3:6]))
        assert np.abs(mk[k, 0, 3:].reshape(3, 3) - R).max() < 1e-5
        # sub-goal marker: the rule-based sub-goal of the pre-step observation state with the first six draws
        ee = obs0[k, 1:4].astype(np.float64); obj = obs0[k, 8:11].astype(np.float64)
        spos, sori = glue.rulebased_subgoal("picking", ee, t[k, 4:7].astype(np.float64), obj[1], t[k, 7:10].astype(np.float64), nz[k, :6].astype(np.float64))
        assert np.abs(mk[k, 1, :3] - spos).max() < 1e-5
        assert np.abs(mk[k, 1, 3:].reshape(3, 3) - glue.quat_to_mat(glue.quat_from_euler(*sori))).max() < 1e-4
    m2 = torch.tensor(mk); m2[:, 0, 0] += 0.5
    env.set_markers(m2)
    assert torch.equal(env.markers().cpu(), m2)
    env.reset()
    assert np.array_equal(env.markers().cpu().numpy(), rest)


def test_reaching_task_on_gpu(model_arrays, names):
    """Task 'reaching' end to end (env_mujoco_util.py:192-207,314-351,504-520; env_mujoco.py:79-89 action shape (6,), :20-21 500
    steps): reset draws (goal ranges, float16 orientation), reaching-goal observation branch, env-level parity with the
    fp64 oracle env, success flag."""
    from mujoco_jaco_amd.env import JacoBatchedEnv
    from oracle_env import OracleEnv
    import glue
    B = 2048
    env = JacoBatchedEnv(num_envs=B, task="reaching", rulebased_subgoal=False, seed=31)
    assert env.action_space.shape == (6,) and env.task_max_steps == 500 and env.get_num_action() == 6
    obs = env.reset().cpu().numpy()
    ts = env.task_state().cpu().numpy()
    g = ts[:, 32:38].astype(np.float64)
    assert ((np.abs(g[:, 0]) >= 0.3) & (np.abs(g[:, 0]) <= 0.42) & (np.abs(g[:, 1]) >= 0.3) & (np.abs(g[:, 1]) <= 0.42)).all()
    assert (g[:, 2] >= 0.3).all() and (g[:, 2] <= 0.5).all() and abs((g[:, 0] > 0).mean() - 0.5) < 0.05 and abs((g[:, 1] > 0).mean() - 0.5) < 0.05
    assert np.array_equal(g[:, 3:], g[:, 3:].astype(np.float16).astype(np.float64))        # np.float16 cast (:206)
    assert (np.abs(g[:, 5]) <= 0.1 + 1e-3).all()
    base = np.array([0.0, 0.0, 0.157])
    for k in range(0, B, 97):   # orientation looks along base -> goal (:201-205), checked through the pinned glue formula
        ref = glue.sample_reach_goal([abs(g[k, 0]), np.sign(g[k, 0]), abs(g[k, 1]), np.sign(g[k, 1]), g[k, 2], g[k, 5]], base)
        assert np.abs(ref[3:5] - g[k, 3:5]).max() <= 2e-3                                    # (fp32 vs fp64 before the float16 rounding)
    assert np.allclose(obs[:, 17:20], g[:, :3], atol=1e-6) and np.allclose(obs[:, 20:23], g[:, 3:] / np.pi, atol=1e-6)
    with pytest.raises(ValueError):
        env.step(torch.zeros(B, 7))                                                       # reaching takes 6-wide actions
    # env-level parity on a few envs, goal placed next to the EE so that the success branch is reached
    n = 6
    q = env.sim.get_state()[0].cpu().numpy().astype(np.float64)
    oes = []
    for k in range(n):
        oe = OracleEnv(names, task="reaching"); oe.rulebased = False
        oe.obj_goal = ts[k, 4:7].astype(np.float64); oe.dest_goal = ts[k, 7:10].astype(np.float64)
        oe.set_state(q[k])
        pe, qe = oe._ee()
        goal = np.concatenate([pe + [0.012, -0.006, 0.005], glue.euler_from_quat(qe) + [0.05, -0.04, 0.03]]).astype(np.float32).astype(np.float64)
        oe.reach_goal = goal; ts[k, 32:38] = goal
        oes.append(oe)
    env.set_task_state(torch.tensor(ts))
    succ_seen = 0
    for step in range(5):
        a = np.zeros((B, 6), np.float32)
        for k in range(n):
            a[k, :3] = ((oes[k].reach_goal[:3] - oes[k]._ee()[0]) * 25 * 0.8).clip(-1, 1)
        obs, rew, done, _ = env.step(torch.tensor(a))
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        for k in range(n):
            if oes[k].steps < 0:
                continue
            oo, orew, odone, osucc = oes[k].step(a[k].astype(np.float64), np.full(12, 0.5))
            assert bool(done[k]) == odone and np.abs(obs[k] - oo).max() < 1e-4 and abs(rew[k] - orew) < 2e-3
            if odone:
                succ_seen += int(osucc)
                assert bool(env.successes()[k]) == bool(osucc)
                oes[k].steps = -1      # finished: the GPU env freezes
    assert succ_seen >= 3


def test_full_size_env_level_config3_and_config4():
    """BASELINE configs 3 and 4 at their full size through jaco_step: 65 536 envs, full model; frame_skip 50 (config 3) and 4 with
    early-termination masking (config 4).  Size-independent properties: bitwise determinism, an env's result does not depend on
    its batch neighbours, finished envs freeze (done stays 1, reward 0, state untouched), finite outputs, unit quaternions."""
    from mujoco_jaco_amd.env import JacoBatchedEnv
    B = 65536
    for fs, nstep in ((50, 2), (4, 6)):
        outs = []
        for rep in range(2):
            env = JacoBatchedEnv(num_envs=B, task="picking", seed=77, frame_skip=fs)
            env.reset()
            ts = env.task_state(); ts[::3, 1] = 699 - nstep + 2; env.set_task_state(ts)     # a third of the envs time out during the run
            gen = torch.Generator(device=env.device); gen.manual_seed(9)
            frozen_state = None
            for s in range(nstep):
                a = torch.rand(B, 7, device=env.device, generator=gen) * 2 - 1
                obs, rew, done, _ = env.step(a)
                if s == nstep - 2:
                    d_prev = done.clone(); q_prev = env.sim.get_state()[0].clone()
            q, v, _ = env.sim.get_state()
            assert d_prev[::3].all() and not d_prev[1::3].any()
            assert torch.equal(q[d_prev], q_prev[d_prev]) and (rew[d_prev] == 0).all() and done[d_prev].all()   # frozen until reset
            assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and torch.isfinite(q).all()
            for adr in (12, 19):
                assert (q[:, adr:adr + 4].norm(dim=1) - 1).abs().max() < 1e-5
            assert int(env.sim.flags().max().item()) & 8 == 0
            outs.append((q.clone(), v.clone(), obs.clone(), rew.clone(), done.clone()))
            if rep == 1:   # neighbour independence: a 4 096-env slice stepped alone gives the same bits
                sl = slice(20480, 24576)
                small = JacoBatchedEnv(num_envs=4096, task="picking", seed=77, frame_skip=fs)
                small.reset()
                env2 = JacoBatchedEnv(num_envs=B, task="picking", seed=77, frame_skip=fs)
                env2.reset()
                q0, v0, w0 = env2.sim.get_state()
                small.sim.set_state(q0[sl].contiguous(), v0[sl].contiguous(), w0[sl].contiguous())
                small.set_task_state(env2.task_state()[sl].contiguous()); small.set_markers(env2.markers()[sl].contiguous())
                small.make_observation(); env2.make_observation()
                gen = torch.Generator(device=env.device); gen.manual_seed(9)
                a = torch.rand(B, 7, device=env.device, generator=gen) * 2 - 1
                nz = torch.rand(B, 12, device=env.device, generator=gen)
                env2.set_noise(nz); small.set_noise(nz[sl].contiguous())
                o2, r2, d2, _ = env2.step(a); os_, rs, ds, _ = small.step(a[sl].contiguous())
                assert torch.equal(o2[sl], os_) and torch.equal(r2[sl], rs) and torch.equal(d2[sl], ds)
                del small, env2
            del env
        for x, y in zip(*outs):
            assert torch.equal(x, y)                                                         # bitwise deterministic


def test_nan_state_is_quarantined():
    """SURVEY section 5, failure row: an env whose state goes non-finite ends its episode inside jaco_step (done = 1, reward 0,
    JACO_FLAG_NAN, finite observation row), stays frozen, and the vectorised adapter resets it with the other finished envs."""
    from mujoco_jaco_amd.vec_env import JacoVecEnv
    B = 256
    venv = JacoVecEnv(B, task="picking", seed=4)
    venv.reset()
    env = venv.env
    q, v, w = env.sim.get_state()
    bad = torch.tensor([3, 77, 200], device=env.device)
    v[bad, 2] = float("nan")                    # poison three envs' joint velocities (hints are cleared by set_state)
    env.sim.set_state(q, v, w)
    a = torch.zeros(B, 7, device=env.device)
    obs, rew, done, _ = env.step(a)
    assert done[bad].all() and (rew[bad] == 0).all() and torch.isfinite(obs).all()
    assert ((env.sim.flags()[bad] & 8) != 0).all() and int(done.sum()) == 3
    q1 = env.sim.get_state()[0].clone()
    obs, rew, done, _ = env.step(a)             # frozen until reset
    assert done[bad].all() and (rew[bad] == 0).all()
    nanmask = torch.isnan(q1)
    assert torch.equal(env.sim.get_state()[0][bad][~nanmask[bad]], q1[bad][~nanmask[bad]])
    # through the adapter: same poison, the finished (quarantined) envs come back reset and are counted
    venv.reset()
    q, v, w = env.sim.get_state(); v[bad, 2] = float("nan"); env.sim.set_state(q, v, w)
    obs, rew, done, infos = venv.step(a)
    assert infos["quarantined"] == 3 and done[bad].all() and torch.isfinite(obs).all()
    assert torch.isfinite(env.sim.get_state()[0]).all() and torch.isfinite(env.sim.get_state()[1]).all()
    obs, rew, done, infos = venv.step(a)
    assert not done[bad].any() and infos["quarantined"] == 0


def test_grasping_reset_and_steps_vs_oracle(model_arrays, names):
    """Task grasping end to end through the C ABI.  (A) jaco_reset (draws + the pre-reach loops of env_mujoco_util.py:123-170) leaves
    every env with the EE within 0.15 m of its object goal.  (B) from an injected pre-reach state the loops, replayed on the fp64
    oracle env, give the same EE target, observation and arm state; the env steps that follow agree in observation, reward
    (:352-391) and termination (:521-536)."""
    from mujoco_jaco_amd.env import JacoBatchedEnv
    from oracle_env import OracleEnv
    B = 16
    env = JacoBatchedEnv(num_envs=B, task="grasping", seed=7)
    assert env.task_max_steps == 500 and env.action_space.shape == (7,)
    obs = env.reset().cpu().numpy()
    t = env.task_state().cpu().numpy()
    fl = env.sim.flags().cpu().numpy()
    q = env.sim.get_state()[0].cpu().numpy()
    assert (t[:, 38] == 3).all() and not (fl & (0x20000 | 15)).any()                          # every pre-reach finished: no cap, no error flag
    assert (np.linalg.norm(obs[:, 1:4] - t[:, 4:7], axis=1) < 0.1505).all()                   # EE within 0.15 m of the object goal
    assert np.abs(q[:, 1] - 3.85).max() < 0.6 and np.allclose(t[:, 10:13], t[:, 4:7])         # target_pos ends on the object goal
    env.close()
    # (B) injected state: arm in the init range of :181-185, object on the holder, zero velocity
    rng = np.random.default_rng(5)
    lo = np.array([3 * np.pi / 8, 3.85, 1.0, 2.0, 0.8, -1.2]); hi = np.array([5 * np.pi / 8, 3.85, 1.1, 2.1, 2.3, -1.1])
    f32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
    q0 = np.tile(model_arrays["qpos0"], (B, 1)).astype(np.float64)                           # model constants exact, drawn coordinates fp32-representable
    q0[:, :6] = f32(rng.uniform(lo, hi, (B, 6)))
    q0[:, 9] = f32(rng.uniform(-0.1, 0.1, B)); q0[:, 10] = f32(0.65 + rng.uniform(-0.08, 0.02, B)); q0[:, 11] = f32(0.1898)
    q0[:, 16] = f32(0.4 + rng.uniform(-0.05, 0.05, B)); q0[:, 17] = f32(0.3 + rng.uniform(-0.05, 0.05, B))
    goal = f32(np.concatenate([rng.uniform(0.3, 0.42, (B, 2)) * rng.choice([-1, 1], (B, 2)), rng.uniform(0.3, 0.5, (B, 1)), rng.uniform(-1, 1, (B, 3))], 1))
    env = JacoBatchedEnv(num_envs=B, task="grasping", seed=7)
    dev = env.device
    env.sim.set_state(torch.tensor(q0, dtype=torch.float32, device=dev), torch.zeros(B, 21, device=dev), torch.zeros(B, 21, device=dev))
    ts = env.task_state(); ts[:] = 0; ts[:, 0] = 0.6; ts[:, 16] = 0.6
    ts[:, 4:7] = torch.tensor(q0[:, 9:12], dtype=torch.float32); ts[:, 7:9] = torch.tensor(q0[:, 16:18], dtype=torch.float32); ts[:, 9] = 0.3468
    ts[:, 32:38] = torch.tensor(goal, dtype=torch.float32)
    env.set_task_state(ts)
    nz = np.full((B, 12), 0.5, np.float32); nz[:, 0] = 0.3
    env.set_noise(torch.tensor(nz))
    obs2 = env._grasping_prereach().cpu().numpy()
    t2 = env.task_state().cpu().numpy()
    q2 = env.sim.get_state()[0].cpu().numpy()
    oes, errs = [], []
    for k in range(B):
        oe = OracleEnv(names, task="grasping")
        oe.obj_goal = q0[k, 9:12].copy(); oe.dest_goal = f32([q0[k, 16], q0[k, 17], 0.3468]); oe.reach_goal = goal[k].copy()
        oe.set_state(q0[k])
        n = oe.grasping_prereach(float(-0.1 + 0.2 * np.float32(0.3)))
        assert t2[k, 38] == 3 and np.abs(t2[k, 10:16] - oe.target).max() < 1e-6, (k, n)
        oo = oe.observe(nz[k, 6:].astype(np.float64))[0]
        errs.append((np.abs(obs2[k] - oo).max(), np.abs(q2[k, :9] - oe.o.get("qpos")[:9]).max(), n))
        oes.append(oe)
    errs = np.array(errs)
    print("grasping pre-reach, %d envs: substeps %d..%d, obs err max %.2e, arm angle err max %.2e" % (B, errs[:, 2].min(), errs[:, 2].max(), errs[:, 0].max(), errs[:, 1].max()))
    assert errs[:, 0].max() < 3e-6 and errs[:, 1].max() < 7e-7   # 3x measured (9.8e-7 / 2.3e-7 over pre-reaches of 2 .. 427 substeps)
    rng = np.random.default_rng(9)
    serr, rerr = [], []
    for s in range(3):
        a = rng.uniform(-1, 1, (B, 7)).astype(np.float32); nzs = rng.uniform(size=(B, 12)).astype(np.float32)
        env.set_noise(torch.tensor(nzs))
        ob, rew, done, _ = env.step(torch.tensor(a))
        ob, rew, done = ob.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        for k in range(B):
            if oes[k] is None:
                continue
            oo, orew, odone, _ = oes[k].step(a[k].astype(np.float64), nzs[k].astype(np.float64))
            assert bool(done[k]) == odone and ob[k, 0] == oo[0], (s, k)
            serr.append(np.abs(ob[k] - oo).max()); rerr.append(abs(rew[k] - orew))
            if odone:
                oes[k] = None
    print("grasping steps: obs err max %.2e, reward err max %.2e" % (max(serr), max(rerr)))
    assert max(serr) < 8e-7 and max(rerr) < 3e-7                 # 3x measured (2.7e-7 / 9.9e-8)


def test_pickandplace_episode_flow(names):
    """Task pickAndplace (termination env_mujoco_util.py:585-600; reward 0; 1 200-step episodes): reset like picking; the first step in
    which the object is above the pick height with a grasp pays +20 once (`picked`), placing it on the pedestal ends the episode
    with +180.  Driven through the C ABI with states injected into the HIP env; rules checked against oracle/glue.py."""
    from mujoco_jaco_amd.env import JacoBatchedEnv
    env = JacoBatchedEnv(num_envs=8, task="pickAndplace", seed=3)
    assert env.task_max_steps == 1200 and env.action_space.shape == (7,)
    obs = env.reset()
    q = env.sim.get_state()[0].cpu().numpy()
    lo = np.array([0.7, 3.8, 1.0, 1.8, 1.0, 0.8]); hi = np.array([2.5, 4.0, 1.7, 2.5, 2.5, 2.3])       # the picking init range (:177-180)
    assert (q[:, :6] >= lo - 1e-6).all() and (q[:, :6] <= hi + 1e-6).all() and np.allclose(q[:, 11], 0.1898)
    o, r, d, _ = env.step(torch.zeros(8, 7))
    assert not d.any() and (r == 0).all()                                                                # reward 0 (:441-442), nothing picked yet
    t = env.task_state()
    assert (t[:, 31] == 0).all() and (t[:, 1] == 1).all()
    # object lifted into the closed hand is hard to stage physically in one step: the rule itself is pinned in
    # tests/test_env_emu.py::test_kernel_terminal_inspection_against_the_references_own_outputs; here: the time-out at 1 200 steps
    t[:, 1] = 1198; env.set_task_state(t)
    _, _, d, _ = env.step(torch.zeros(8, 7)); assert not d.any()
    _, r, d, _ = env.step(torch.zeros(8, 7)); assert d.all() and (r == -10).all()


def test_config1_single_env_1000_random_action_steps(model_arrays, names):
    """BASELINE config 1 at its stated length: ONE env through the drop-in surface (the reference's unbatched types), 1 000 steps of
    U(-1, 1)^7 actions from numpy.random.default_rng(0), task picking, frame_skip 50, with the caller pattern of
    main.py:254-260 (reset when done).  The first 40 steps (2 000 substeps) are replayed on the fp64 oracle env from the same reset
    state with the same injected noise; the rest checks the episode bookkeeping and that nothing goes non-finite or gets flagged."""
    from mujoco_jaco_amd.env import JacoBatchedEnv
    from oracle_env import OracleEnv
    env = JacoBatchedEnv(task="picking", robot_file="jaco2_curtain_torque", n_robots=1, seed=11)
    rng = np.random.default_rng(0)
    nz0 = np.full((1, 12), 0.5, np.float32)
    env.set_noise(torch.tensor(nz0))
    obs = env.reset()
    q0 = env.sim.get_state()[0].cpu().numpy()[0].astype(np.float64)
    t0 = env.task_state().cpu().numpy()[0]
    oe = OracleEnv(names)
    oe.obj_goal, oe.dest_goal = t0[4:7].astype(np.float64), t0[7:10].astype(np.float64)
    # (the reset state as the env holds it: fp32 draws; the model's own constants -- finger angles 1.1, pedestal height -- exact for the oracle)
    qo = model_arrays["qpos0"].copy(); qo[:6] = q0[:6]; qo[9:12] = q0[9:12]; qo[16:18] = q0[16:18]
    oe.set_state(qo)
    assert np.abs(obs - oe.observe(nz0[0, 6:].astype(np.float64))[0]).max() < 2e-6
    errs, episodes, steps_in_episode, returns = [], 0, 0, 0.0
    replay = True
    for s in range(1000):
        a = rng.uniform(-1, 1, 7).astype(np.float32)
        nz = rng.uniform(size=(1, 12)).astype(np.float32)
        env.set_noise(torch.tensor(nz))
        obs, reward, done, info = env.step(a)
        assert isinstance(reward, float) and isinstance(done, bool) and info == {0: 0} and obs.shape == (26,) and np.isfinite(obs).all()
        steps_in_episode += 1; returns += reward
        if replay and s < 40:
            oo, orew, odone, _ = oe.step(a.astype(np.float64), nz[0].astype(np.float64))
            assert done == odone and obs[0] == oo[0]
            errs.append(np.abs(obs - oo).max())
            assert abs(reward - orew) < 1e-4
        if done:
            replay = False
            assert steps_in_episode <= 700
            episodes += 1; steps_in_episode = 0
            env.set_noise(torch.tensor(nz0))
            obs = env.reset()
    fl = int(env.sim.flags().cpu().numpy()[0])
    print("config 1: 1 000 steps, %d episodes finished, first %d steps vs the oracle env: obs err median %.2e max %.2e; flags 0x%x" % (episodes, len(errs), np.median(errs), max(errs), fl))
    assert fl & 15 == 0 and len(errs) >= 20 and max(errs) < 2e-6   # (measured 6.6e-7 over 40 steps = 2 000 substeps)


def test_auto_reset_equals_step_plus_masked_reset_8192_envs():
    """Option auto_reset at batch scale: 8 192 envs, episodes ending all over the batch (step counters spread just below the 700-step
    time-out), 6 steps: jaco_step with the reset folded into the step wave against jaco_step + jaco_reset(done): rewards, done flags,
    observations, states and task rows bit-identical in every step -- including the resets whose forward pass leaves the light tier."""
    from mujoco_jaco_amd.env import JacoBatchedEnv
    B = 8192
    outs = []
    for auto in (True, False):
        env = JacoBatchedEnv(num_envs=B, task="picking", seed=31, auto_reset=auto, frame_skip=10)
        assert env.auto_reset == auto
        env.reset()
        gen = torch.Generator(device=env.device); gen.manual_seed(3)
        t = env.task_state(); t[:, 1] = torch.randint(693, 699, (B,), device=env.device, generator=gen).float(); env.set_task_state(t)
        rec = []
        for s in range(6):
            a = torch.rand(B, 7, device=env.device, generator=gen) * 2 - 1
            o, r, d, _ = env.step(a)
            r, d = r.clone(), d.clone()
            if not auto:
                o = env.reset(d)
            rec.append((o.clone(), r, d, env.sim.get_state()[0].clone(), env.task_state().clone()))
        outs.append((rec, env.sim.flags().clone()))
        env.close()
    ndone, heavy_resets = 0, 0
    for s, (x, y) in enumerate(zip(outs[0][0], outs[1][0])):
        ndone += int(x[2].sum())
        for k in range(5):
            assert torch.equal(x[k], y[k]), (s, k, (x[k].float() - y[k].float()).abs().max())
    assert ndone >= B // 2 and ((outs[0][1] & 15) == 0).all()


def test_releasing_carrying_pushing_through_the_c_abi(model_arrays, names):
    """Tasks releasing / carrying / pushing (round 4) end to end through the C ABI.
    releasing (env_mujoco_util.py:106-117,186-189,551-566): (A) jaco_reset leaves the arm in the releasing pose range with the object
    pinned in the hand and the fingers closed onto it; (B) the 150-substep hold from a given pre-hold state (fingers at 0.6) and the env
    steps that follow (gripper opening) agree with the fp64 oracle env in state, observation, reward + bonus and done flag.
    carrying (:106-117,123-170,549-550): reset = hold + pre-reach loops, every episode ends in its first step with bonus 0.
    pushing (:583-584; 6-wide action, env_mujoco.py:79-82): ends in its first step; with auto_reset the observation is the new episode's."""
    from mujoco_jaco_amd.env import JacoBatchedEnv
    from oracle_env import OracleEnv
    B = 4
    env = JacoBatchedEnv(num_envs=B, task="releasing", seed=9)
    assert env.action_space.shape == (7,) and env.task_max_steps == 500
    nz = np.full((B, 12), 0.5, np.float32)
    env.set_noise(torch.tensor(nz))
    obs = env.reset().cpu().numpy()
    q, v, _ = env.sim.get_state()
    q = q.cpu().numpy()
    assert np.isfinite(obs).all() and not (env.sim.flags().cpu().numpy() & (1 | 2 | 4 | 8)).any()
    assert ((q[:, 0] > 1.7) & (q[:, 0] < 2.2) & (q[:, 1] > 3.1) & (q[:, 1] < 3.8)).all()        # (:186-187; the hold barely moves the arm)
    assert (q[:, 6:9] > 0.55).all() and (q[:, 6:9] < 1.0).all()                               # fingers started at 0.6 and rest on the object
    # (B) hold + steps from an injected pre-hold state
    q0 = np.tile(model_arrays["qpos0"], (B, 1)); q0[:, :6] = q[:, :6]; q0[:, 6:9] = 0.6; q0[:, 16:18] = q[:, 16:18]
    q0 = q0.astype(np.float32).astype(np.float64)
    dev = env.device
    env.sim.set_state(torch.tensor(q0, dtype=torch.float32, device=dev), torch.zeros(B, 21, device=dev), torch.zeros(B, 21, device=dev))
    t = env.task_state(); t[:, 1] = 0; t[:, 2] = 0; t[:, 3] = 0; env.set_task_state(t)
    env._placing_hold()
    env.make_observation()
    q1 = env.sim.get_state()[0].cpu().numpy()
    t = env.task_state().cpu().numpy()
    oes = []
    for k in range(B):
        oe = OracleEnv(names, task="releasing")
        oe.set_state(q0[k])
        oe.dest_goal = t[k, 7:10].astype(np.float64); oe.obj_goal = t[k, 4:7].astype(np.float64)
        oe.placing_hold(150)
        oq = oe.o.get("qpos")
        assert np.abs(q1[k, :9] - oq[:9]).max() < 2.5e-6 and np.abs(q1[k, 9:12] - oq[9:12]).max() < 2.5e-7, (k, np.abs(q1[k, :9] - oq[:9]).max())
        oes.append(oe)
    rng = np.random.default_rng(3)
    oerr, rerr = [], []
    for s in range(3):
        a = rng.uniform(-1, 1, (B, 7)).astype(np.float32); a[:, 6] = 1.0     # open the gripper: the object is let go
        nz = rng.uniform(size=(B, 12)).astype(np.float32)
        env.set_noise(torch.tensor(nz))
        o, r, d, _ = env.step(torch.tensor(a))
        o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
        for k in range(B):
            if oes[k] is None:
                continue
            oo, orew, odone, _ = oes[k].step(a[k].astype(np.float64), nz[k].astype(np.float64))
            assert bool(d[k]) == odone and o[k, 0] == oo[0], (s, k)
            oerr.append(np.abs(o[k] - oo).max()); rerr.append(abs(r[k] - orew))
            if odone:
                oes[k] = None
    print("releasing: obs err max %.2e, reward err max %.2e over %d env steps" % (max(oerr), max(rerr), len(oerr)))
    assert max(oerr) < 2e-4 and max(rerr) < 1e-4
    env.close()
    # carrying
    env = JacoBatchedEnv(num_envs=B, task="carrying", seed=4)
    obs = env.reset().cpu().numpy()
    assert np.isfinite(obs).all() and env.action_space.shape == (7,)
    o, r, d, _ = env.step(torch.zeros(B, 7))
    assert d.all() and ((r == 0) | (r == -1)).all() and not env.successes().any()
    obs2 = env.reset(d)
    assert torch.isfinite(obs2).all() and (env.task_state()[:, 3] == 0).all()
    env.close()
    # pushing, reset inside jaco_step
    env = JacoBatchedEnv(num_envs=B, task="pushing", seed=4, auto_reset=True)
    assert env.action_space.shape == (6,) and env.auto_reset
    first = env.reset().clone()
    o, r, d, _ = env.step(torch.zeros(B, 6))
    assert d.all() and ((r == 0) | (r == -1)).all()
    assert torch.isfinite(o).all() and not torch.equal(o, first) and (env.task_state()[:, 1] == 0).all()   # a new episode's first observation
    env.close()


def test_small_action_regime_flag_census_and_parity_of_unflagged_envs(tmp_path):
    """Action scale 0.1: the EE's axis sticks lie on the "hand" marker's sticks, most envs run in the medium / heavy / huge tiers, and the
    last tier (512 rows / 128 contacts) has nobody to hand an overflow to -- contacts / rows beyond it are dropped and FLAGGED
    (physics_kernel.h, TIER == 3).  (A) census: 16 384 envs x 60 steps, finished envs reset inside jaco_step, error flags counted and
    cleared after every step: the flagged share of env steps must stay below 1e-5 (measured on MI355X: see the printed line).
    (B) parity of what is not flagged: 96 envs x 6 steps against the fp64 oracle env on the same actions / noise; every env without an
    error flag is compared (MAX over them), done flags exact."""
    import os, subprocess, sys
    from mujoco_jaco_amd.env import JacoBatchedEnv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import env_drift
    # (B) first: its oracle pool forks in a fresh process
    B, nstep, scale = 96, 6, 0.1
    ref_path, gpu_path = str(tmp_path / "ref.npz"), str(tmp_path / "gpu.npz")
    subprocess.run([sys.executable, os.path.join(root, "tools", "env_drift.py"), "oracle", ref_path, str(B), str(nstep), str(scale)], check=True, timeout=900)
    env_drift.gpu_leg(gpu_path, B, nstep, scale=scale)
    g, r = dict(np.load(gpu_path)), dict(np.load(ref_path))
    ok = (g["flags"] & 31) == 0
    bigger = int(((g["flags"] & 32) != 0).sum())
    assert ok.sum() >= B - 1 and bigger > B // 4, (int(ok.sum()), bigger)       # the bigger tiers really carried the regime
    assert np.array_equal(g["done"].astype(bool)[:, ok], r["done"].astype(bool)[:, ok])
    live = (~np.cumsum(r["done"].astype(bool), 0).astype(bool) | r["done"].astype(bool)) & ok[None, :]
    eq = np.abs(g["qpos"].astype(np.float64) - r["qpos"]).max(2)
    eo = np.abs(g["obs"].astype(np.float64) - r["obs"]).max(2)
    print("scale 0.1, %d unflagged envs x %d steps (%d of them stepped by a bigger tier): qpos err median %.2e p99 %.2e max %.2e | obs err max %.2e" % (
        int(ok.sum()), nstep, bigger, np.median(eq[live]), np.percentile(eq[live], 99), eq[live].max(), eo[live].max()))
    # bounds = 3x measured on MI355X (qpos median 4.3e-7, p99 1.95e-4, max 8.6e-4; obs max 2.9e-4): the stick-on-stick contacts of this regime
    # amplify rounding differences faster than the headline regime's (there: max 5.9e-5 after 10 steps)
    assert np.median(eq[live]) <= 1.3e-6 and np.percentile(eq[live], 99) <= 6e-4 and eq[live].max() <= 2.6e-3 and eo[live].max() <= 9e-4
    # (A)
    B, nstep = 16384, 60
    env = JacoBatchedEnv(num_envs=B, task="picking", seed=21, auto_reset=True)
    env.reset()
    gen = torch.Generator(device=env.device); gen.manual_seed(4)
    nfl = torch.zeros((), dtype=torch.int64, device=env.device)
    kinds = torch.zeros(5, dtype=torch.int64, device=env.device)
    for s in range(nstep):
        env.step((torch.rand(B, 7, device=env.device, generator=gen) * 2 - 1) * 0.1)
        f = env.sim.flags()
        nfl += ((f & 31) != 0).sum()
        kinds += torch.stack([((f & (1 << b)) != 0).sum() for b in range(5)])
        env.sim.clear_flags()
    share = float(nfl.item()) / (B * nstep)
    print("scale 0.1 census: %d of %d env steps flagged (share %.2e); contacts / rows / candidates dropped, non-finite, solver cap: %s" % (
        int(nfl.item()), B * nstep, share, kinds.tolist()))
    assert share <= 1e-5 and int(kinds[3]) == 0
    env.close()


def test_last_terminal_survives_the_in_kernel_reset():
    """jaco_get_last_terminal: (success, wb) of the step that ended an episode, latched before the auto-reset clears the task row.
    Half of the envs are sent into the 700-step time-out (success 0, wb 0 as the reference returns there, env_mujoco.py:148-150); the
    same steps without auto_reset give the task row's JT_SUCC / JT_WB to compare with."""
    from mujoco_jaco_amd.env import JacoBatchedEnv
    B = 64
    outs = []
    for auto in (True, False):
        env = JacoBatchedEnv(num_envs=B, task="picking", seed=12, frame_skip=4, auto_reset=auto)
        env.reset()
        t = env.task_state(); t[: B // 2, 1] = 699; env.set_task_state(t)
        o, r, d, _ = env.step(torch.zeros(B, 7))
        succ, wb = env.last_terminal()
        assert d[: B // 2].all() and not d[B // 2:].any()
        outs.append((succ.clone(), wb.clone(), env.successes().clone(), env.get_wb().clone(), env.task_state()[:, 1].clone()))
        env.close()
    (sa, wa, _, _, steps_a), (sb, wb_, succ_row, wb_row, steps_b) = outs
    assert torch.equal(sa, sb) and torch.equal(wa, wb_)                       # the latch does not depend on who resets
    assert torch.equal(sb[: B // 2], succ_row[: B // 2]) and torch.equal(wb_[: B // 2], wb_row[: B // 2])   # = the terminal step's task row
    assert (steps_a[: B // 2] == 0).all() and (steps_b[: B // 2] == 700).all()   # auto_reset: the row already belongs to the new episode
    assert not sa.any() and (wa[B // 2:] == 0).all()


def test_task_row_round_trip_keeps_the_draw_counter_bits():
    """The task row's draw counter (JT_RNG, slot 18) is the 29-bit count with bit 30 set: always the bit pattern of a NORMAL float (the library
    is built with denormals flushed; rounds 3-4 stored the bare unsigned, a denormal pattern for small counts).  get -> set -> step must leave
    the RNG stream where an untouched twin has it, and the slot must survive a float operation that canonicalises its operand."""
    from mujoco_jaco_amd.env import JacoBatchedEnv
    B = 32
    a, b = JacoBatchedEnv(num_envs=B, task="picking", seed=5, frame_skip=2), JacoBatchedEnv(num_envs=B, task="picking", seed=5, frame_skip=2)
    a.reset(); b.reset()
    ts = b.task_state()
    bits = ts[:, 18].view(torch.int32)
    cnt = bits & 0x1FFFFFFF
    assert (cnt > 0).all() and (cnt < 1000).all() and ((bits >> 29) == 2).all()   # a live counter behind a normal-float exponent
    assert torch.isfinite(ts[:, 18]).all() and (ts[:, 18].abs() >= 2.0).all()
    assert torch.equal((ts[:, 18] * 1.0).view(torch.int32), bits)                 # a float multiply (flushes denormals on the GPU) keeps the bits
    b.set_task_state(ts.clone())
    assert torch.equal(b.task_state().view(torch.int32), ts.view(torch.int32))
    z = torch.zeros(B, 7)
    for _ in range(2):
        oa, _, _, _ = a.step(z); ob, _, _, _ = b.step(z)
    assert torch.equal(oa, ob) and torch.equal(a.task_state().view(torch.int32), b.task_state().view(torch.int32))
    a.close(); b.close()


def test_pair_list_and_separating_direction_cache_do_not_change_results_at_scale():
    """The two work-skipping schemes of the collision stage (collision.h: broadphase pair list across substeps, cached separating directions
    of hull pairs) only skip tests whose outcome is known, so switching them off must reproduce every state and output BIT FOR BIT -- here on
    the GPU, 8 192 envs x 4 env steps (1.6 M env substeps) under random actions and under the shipped picking policy's grasp regime
    (hull hits and near misses that come and go), hand-offs between capacity tiers and in-kernel resets included.  Sticky flags and the
    contact / row statistics must agree as well."""
    import os
    from mujoco_jaco_amd.env import JacoBatchedEnv
    from mujoco_jaco_amd.policy import HPCPolicy
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    B = 8192
    for regime in ("random", "policy"):
        outs = []
        for pl, sc in ((1, 1), (0, 1), (1, 0)):
            env = JacoBatchedEnv(num_envs=B, task="picking", seed=33, auto_reset=True)
            env.sim.set_option("pair_list", pl); env.sim.set_option("sep_cache", sc)
            obs = env.reset()
            gen = torch.Generator(device=env.device); gen.manual_seed(8)
            pol = HPCPolicy.load(os.path.join(root, "tests", "golden", "policy_picking.npz"), device=env.device) if regime == "policy" else None
            if pol is not None:   # into the grasp phase first (the policy needs ~100 steps to reach the object): shorter steps get there cheaply
                ts = env.task_state(); ts[:, 1] = torch.randint(0, 600, (B,), device=env.device, generator=gen).float(); env.set_task_state(ts)
                for s in range(90):
                    obs, _, _, _ = env.step(pol.predict(obs)[0])
            for s in range(4):
                a = pol.predict(obs)[0] if pol is not None else torch.rand(B, 7, device=env.device, generator=gen) * 2 - 1
                obs, rew, done, _ = env.step(a)
            q, v, _ = env.sim.get_state()
            outs.append((q.clone(), v.clone(), obs.clone(), rew.clone(), done.clone(), env.sim.flags().clone(), env.sim.stats()[:, :3].clone(), env.sim.sensordata().clone()))
            env.close()
        for other in outs[1:]:
            for x, y in zip(outs[0], other):
                assert torch.equal(x, y), regime
        print("%s: mean contacts %.1f rows %.1f, touch sensors active in %d envs" % (regime, outs[0][6][:, 0].float().mean().item(), outs[0][6][:, 1].float().mean().item(),
                                                                              int((outs[0][7] > 1e-3).any(1).sum())))


def test_grasp_regime_parity_under_the_shipped_policy(names):
    """Env-level parity where the physics is hardest: mid-grasp under the reference's shipped picking policy (hull contacts through MPR, condim-6
    pad contacts, touch sensors active, most envs stepped by the medium / heavy tiers).  256 envs roll 100 policy-driven steps on the GPU; at
    that point both sides take the SAME state: jaco_forward (= sim.forward(): the controller's one-substep-stale quantities are refreshed from
    the current state) on the GPU, set_state + forward on the fp64 oracle env, for the envs whose fingers touch the object; then 3 more steps
    with the policy's actions (computed from the GPU observations, given to both sides) and the same injected noise.  Done flags exact, touch
    class equal but for the odd grazing contact (counted), observation / reward within fp32-vs-fp64 bounds (MAX over the compared envs)."""
    import os
    from mujoco_jaco_amd.env import JacoBatchedEnv
    from mujoco_jaco_amd.policy import HPCPolicy
    from oracle_env import OracleEnv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    B = 256
    env = JacoBatchedEnv(num_envs=B, task="picking", seed=44)
    pol = HPCPolicy.load(os.path.join(root, "tests", "golden", "policy_picking.npz"), device=env.device)
    rng = np.random.default_rng(9)
    obs = env.reset()
    alive = torch.ones(B, dtype=torch.bool, device=env.device)
    for s in range(100):
        obs, rew, done, _ = env.step(pol.predict(obs)[0])
        alive &= ~done
    env.set_noise(torch.full((B, 12), 0.5)); obs = env.make_observation()          # sim.forward() + observation from the current state
    touching = (obs[:, 0] > 0) & alive
    sel = torch.nonzero(touching).flatten().cpu().numpy()[:24]
    assert len(sel) >= 8, len(sel)                                                  # the policy has its fingers on the object in many envs by now
    q, v, w = [t.cpu().numpy().astype(np.float64) for t in env.sim.get_state()]
    ts = env.task_state().cpu().numpy().astype(np.float64)
    bigger = int(((env.sim.flags()[torch.as_tensor(sel, device=env.device)] & 32) != 0).sum())
    oes = {}
    for k in sel:
        oe = OracleEnv(names, task="picking")
        oe.obj_goal, oe.dest_goal = ts[k, 4:7].copy(), ts[k, 7:10].copy()
        oe.grip, oe.steps, oe.episodes = float(ts[k, 0]), int(ts[k, 1]), int(ts[k, 2])
        oe.set_state(q[k], v[k], w[k])
        oo = oe.observe(np.full(6, 0.5))[0]
        assert oo[0] == obs[k, 0].item() and np.abs(oo - obs[k].cpu().numpy()).max() < 2e-6, (k, oo[0], obs[k, 0].item())
        oes[k] = oe
    oerr, rerr, ncmp, flips = [], [], 0, 0
    for s in range(3):
        a = pol.predict(obs)[0]
        nz = rng.uniform(size=(B, 12)).astype(np.float32)
        env.set_noise(torch.tensor(nz))
        obs, rew, done, _ = env.step(a)
        an, on, rn, dn = a.cpu().numpy(), obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        for k in list(oes):
            oo, orew, odone, _ = oes[k].step(an[k].astype(np.float64), nz[k].astype(np.float64))
            assert bool(dn[k]) == odone, (s, k)
            if on[k, 0] != oo[0]:   # a grazing pad contact (sensor force around the 1e-3 threshold) may be on one side only for a step:
                flips += 1          # counted, and that env leaves the comparison (its reward term differs by construction)
                del oes[k]
                continue
            oerr.append(np.abs(on[k] - oo).max()); rerr.append(abs(rn[k] - orew)); ncmp += 1
            if odone:
                del oes[k]
    oerr, rerr = np.array(oerr), np.array(rerr)
    print("grasp regime, %d envs with finger-object contact (%d stepped by a bigger tier), %d env steps compared, %d touch-class flips: obs err median %.2e p90 %.2e max %.2e; reward err max %.2e" % (
        len(sel), bigger, ncmp, flips, np.median(oerr), np.percentile(oerr, 90), oerr.max(), rerr.max()))
    assert flips <= 2 and ncmp >= 40
    # 3x measured on MI355X (round 5, half-depth box-box contacts: obs median 8.9e-8, p90 2.9e-7, max 1.0e-5; reward 7.2e-6; 24 envs, 67 env steps;
    # round 4: 8.6e-8 / 2.5e-7 / 2.0e-6 / 5.2e-6 -- the maximum belongs to one env step)
    assert np.median(oerr) < 2.7e-7 and np.percentile(oerr, 90) < 8.8e-7 and oerr.max() < 3.1e-5 and rerr.max() < 2.2e-5
    env.close()



def test_in_kernel_reset_equals_the_explicit_chain_across_tasks():
    """Option auto_reset (reset + sim.forward() + first observation by the wave that finished the episode, hand-offs to bigger tiers and
    resident workers included) against step() + reset(done): observations, rewards, done flags, states and task rows BIT FOR BIT, for the
    tasks whose reset is draws + forward pass -- pushing ends every episode in its first step (65 536 in-kernel resets in 8 steps), the
    others run into their time-outs (step counters staggered over the episode length), one leg in the small-action regime."""
    from mujoco_jaco_amd.env import JacoBatchedEnv
    B = 8192

    def run(task, auto, nsteps, scale):
        env = JacoBatchedEnv(num_envs=B, task=task, seed=17, auto_reset=auto)
        env.reset()
        gen = torch.Generator(device=env.device); gen.manual_seed(3)
        nact = env.action_space.shape[0]
        ts = env.task_state(); ts[:, 1] = torch.randint(0, env.task_max_steps, (B,), device=env.device, generator=gen).float(); env.set_task_state(ts)
        out, ends = [], 0
        for s in range(nsteps):
            obs, rew, done, _ = env.step((torch.rand(B, nact, device=env.device, generator=gen) * 2 - 1) * scale)
            obs, rew, done = obs.clone(), rew.clone(), done.clone()
            ends += int(done.sum())
            if not env.auto_reset and bool(done.any()):
                obs[done] = env.reset(done)[done]
            out.append((obs, rew, done, env.sim.get_state()[0].clone(), env.task_state().clone()))
        assert int((env.sim.flags() & 31).max()) == 0
        env.close()
        return out, ends

    for task, n, scale in (("pushing", 8, 1.0), ("reaching", 30, 1.0), ("pickAndplace", 20, 1.0), ("picking", 16, 0.1)):
        (a, ends), (b, _) = run(task, True, n, scale), run(task, False, n, scale)
        assert ends > 100, (task, ends)
        for s, (x, y) in enumerate(zip(a, b)):
            for u, v in zip(x, y):
                assert torch.equal(u, v), (task, s)


def test_contact_free_kernel_at_env_level_matches_the_general_kernel():
    """The contact-free instantiation also serves env-level steps (task reaching on the arm-only model with contacts off: controller, 50
    substeps, observation, reward, termination).  Against the general kernel with its runtime disable_contact flag (option arm_kernel = 0):
    the two differ in how the joint-limit rows are solved (dof lanes vs matrix-core pass), so to fp32 rounding, done flags exactly."""
    from mujoco_jaco_amd.env import JacoBatchedEnv
    B = 512
    outs = []
    for arm in (1, 0):
        env = JacoBatchedEnv(num_envs=B, task="reaching", robot_file="jaco2_reaching_torque", seed=5)
        env.sim.set_option("disable_contact", 1); env.sim.set_option("arm_kernel", arm)
        env.reset()
        gen = torch.Generator(device=env.device); gen.manual_seed(2)
        env.sim.launch_count()
        for s in range(6):
            obs, rew, done, _ = env.step(torch.rand(B, 6, device=env.device, generator=gen) * 2 - 1)
        n = env.sim.launch_count()
        assert (n == 6) == (arm == 1), n                      # one launch per step on the contact-free kernel
        assert int((env.sim.flags() & 31).max()) == 0
        outs.append((obs.clone(), rew.clone(), done.clone(), env.sim.get_state()[0].clone()))
        env.close()
    (oa, ra, da, qa), (ob, rb, db, qb) = outs
    print("env-level reaching, contact-free vs general kernel after 6 steps: obs diff max %.2e, qpos diff max %.2e, reward diff max %.2e" % (
        (oa - ob).abs().max().item(), (qa - qb).abs().max().item(), (ra - rb).abs().max().item()))
    assert torch.equal(da, db) and torch.isfinite(oa).all()
    assert (oa - ob).abs().max().item() < 1e-4 and (qa - qb).abs().max().item() < 1e-4


def test_vec_env_adapter_with_the_in_kernel_reset():
    """JacoVecEnv(auto_reset=True): finished envs are reset inside jaco_step (no reset launch chain); the terminal observation and the success
    flag come from the kernel's latches (jaco_get_terminal_obs / jaco_get_last_terminal).  Everything the adapter returns -- observations,
    rewards, dones, terminal observations, success flags, episode lengths -- equals the explicit-reset adapter's, bit for bit."""
    from mujoco_jaco_amd.vec_env import JacoVecEnv
    B = 256
    outs = []
    for auto in (True, False):
        venv = JacoVecEnv(B, task="picking", frame_skip=4, seed=9, auto_reset=auto)
        assert venv.env.auto_reset == auto
        venv.reset()
        t = venv.env.task_state(); t[:64, 1] = 697; t[64:96, 1] = 698; venv.env.set_task_state(t)   # two groups of time-outs, one step apart
        gen = torch.Generator(device=venv.env.device); gen.manual_seed(1)
        rec = []
        for s in range(4):
            o, r, d, info = venv.step(torch.rand(B, 7, device=venv.env.device, generator=gen) * 2 - 1)
            rec.append((o.clone(), r.clone(), d.clone(), info["terminal_observation"], info["is_success"], info["episode_length"]))
        outs.append(rec)
        venv.close()
    ends = 0
    for x, y in zip(*outs):
        for u, v in zip(x, y):
            assert (u is None) == (v is None)
            if u is not None:
                assert torch.equal(u, v)
        ends += int(x[2].sum())
    assert ends == 96
    # dense_infos=True: the same rollout without a host synchronisation per step -- full-size tensors, valid where done is set
    venv = JacoVecEnv(B, task="picking", frame_skip=4, seed=9, auto_reset=True, dense_infos=True)
    venv.reset()
    t = venv.env.task_state(); t[:64, 1] = 697; t[64:96, 1] = 698; venv.env.set_task_state(t)
    gen = torch.Generator(device=venv.env.device); gen.manual_seed(1)
    for s in range(4):
        o, r, d, info = venv.step(torch.rand(B, 7, device=venv.env.device, generator=gen) * 2 - 1)
        ro, rr, rd, rt, rs, rl = outs[0][s]
        assert torch.equal(o, ro) and torch.equal(r, rr) and torch.equal(d, rd)
        assert info["terminal_observation"].shape == (B, 26) and info["is_success"].shape == (B,) and info["quarantined"].ndim == 0
        if rt is not None:
            assert torch.equal(info["terminal_observation"][d], rt) and torch.equal(info["is_success"][d], rs) and torch.equal(info["episode_length"][d], rl)
        assert (venv.episode_lengths[d] == 0).all()
    assert int(venv.quarantined_total_dev) == 0
    venv.close()
    # without auto_reset the terminal observation is latched as well (it equals the frozen env's obs row)
    env = JacoVecEnv(B, task="picking", frame_skip=4, seed=9).env
    env.reset()
    t = env.task_state(); t[:8, 1] = 699; env.set_task_state(t)
    o, r, d, _ = env.step(torch.zeros(B, 7, device=env.device))
    assert d[:8].all() and not d[8:].any()
    assert torch.equal(env.terminal_observation()[:8], o[:8]) and (env.terminal_observation()[8:] == 0).all()
    env.close()


def test_init_buffer_goals_through_the_c_abi():
    """kwarg init_buffer (env_mujoco_util.py:46,208-212) through JacoBatchedEnv -> jaco_set_init_buffer: every reset -- the explicit jaco_reset and
    the in-kernel auto-reset -- takes the reaching goal from rows 0 .. n - 2 of the buffer (row[1:4], row[4:7], no float16 cast); observation
    slots obs[17:23] of the reaching-goal branch show it.  The buffer is the golden one the reference's own branch was run on."""
    import os
    from mujoco_jaco_amd.env import JacoBatchedEnv
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glue_vectors_init_buffer.npz"))
    buf = G["buffer"].astype(np.float32)
    n, B = len(buf), 512
    goals = torch.tensor(np.hstack([buf[:, 1:4], buf[:, 4:7]]))
    for auto in (False, True):
        env = JacoBatchedEnv(num_envs=B, task="reaching", seed=3, frame_skip=2, init_buffer=buf, rulebased_subgoal=False, auto_reset=auto)
        obs = env.reset()
        g = env.task_state()[:, 32:38].cpu()
        which = (g[:, None, :] == goals[None]).all(-1)                      # [B, n]
        assert (which.sum(1) == 1).all() and not which[:, n - 1].any() and which[:, : n - 1].any(0).all()
        assert torch.allclose(obs[:, 17:20].cpu(), g[:, :3]) and torch.allclose(obs[:, 20:23].cpu() * np.pi, g[:, 3:], atol=1e-6)
        t = env.task_state(); t[:, 1] = 499; env.set_task_state(t)          # every env one step before the time-out
        o, r, d, _ = env.step(torch.zeros(B, 6, device=env.device))
        assert d.all()
        if not auto:
            env.reset(d)
        g2 = env.task_state()[:, 32:38].cpu()
        which2 = (g2[:, None, :] == goals[None]).all(-1)
        assert (which2.sum(1) == 1).all() and not which2[:, n - 1].any() and not torch.equal(g, g2)   # fresh draws, again from the buffer
        env.set_init_buffer(None)
        env.reset()
        assert not (env.task_state()[:, 32:38].cpu()[:, None, :] == goals[None]).all(-1).any()           # sampled goals again
        env.close()
    with pytest.raises(NotImplementedError):
        JacoBatchedEnv(num_envs=2, task="picking", reward_method="x")

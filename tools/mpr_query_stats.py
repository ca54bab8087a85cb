"""CPU diagnostic: support queries per MPR call under the reference's shipped picking policy (the regime in which the hull narrowphase
dominates the substep), split into penetrating / non-penetrating calls.  Two sides:

  python tools/mpr_query_stats.py oracle [episodes] [max_steps]     the oracle's MPR (single process: its counters are plain statics)
  python tools/mpr_query_stats.py kernel [max_steps] [seed]         the KERNEL's narrowphase (host build under the wavefront emulator: after the
                                                                    bounding-sphere test and the OBB cull), one env
"""
import sys
side = sys.argv[1] if len(sys.argv) > 1 else "oracle"
assert side in ("oracle", "kernel"), side
sys.argv = sys.argv[:1] + sys.argv[2:]


def oracle_side():
    import ctypes, os, sys
    ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch
    from mujoco_jaco_amd import workload
    from mujoco_jaco_amd.modelc import blob
    from mujoco_jaco_amd.policy import HPCPolicy
    from oracle_env import OracleEnv
    import oracle_binding

    neps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    maxsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    M = blob.load(os.path.join(ROOT, "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.jacomdl"))
    names = {}
    for line in open(os.path.join(ROOT, "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.names.txt")):
        k, v = line.strip().split(": ", 1); names[k] = v.split()
    pol = HPCPolicy.load(os.path.join(ROOT, "tests", "golden", "policy_picking.npz"), device=torch.device("cpu"))
    L = oracle_binding.Oracle().L if hasattr(oracle_binding.Oracle(), "L") else None
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libjaco_oracle.so"))
    lib.orc_debug_counter.restype = ctypes.c_long; lib.orc_debug_counter.argtypes = [ctypes.c_int, ctypes.c_int]
    rng = np.random.default_rng(0)
    q0 = workload.reset_states(M["qpos0"], neps, seed=5, f32_draws=True)
    tot = np.zeros(4)
    for ep in range(neps):
        oe = OracleEnv(names)
        oe.obj_goal = q0[ep, 9:12].copy(); oe.dest_goal = np.array([q0[ep, 16], q0[ep, 17], 0.3468])
        oe.set_state(q0[ep])
        obs = oe.observe(rng.uniform(size=6))[0]
        for i in range(4): lib.orc_debug_counter(i, 1)
        for s in range(maxsteps):
            a, _ = pol.predict(torch.tensor(obs[None], dtype=torch.float32))
            obs, r, d, succ = oe.step(a[0].numpy().astype(np.float64), rng.uniform(size=12))
            if d:
                break
        c = np.array([lib.orc_debug_counter(i, 1) for i in range(4)], float)
        tot += c
        print("episode %d: %d steps, done %s succ %s | MPR calls %d (%.2f per substep), hits %d, queries per hit %.1f, per miss %.1f" % (
            ep, s + 1, d, succ, c[0], c[0] / ((s + 1) * 50), c[1], c[2] / max(c[1], 1), c[3] / max(c[0] - c[1], 1)), flush=True)
    print("total: calls %d, hit share %.2f, queries per hit %.2f, per miss %.2f, share of all queries spent in hits %.2f" % (
        tot[0], tot[1] / tot[0], tot[2] / max(tot[1], 1), tot[3] / max(tot[0] - tot[1], 1), tot[2] / (tot[2] + tot[3])))


def kernel_side():
    import ctypes, os, sys
    ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch
    from mujoco_jaco_amd import workload
    from mujoco_jaco_amd.modelc import blob
    from mujoco_jaco_amd.policy import HPCPolicy
    from emu_binding import EmuJacoEnv
    maxsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    M = blob.load(os.path.join(ROOT, "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.jacomdl"))
    pol = HPCPolicy.load(os.path.join(ROOT, "tests", "golden", "policy_picking.npz"), device=torch.device("cpu"))
    e = EmuJacoEnv(nenv=1, frame_skip=50, seed=seed)
    e.L.emu_get_counter.argtypes = [ctypes.c_int, ctypes.c_int]; e.L.emu_get_counter.restype = ctypes.c_long
    q = workload.reset_states(M["qpos0"], 1, seed=seed)
    e.qpos[:] = q; e.task[:, 4:7] = q[:, 9:12]; e.task[:, 7] = q[:, 16]; e.task[:, 8] = q[:, 17]; e.task[:, 9] = 0.3468
    obs = e.forward()
    for i in range(8): e.L.emu_get_counter(i, 1)
    rows = []
    for s in range(maxsteps):
        a, _ = pol.predict(torch.tensor(obs, dtype=torch.float32))
        obs, r, d = e.env_step(a.numpy())
        c = [e.L.emu_get_counter(i, 1) for i in range(8)]
        rows.append(c)
        if s % 10 == 9 or d[0]:
            t = np.array(rows[-10:], float).sum(0)
            print("steps %3d-%3d: MPR calls per substep %.2f, hit share %.2f, queries per hit %.1f, per miss %.1f | contacts %d rows %d touch %d" % (
                s - 8, s + 1, t[3] / (50 * len(rows[-10:])), t[4] / max(t[3], 1), t[5] / max(t[4], 1), t[6] / max(t[3] - t[4], 1), e.stats[0, 0], e.stats[0, 1], int(obs[0, 0])), flush=True)
        if d[0]:
            break
    t = np.array(rows, float).sum(0)
    print("total over %d steps: MPR calls per substep %.2f, hit share %.2f, queries per hit %.2f, per miss %.2f, share of queries in hits %.2f; done %d reward %.1f" % (
        len(rows), t[3] / (50 * len(rows)), t[4] / max(t[3], 1), t[5] / max(t[4], 1), t[6] / max(t[3] - t[4], 1), t[5] / max(t[5] + t[6], 1), d[0], r[0]))


(oracle_side if side == "oracle" else kernel_side)()

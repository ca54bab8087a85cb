"""Env-level, closed-loop drift of the HIP env against the fp64 oracle env over 1 000 physics substeps.

The reference's loop is env_mujoco_util.py:73-90: every substep the operational-space controller reads the (one substep stale)
J / M / bias / EE pose and sets new motor torques, so unlike the ctrl-level drift metric (constant torques) the controller
closes the loop around the arm -- it pulls both trajectories towards the same target and damps the arm's share of the
divergence, while the contact dynamics of the free bodies stay open-loop.

Workload: `B` envs of the picking reset distribution, `nstep` env steps of 50 substeps (20 x 50 = 1 000), fresh random actions
U(-1, 1)^7 every env step, the same injected sub-goal noise on both sides.  Compared: qpos after every env step.
  GPU leg    JacoBatchedEnv (libjaco_env.so: jaco_step)                              -> .npz
  oracle leg tests/oracle_env.py OracleEnv (oracle/ fp64 C physics + oracle/glue.py), one process per host core
TEST / MEASUREMENT INFRASTRUCTURE: imported by bench.py's drift leg and tests only.

  python tools/env_drift.py [B] [nstep]        # both legs + summary (GPU box)
"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

MODEL = "jaco2_curtain_torque"
MARKS = (2, 6, 20)   # env steps = 100 / 300 / 1 000 substeps at frame_skip 50


def inputs(B, nstep, seed=71, scale=1.0):
    from mujoco_jaco_amd import workload
    from mujoco_jaco_amd.modelc import blob
    M = blob.load(os.path.join(ROOT, "mujoco_jaco_amd", "assets", MODEL + ".jacomdl"))
    q0 = workload.reset_states(M["qpos0"], B, seed=seed, f32_draws=True)
    rng = np.random.default_rng(seed + 1)
    act = (rng.uniform(-1, 1, (nstep, B, 7)) * scale).astype(np.float32)
    noise = rng.uniform(size=(nstep + 1, B, 12)).astype(np.float32)
    return q0, act, noise


def gpu_leg(out_path, B, nstep, compensated=1, task="picking", scale=1.0):
    import torch
    from mujoco_jaco_amd.env import JacoBatchedEnv
    q0, act, noise = inputs(B, nstep, scale=scale)
    env = JacoBatchedEnv(num_envs=B, task=task)
    env.sim.set_option("compensated", compensated)
    for kv in os.environ.get("JACO_DRIFT_OPTS", "").split(","):   # e.g. JACO_DRIFT_OPTS=sep_cache=0,pair_list=0 (which execution option owns an outlier?)
        if kv:
            env.sim.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    dev = env.device
    env.sim.set_state(torch.tensor(q0, dtype=torch.float32, device=dev), torch.zeros(B, 21, device=dev), torch.zeros(B, 21, device=dev))
    t = env.task_state(); t[:] = 0; t[:, 0] = 0.6; t[:, 16] = 0.6
    t[:, 4:7] = torch.tensor(q0[:, 9:12], dtype=torch.float32); t[:, 7:9] = torch.tensor(q0[:, 16:18], dtype=torch.float32); t[:, 9] = 0.3468
    env.set_task_state(t)
    env.set_noise(torch.tensor(noise[0])); env.make_observation()
    qs, obs_all, dones, rews = [], [], [], []
    for s in range(nstep):
        env.set_noise(torch.tensor(noise[s + 1]))
        obs, rew, done, _ = env.step(torch.tensor(act[s]))
        qs.append(env.sim.get_state()[0].cpu().numpy()); obs_all.append(obs.cpu().numpy()); dones.append(done.cpu().numpy()); rews.append(rew.cpu().numpy())
    np.savez(out_path, qpos=np.array(qs), obs=np.array(obs_all), done=np.array(dones), reward=np.array(rews), flags=env.sim.flags().cpu().numpy())
    env.close()


_W = {}


def _oracle_one(k):
    from oracle_env import OracleEnv
    q0, act, noise, names, nstep = _W["q0"], _W["act"], _W["noise"], _W["names"], _W["nstep"]
    oe = OracleEnv(names)
    oe.obj_goal = q0[k, 9:12].astype(np.float32).astype(np.float64)
    oe.dest_goal = np.array([q0[k, 16], q0[k, 17], 0.3468]).astype(np.float32).astype(np.float64)
    oe.set_state(q0[k])
    oe.observe(noise[0, k, 6:].astype(np.float64))
    qs, obs_all, dones, rews = [], [], [], []
    alive = True
    for s in range(nstep):
        if alive:
            o, r, d, _ = oe.step(act[s, k].astype(np.float64), noise[s + 1, k].astype(np.float64))
            qlast, olast = oe.o.get("qpos").copy(), np.asarray(o, np.float64)
            alive = not d
        else:
            r, d = 0.0, True   # (the HIP env freezes a finished env until it is reset)
        qs.append(qlast); obs_all.append(olast); dones.append(d); rews.append(r)
    return np.array(qs), np.array(obs_all), np.array(dones), np.array(rews)


def oracle_leg(B, nstep, nproc=None, scale=1.0):
    """(before anything in this process has touched the GPU: the pool forks)"""
    import multiprocessing as mp
    q0, act, noise = inputs(B, nstep, scale=scale)
    names = {}
    for line in open(os.path.join(ROOT, "mujoco_jaco_amd", "assets", MODEL + ".names.txt")):
        k, v = line.strip().split(": ", 1)
        names[k] = v.split()
    _W.update(q0=q0, act=act, noise=noise, names=names, nstep=nstep)
    nproc = nproc or int(os.environ.get("JACO_DRIFT_NPROC", "0")) or min(B, os.cpu_count() or 1)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    with mp.get_context("fork").Pool(nproc) as pool:
        res = pool.map(_oracle_one, range(B), chunksize=max(1, B // (4 * nproc)))
    return {"qpos": np.stack([r[0] for r in res], 1), "obs": np.stack([r[1] for r in res], 1),
            "done": np.stack([r[2] for r in res], 1), "reward": np.stack([r[3] for r in res], 1)}


def summarize(gpu, ref, marks=MARKS, frame_skip=50):
    """Errors over the envs still running on both sides at the mark (a finished env is frozen); done flags compared exactly."""
    out = {"envs": int(gpu["qpos"].shape[1]), "done_flags_equal": bool(np.array_equal(gpu["done"].astype(bool), ref["done"].astype(bool)))}
    for mk in marks:
        if mk > gpu["qpos"].shape[0]:
            continue
        live = ~(gpu["done"][:mk].astype(bool).any(0) | ref["done"][:mk].astype(bool).any(0))
        e = np.abs(gpu["qpos"][mk - 1].astype(np.float64) - ref["qpos"][mk - 1]).max(1)[live]
        eo = np.abs(gpu["obs"][mk - 1].astype(np.float64) - ref["obs"][mk - 1]).max(1)[live]
        out["after_%d_substeps" % (mk * frame_skip)] = {
            "live_envs": int(live.sum()), "median": float(np.median(e)), "p90": float(np.percentile(e, 90)), "max": float(e.max()),
            "frac_le_1e-4": float(np.mean(e <= 1e-4)), "obs_median": float(np.median(eo)), "obs_max": float(eo.max())}
    return out


if __name__ == "__main__":
    import json
    import subprocess
    if len(sys.argv) > 1 and sys.argv[1] == "gpu":
        gpu_leg(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 1, scale=float(sys.argv[6]) if len(sys.argv) > 6 else 1.0)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "oracle":   # the oracle leg alone -> .npz (a fresh process: its pool forks before any GPU use)
        np.savez(sys.argv[2], **oracle_leg(int(sys.argv[3]), int(sys.argv[4]), scale=float(sys.argv[5]) if len(sys.argv) > 5 else 1.0))
        sys.exit(0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    nstep = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    ref = oracle_leg(B, nstep)
    for comp in (0, 1):
        path = os.path.join(ROOT, "gpurun_out", "env_drift_gpu_%d.npz" % comp)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "gpu", path, str(B), str(nstep), str(comp)], check=True)
        print("compensated %d:" % comp, json.dumps(summarize(dict(np.load(path)), ref)), flush=True)

#!/usr/bin/env python
"""Headline benchmark: env-steps/sec of the batched Jaco physics step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

Workload: 65 536 environments per GPU, full Jaco + 3-finger gripper +
table/object contacts, states drawn from the reference's `picking` reset distribution, inputs resident in HBM.
Default `--level env`: one "step" = one JacoMujocoEnv.step (env_mujoco.py:116-139) for every env of the batch:
random actions U(-1,1)^7 -> _take_action, `--frame-skip` (default 50 = the reference, env_mujoco.py:24) substeps of
operational-space control + physics, observation, reward, termination (jaco_step).  `--level ctrl` times the ctrl-level
entry jaco_physics_step (random motor torques, `--frame-skip` substeps per step, default 1) used for oracle parity.
Environments are independent, so ranks shard them with no data-path collective; the only collective is
the per-step all_gather of the observation rows, as the north star prescribes.
Also reports: roofline of the physics kernel (algorithmic bytes / HIP-event kernel time vs 8 TB/s) and,
on rank 0 at N=1, the fp64 oracle timed on the host cores as a CPU baseline ("port").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(model, frame_skip, budget_s=15.0):
    """fp64 oracle (oracle/, test infrastructure) on the host cores: a reported baseline, not the target."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from mujoco_jaco_amd import workload
    from mujoco_jaco_amd.modelc import blob
    from oracle_binding import Oracle
    cores = os.cpu_count() or 1
    M = blob.load(os.path.join(ROOT, "mujoco_jaco_amd", "assets", model + ".jacomdl"))
    o = Oracle(model)
    nenv, nsub = 64 * cores, 25
    q = workload.reset_states(M["qpos0"], nenv, seed=123)
    v = np.zeros((nenv, o.nv)); w = np.zeros((nenv, o.nv))
    c = workload.random_ctrl(nenv, seed=124, scale=0.2)[:, :o.nu].copy()
    o.step_batch(q, v, w, c, nsub=5, nthreads=cores)  # warm
    t = time.time(); done = 0
    while time.time() - t < budget_s:
        o.step_batch(q, v, w, c, nsub=nsub, nthreads=cores)
        done += nenv * nsub
    dt = time.time() - t
    return {"value": done / dt / frame_skip, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d physics substeps of the same workload (random torques; the controller / observation glue is not timed), fp64 C oracle, OpenMP over %d threads, %.1f s; frame_skip %d"
                      % (nenv, done // nenv, cores, dt, frame_skip)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=65536, help="environments per GPU")
    ap.add_argument("--frame-skip", type=int, default=None)
    ap.add_argument("--level", choices=["env", "ctrl"], default="env")
    ap.add_argument("--model", default="jaco2_curtain_torque")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from mujoco_jaco_amd import workload
    from mujoco_jaco_amd.modelc import blob
    from mujoco_jaco_amd.physics import BatchedMujoco

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: WORLD_SIZE=%d but --gpus %d (launch N>1 through torch.distributed.run)" % (world, args.gpus), file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    B = args.batch
    fs = args.frame_skip if args.frame_skip is not None else (50 if args.level == "env" else 1)
    M = blob.load(os.path.join(ROOT, "mujoco_jaco_amd", "assets", args.model + ".jacomdl"))
    if args.level == "env":
        from mujoco_jaco_amd.env import JacoBatchedEnv
        from mujoco_jaco_amd.sharding import ObsGather, env_seed
        genv = JacoBatchedEnv(num_envs=B, device=local_rank, frame_skip=fs, seed=env_seed(1000, rank), task="picking", robot_file=args.model)
        env = genv.sim
        obs = genv.reset()
        gen = torch.Generator(device=dev); gen.manual_seed(2000 + rank)
        actions = [torch.rand(B, 7, device=dev, generator=gen) * 2 - 1 for _ in range(4)]
        gather = ObsGather(B, 26, dev) if world > 1 else None
        it = [0]

        def step():
            o, r, d, _ = genv.step(actions[it[0] % 4]); it[0] += 1
            if world > 1:  # one collective per rollout step: concatenate the observation rows of all shards
                gather(o)
    else:
        env = BatchedMujoco(B, robot_file=args.model, device=local_rank, frame_skip=fs, seed=rank)
        q = torch.tensor(workload.reset_states(M["qpos0"], B, seed=1000 + rank), dtype=torch.float32, device=dev)
        ctrl = torch.tensor(workload.random_ctrl(B, seed=2000 + rank, scale=0.2)[:, :env.nu].copy(), dtype=torch.float32, device=dev)
        env.set_state(q, None, None)
        gathered = torch.empty(world * B, env.nq, device=dev) if world > 1 else None

        def step():
            env.send_forces(ctrl, nsub=fs)
            if world > 1:
                qpos, _, _ = env.state_views()
                dist.all_gather_into_tensor(gathered, qpos)

    for _ in range(args.warmup):
        step()
    env.enable_timing(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    kern_ms, launches = env.kernel_time_ms()
    env.enable_timing(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    flags = int(env.flags().max().item())
    stats = env.stats().float().mean(0).cpu().numpy()

    if rank == 0:
        # algorithmic HBM bytes of one launch of jaco_physics_kernel per env (DESIGN.md "Measurement"):
        # reads qpos, qvel, qacc_warmstart, ctrl; writes qpos, qvel, qacc_warmstart, sensordata.
        bytes_per_env = 4 * (2 * env.nq + 4 * env.nv + env.nu + env.nsensor)
        if args.level == "env":  # + action in, obs / reward / done out, task and controller-cache rows in and out
            bytes_per_env = 4 * (2 * env.nq + 4 * env.nv + env.nsensor + 7 + 26 + 1 + 2 * 32 + 2 * 96) + 1
        achieved = bytes_per_env * B / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        traffic = None
        try:  # HBM bytes per launch from the committed rocprofv3 PMC passes (tools/profile_pmc.sh), same level / batch only
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            key = "%s_B%d_fs%d" % (args.level, B, fs)
            traffic = pm.get(key, {}).get("hbm_bytes_per_launch")
        except Exception:
            pass
        out = {
            "metric": "env-steps/sec at batch 65 536 (full Jaco + gripper + contacts)",
            "value": world * B * args.steps / dt, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config3: %d envs/GPU, %s, picking reset distribution, %s"
                                   % (B, args.model, "random actions, env-level jaco_step (OSC + %d substeps + obs/reward/done)" % fs
                                      if args.level == "env" else "random motor ctrl, ctrl-level jaco_physics_step"),
                       "level": args.level,
                       "envs_per_gpu": B, "frame_skip": fs, "substeps_per_s": world * B * args.steps * fs / dt,
                       "sharding": "independent env shards per rank" + ("; one all_gather of [B,%d] f32 rows per step" % (26 if args.level == "env" else env.nq) if world > 1 else ""),
                       "mean_contacts": float(stats[0]), "mean_rows": float(stats[1]), "mean_newton_iters": float(stats[2]), "flags_or": flags,
                       "heavy_tier_fraction": float(((env.flags() & 32) != 0).float().mean().item())},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "jaco_physics_kernel", "kernel_ms": kern_ms, "launches": launches,
                         "algorithmic_bytes_per_env_launch": bytes_per_env,
                         "note": "latency/occupancy-bound by design (SURVEY 8d): algorithmic traffic is ~1 KB per env per launch; `traffic` = L2 fabric-side bytes per launch from profiles/r01_pmc_traffic.json (FETCH_SIZE + WRITE_SIZE, Infinity-Cache hits included): 95 % of it are register-spill lines (6 MB per XCD against a 4 MB L2) cycling between L2 and the Infinity Cache, the env state is 0.1 GB"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.model, fs)
        print(json.dumps(out))
    env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

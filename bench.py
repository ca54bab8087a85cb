#!/usr/bin/env python
"""Headline benchmark: env-steps/sec of the batched Jaco environment step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed launcher: this process spawns N fresh rank processes itself BEFORE anything touches a
GPU (never an exec of a process that initialised HIP) and forwards rank 0's JSON line; under `torch.distributed.run`
(WORLD_SIZE set) it is a rank.

Workload: 65 536 environments per GPU, full Jaco + 3-finger gripper + table/object
contacts, the reference's `picking` reset distribution, inputs resident in HBM.
Default `--level env`: one "step" = one JacoMujocoEnv.step (env_mujoco.py:116-139) for every env of the batch: a fresh
random action U(-1,1)^7 x --action-scale -> _take_action, `--frame-skip` (default 50 = the reference, env_mujoco.py:24)
substeps of operational-space control + physics, observation, reward, termination (jaco_step), and the reset of the envs that
finished -- inside the timed region, no host synchronisation: by default inside jaco_step itself (option auto_reset: the wave that
finished an episode does sim.reset(), the draws and sim.forward()), with --explicit-reset as a masked jaco_reset launch chain.  The batch is first rolled `--preroll`
untimed steps (with the same masked resets, episode ages staggered over [0, 700)) so the timed window sees a rollout in
progress, not the 25 steps after a global reset.  `--level ctrl` times the ctrl-level entry jaco_physics_step (random motor
torques, `--frame-skip` substeps per step, default 1) used for oracle parity.
Environments are independent, so ranks shard them with no data-path collective; the only collective is the per-step
all_gather of the observation rows, as the north star prescribes.
Also reported: the roofline of the physics kernel (algorithmic bytes / HIP-event kernel time vs 8 TB/s), VALU issue rate
(from the committed PMC pass), throughput at small action scales, and -- on rank 0 at N = 1 -- the fp64 oracle timed on the
host cores as a CPU baseline ("port") together with the 1 000-substep drift of the HIP path against that same oracle run.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2   # wave-instructions/s: 1 024 SIMD-32 units, one wave64 VALU instruction per 2 cycles, 2.4 GHz
PMC_FILE = os.path.join(ROOT, "profiles", "r05_pmc.json")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=65536, help="environments per GPU")
    ap.add_argument("--frame-skip", type=int, default=None)
    ap.add_argument("--level", choices=["env", "ctrl"], default="env")
    ap.add_argument("--model", default="jaco2_curtain_torque")
    ap.add_argument("--task", default="picking")
    ap.add_argument("--action-scale", type=float, default=1.0, help="actions are U(-1,1)^7 times this (1.0 = the headline)")
    ap.add_argument("--preroll", type=int, default=700,
                    help="untimed env steps (finished envs reset as in the timed loop) before warmup; default = one full episode length: with the step "
                         "counters staggered over [0, 700) every env has then been reset once and the episodes' physical ages are spread over [0, 700) "
                         "(profiles/r04_soak_curve.txt: ms/step does not depend on the rollout's age from step 50 on)")
    ap.add_argument("--extra-scales", default="0.3,0.05", help="action scales also timed (briefly, each in a child process before the headline run); '' = none")
    ap.add_argument("--policy", default=None,
                    help="actions from one of the reference's shipped policies instead of U(-1,1): 'picking' | 'placing' (tests/golden/policy_<task>.npz) or a policy.zip / .npz path; the policy's forward pass is inside the timed step")
    ap.add_argument("--policy-leg", default="picking", help="policy also timed (briefly, in a child process before the headline run); '' = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hints", type=int, default=None, help="tier hint mode of the env handle (diagnostic; default = the library's)")
    ap.add_argument("--drift-gpu-leg", default=None, help=argparse.SUPPRESS)   # internal: child process that steps the HIP path on the drift workload
    ap.add_argument("--env-drift-gpu-leg", default=None, help=argparse.SUPPRESS)   # internal: child process, HIP leg of the env-level closed-loop drift
    ap.add_argument("--no-reset", action="store_true", help="finished envs stay frozen (BASELINE config 4: termination masking, no auto-reset)")
    ap.add_argument("--explicit-reset", action="store_true", help="reset finished envs with a masked jaco_reset launch chain after every step instead of inside jaco_step (option auto_reset)")
    ap.add_argument("--no-contact", action="store_true", help="contacts disabled (BASELINE config 2: arm-only model)")
    ap.add_argument("--config-legs", default="2,4", help="BASELINE configs also timed (briefly, each in a child process before the headline run); '' = none")
    ap.add_argument("--flag-census", type=int, default=20, help="untimed env steps after the timed window in which error flags are counted per step (0 = off)")
    ap.add_argument("--set-option", action="append", default=[], help="name=value passed to jaco_set_option (e.g. compensated=0); repeatable")
    ap.add_argument("--dry-gather", action="store_true",
                    help="CPU rehearsal of the N > 1 plumbing (spawn, rendezvous, gather, max-over-ranks timing) on gloo: no GPU, no physics")
    return ap.parse_args(argv)


def self_launch(args):
    """--gpus N > 1 outside a launcher: N child ranks, started before this process has made any GPU call."""
    # (bind-close-reuse leaves a window in which another process may take the port: the ranks then fail their rendezvous, the
    # poll below reports it and the caller can re-run; SO_REUSEADDR keeps the port usable right after the close)
    s = socket.socket(); s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # poll all ranks: the first one that fails takes the others down (a rank that died before or inside the RCCL rendezvous would
    # otherwise leave rank 0 -- and this launcher -- waiting for ever); overall time limit as a last resort
    import threading
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("JACO_BENCH_LAUNCH_TIMEOUT", "3000"))
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [r for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad or time.time() > deadline:
            failed = ("rank %d exited with %d" % (bad[0], procs[bad[0]].returncode)) if bad else "time limit"
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    sys.stdout.write((out[0] if out else b"").decode())
    sys.stdout.flush()
    if failed:
        print("bench.py: multi-rank launch failed: %s" % failed, file=sys.stderr)
        return 1
    return max(abs(p.returncode) for p in procs)


LEG_FLAGS = {}   # per side leg: capacity / solver error bits seen in its timed window


def leg_flags(line):
    """What a side leg's own JSON line says about dropped contacts / rows (the last capacity tier has nobody to hand over to)."""
    c = line["config"]
    return {"error_flags_or": c.get("error_flags_or"), "flagged_env_steps_per_million": c.get("flagged_env_steps_per_million")}


def small_action_runs(args):
    """Throughput at the smaller action scales (the regime a converged policy lives in: the EE sits on its "hand" marker), each in
    its own child process started BEFORE this process touches the GPU: the parent's launch statistics (rocprofv3 --stats, HIP-event
    kernel time) then hold the headline workload only."""
    out = {}
    for sc in [x for x in args.extra_scales.split(",") if x]:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--action-scale", sc, "--steps", "6", "--warmup", "2", "--preroll", "16",
               "--batch", str(args.batch), "--model", args.model, "--task", args.task, "--no-cpu-baseline", "--extra-scales", "", "--policy-leg", "", "--config-legs", ""]
        if args.frame_skip is not None:
            cmd += ["--frame-skip", str(args.frame_skip)]
        cmd += [x for kv in args.set_option for x in ("--set-option", kv)]   # (library options follow into the side legs: A/B runs)
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            line = json.loads(p.stdout.strip().splitlines()[-1])
            out[sc] = line["value"]
            LEG_FLAGS["action_scale_%s" % sc] = leg_flags(line)
        except Exception as e:  # a failed side measurement must not take the headline down
            out[sc] = None
            print("bench.py: action scale %s run failed: %r" % (sc, e), file=sys.stderr)
    return out


def policy_run(args):
    """Throughput with the reference's shipped policy choosing the actions (deterministic HPC forward pass inside the timed step,
    episodes in every phase after a long pre-roll): the regime the env is used in once a policy has converged.  Child process, as above."""
    if not args.policy_leg or args.policy:
        return None
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--policy", args.policy_leg, "--task", args.policy_leg, "--steps", "10", "--warmup", "2",
           "--preroll", "180", "--batch", str(args.batch), "--model", args.model, "--no-cpu-baseline", "--extra-scales", "", "--policy-leg", "", "--config-legs", ""]
    if args.frame_skip is not None:
        cmd += ["--frame-skip", str(args.frame_skip)]
    cmd += [x for kv in args.set_option for x in ("--set-option", kv)]   # (library options follow into the side legs: A/B runs)
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        line = json.loads(p.stdout.strip().splitlines()[-1])
        LEG_FLAGS["policy_%s" % args.policy_leg] = leg_flags(line)
        return {"policy": args.policy_leg, "value": line["value"], "ms_per_step": line["ms_per_step"], "done_fraction": line["config"]["done_fraction"],
                "mean_rows": line["config"]["mean_rows"], "heavy_tier_fraction": line["config"]["heavy_tier_fraction"], "preroll_steps": 180}
    except Exception as e:
        print("bench.py: policy run failed: %r" % (e,), file=sys.stderr)
        return None


def config_legs(args):
    """BASELINE.json configs 2 and 4 (child processes, as above):
    config 2: 4 096 envs, arm-only model (gripper frozen, no free bodies, contacts disabled), random motor torques, ctrl level, one
              launch = `frame_skip` (50) substeps;
    config 4: 65 536 envs, full model, env-level step with frame_skip 4 and termination masking (finished envs stay frozen, no reset)."""
    out = {}
    common = ["--gpus", "1", "--no-cpu-baseline", "--extra-scales", "", "--policy-leg", "", "--config-legs", ""]
    legs = {"2": ["--level", "ctrl", "--model", "jaco2_reaching_torque", "--batch", "4096", "--frame-skip", "50", "--no-contact", "--steps", "200", "--warmup", "20"],
            "4": ["--level", "env", "--model", args.model, "--task", args.task, "--batch", str(args.batch), "--frame-skip", "4", "--no-reset", "--steps", "60", "--warmup", "10", "--preroll", "60"]}
    for c in [x for x in args.config_legs.split(",") if x]:
        if c not in legs:
            continue
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__)] + legs[c] + common + [x for kv in args.set_option for x in ("--set-option", kv)], capture_output=True, text=True, timeout=600)
            line = json.loads(p.stdout.strip().splitlines()[-1])
            LEG_FLAGS["config%s" % c] = leg_flags(line)
            out["config%s" % c] = {"env_steps_per_s": line["value"], "ms_per_step": line["ms_per_step"], "kernel_ms": line["roofline"]["kernel_ms"],
                                   "frame_skip": line["config"]["frame_skip"], "envs": line["config"]["envs_per_gpu"], "substeps_per_s": line["config"]["substeps_per_s"],
                                   "launches_per_step": line["config"].get("launches_per_step"), "done_fraction": line["config"]["done_fraction"],
                                   # (no-reset legs: a finished env stays frozen and its env steps cost next to nothing -- the share of env steps that were real work)
                                   "live_env_steps_per_s": line["value"] * (1.0 - line["config"]["done_fraction"]),
                                   "outside_kernel_ms": line["ms_per_step"] - line["roofline"]["kernel_ms"]}
        except Exception as e:
            out["config%s" % c] = None
            print("bench.py: config %s leg failed: %r" % (c, e), file=sys.stderr)
    return out


DRIFT_MARKS = (100, 300, 1000)


def usable_cores():
    """The CPU share this process may actually use -- the smaller of its affinity mask and its cgroup's CPU quota (cpu.max of cgroup v2,
    cpu.cfs_quota_us / cpu.cfs_period_us of v1) -- not the machine's logical CPU count: a GPU box that shows 256 logical CPUs (and a
    256-wide affinity mask) hands a one-GPU lease 16 of them through the quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        n = min(n, max(1, int(quota + 0.5)))
    env = os.environ.get("JACO_CPU_CORES")   # (explicit override for boxes whose share is not visible from inside)
    return max(1, int(env)) if env else max(1, n)


ORACLE_STATUS = ("port: fp64 C restatement, pinned bit for bit to the object transients the reference's MuJoCo recorded (plane-box and box-box "
                 "contacts, soft-constraint chain, Euler step: tests/test_mujoco_statics.py); articulated-arm dynamics, hull contacts and the controller unpinned")
DRIFT_ENVS = 2048   # fixed: the HIP leg (child process) and the oracle leg must agree on it whatever each thinks the core count is


def drift_workload(model, nenv=DRIFT_ENVS):
    import numpy as np
    from mujoco_jaco_amd import workload
    from mujoco_jaco_amd.modelc import blob
    M = blob.load(os.path.join(ROOT, "mujoco_jaco_amd", "assets", model + ".jacomdl"))
    nu = int(M["nu"][0])
    q0 = workload.reset_states(M["qpos0"], nenv, seed=41, f32_draws=True)
    c = np.ascontiguousarray(workload.random_ctrl(nenv, seed=42, scale=0.2)[:, :nu].astype(np.float32).astype(np.float64))
    return nenv, q0, c


def drift_gpu_leg(args):
    """Child process (started before the parent touches the GPU): the HIP path on the drift workload, qpos at the marks -> .npz.
    Kept out of the parent so that its handful of 100..700-substep ctrl-level launches do not sit in the headline run's
    launch statistics."""
    import numpy as np
    import torch
    from mujoco_jaco_amd.physics import BatchedMujoco
    nenv, q0, c = drift_workload(args.model)
    e2 = BatchedMujoco(nenv, robot_file=args.model, device=0)
    e2.set_state(torch.tensor(q0, dtype=torch.float32, device=e2.device), None, None)
    cc = torch.tensor(c, dtype=torch.float32, device=e2.device)
    got, done = {}, 0
    for mk in DRIFT_MARKS:
        e2.send_forces(cc, nsub=mk - done)
        done = mk
        got["q%d" % mk] = e2.get_state()[0].cpu().numpy().astype(np.float64)
    np.savez(args.drift_gpu_leg, flags=e2.flags().cpu().numpy(), **got)
    e2.close()
    return 0


def cpu_baseline_and_drift(model, frame_skip, gpu_npz):
    """fp64 oracle (oracle/, test infrastructure) on the host cores: a reported baseline, not the target.  The same oracle
    run doubles as the reference trajectory of the drift metric: `gpu_npz` holds the HIP path's qpos on the same inputs at
    the marks (drift_gpu_leg)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from oracle_binding import Oracle
    cores = usable_cores()
    o = Oracle(model)
    calib = None
    if cores > 32 and not os.environ.get("JACO_CPU_CORES"):
        # neither the affinity mask nor a visible cgroup quota narrows the machine's logical CPUs down: find the thread count the box really
        # serves by a short sweep (the smallest count within 5 % of the best throughput) instead of claiming all of them
        nen, qq, cc = drift_workload(model)
        rates = {}
        for nt in (4, 8, 16, 32, 64, 128):
            if nt > cores:
                break
            qa, va, wa = np.ascontiguousarray(qq[:512].copy()), np.zeros((512, o.nv)), np.zeros((512, o.nv))
            o.step_batch(qa, va, wa, cc[:512], nsub=2, nthreads=nt)
            t0 = time.time(); o.step_batch(qa, va, wa, cc[:512], nsub=60, nthreads=nt); rates[nt] = 512 * 60 / (time.time() - t0)
        best = max(rates.values())
        cores = min(nt for nt, r in rates.items() if r >= 0.95 * best)
        calib = {str(k): round(v) for k, v in rates.items()}
    nenv, q0, c = drift_workload(model)
    marks = DRIFT_MARKS
    q, v, w = np.ascontiguousarray(q0.copy()), np.zeros((nenv, o.nv)), np.zeros((nenv, o.nv))
    o.step_batch(q.copy(), v.copy(), w.copy(), c, nsub=2, nthreads=cores)  # warm the thread pool
    ref, done, t = {}, 0, time.time()
    for mk in marks:
        o.step_batch(q, v, w, c, nsub=mk - done, nthreads=cores)
        done = mk
        ref[mk] = q.copy()
    dt = time.time() - t
    # ... and one env on one core (plan item 2: single-thread and all-cores)
    q1, v1, w1 = np.ascontiguousarray(q0[:1].copy()), np.zeros((1, o.nv)), np.zeros((1, o.nv))
    t1 = time.time(); o.step_batch(q1, v1, w1, c[:1], nsub=2000, nthreads=1); dt1 = time.time() - t1
    value, single = nenv * done / dt / frame_skip, 2000 / dt1 / frame_skip
    base = {"value": value, "unit": "env-steps/s", "cores": cores, "cores_usable": cores, "cores_logical": os.cpu_count() or 1, "kind": "port",
            "single_thread_env_steps_per_s": single, "parallel_efficiency": value / (cores * single), "thread_sweep_substeps_per_s": calib,
            "sample": "%d envs x %d physics substeps of the same workload (picking reset distribution, constant random torques; the controller / observation glue is not timed), fp64 C oracle, OpenMP over %d threads, %.1f s; frame_skip %d"
                      % (nenv, done, cores, dt, frame_skip)}
    if gpu_npz is None or not os.path.exists(gpu_npz):
        return base, None
    got = {mk: np.load(gpu_npz)["q%d" % mk] for mk in marks}
    drift = {"metric": "max-abs qpos error of the HIP path vs the fp64 oracle, same (qpos, qvel, ctrl), ctrl level", "envs": nenv, "oracle": ORACLE_STATUS}
    for mk in marks:
        e = np.abs(got[mk] - ref[mk]).max(1)
        drift["after_%d_substeps" % mk] = {"median": float(np.median(e)), "p90": float(np.percentile(e, 90)), "max": float(e.max()),
                                            "frac_le_1e-4": float(np.mean(e <= 1e-4))}
    # controls: the same fp64 code (no fp32 arithmetic anywhere) against itself, (1) carrying fp32-rounded state -- what an engine
    # with a plain fp32 state is bound by (rounds 1-2) -- and (3) keeping its state in fp64 but evaluating every forward pass at the
    # fp32 rounding of it -- the ceiling of an fp32 engine that carries its state compensated, as this one does
    for mode, key in ((1, "control_fp64_oracle_with_fp32_rounded_state"), (3, "control_fp64_state_fp32_rounded_evaluation")):
        oc = Oracle(model); oc.option("round_state", mode)
        q, v, w = np.ascontiguousarray(q0.copy()), np.zeros((nenv, o.nv)), np.zeros((nenv, o.nv))
        done, ctl = 0, {}
        for mk in marks:
            oc.step_batch(q, v, w, c, nsub=mk - done, nthreads=cores)
            done = mk
            e = np.abs(q - ref[mk]).max(1)
            ctl["after_%d_substeps" % mk] = {"median": float(np.median(e)), "frac_le_1e-4": float(np.mean(e <= 1e-4))}
        drift[key] = ctl
    drift["note"] = "qpos / qvel are carried as compensated float pairs (hi + lo): rounding perturbs each evaluation but does not accumulate in the state; what remains is the contact dynamics amplifying last-bit differences -- the fp64 oracle evaluated at fp32-rounded state parts from itself the same way (second control; the first is the bound of a plain fp32 state)"
    return base, drift


ENV_DRIFT_ENVS, ENV_DRIFT_STEPS = 1024, 20   # 20 env steps x 50 substeps = 1 000 substeps


def env_level_drift(args):
    """Closed-loop drift (tools/env_drift.py): JacoBatchedEnv (child process, HIP) against the fp64 oracle env (process pool on the
    host cores, forked here BEFORE this process touches the GPU) on the same actions and injected noise; the reference's loop is
    env_mujoco_util.py:73-90 (the controller runs every substep)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import numpy as np
        import env_drift
        path = os.path.join(ROOT, "gpurun_out", "bench_env_drift_gpu_%d.npz" % os.getpid())
        os.makedirs(os.path.dirname(path), exist_ok=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--env-drift-gpu-leg", path], timeout=600, check=True, stdout=subprocess.DEVNULL)
        t = time.time()
        ref = env_drift.oracle_leg(ENV_DRIFT_ENVS, ENV_DRIFT_STEPS)
        res = env_drift.summarize(dict(np.load(path)), ref)
        os.remove(path)
        res["metric"] = "max-abs qpos error of the HIP env vs the fp64 oracle env, same actions + injected noise, env level (OSC every substep), frame_skip 50"
        res["oracle"] = ORACLE_STATUS
        res["oracle_seconds"] = time.time() - t
        return res
    except Exception as e:
        print("bench.py: env-level drift leg failed: %r" % (e,), file=sys.stderr)
        return None


def dry_rank(args, world, rank):
    """The N > 1 plumbing without a GPU: gloo, synthetic observation rows, the same barrier / max-over-ranks / JSON path."""
    import torch
    import torch.distributed as dist
    from mujoco_jaco_amd.sharding import ObsGather
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = min(args.batch, 4096)
    gather = ObsGather(B, 26, torch.device("cpu"))
    local = torch.full((B, 26), float(rank))
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = gather(local)
    dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, -dt], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)   # (max, -min) over ranks in one reduction, as the GPU path does
    ok = bool(torch.equal(full.view(world, B, 26)[:, 0, 0], torch.arange(world, dtype=torch.float32)))
    if rank == 0:
        print(json.dumps({"metric": "dry-gather rehearsal (no GPU, no physics)", "value": world * B * args.steps / float(t[0].item()), "unit": "rows/s",
                          "n_gpus": world, "steps": args.steps, "warmup": 0, "data": "dry-run", "gather_ok": ok,
                          "per_rank": {"ms_per_step_max": 1e3 * float(t[0]) / args.steps, "ms_per_step_min": -1e3 * float(t[1]) / args.steps}}))
    dist.destroy_process_group()
    return 0 if ok else 1


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus), file=sys.stderr)
        sys.exit(2)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.dry_gather:
        sys.exit(dry_rank(args, world, rank))
    if args.drift_gpu_leg:
        sys.exit(drift_gpu_leg(args))
    if args.env_drift_gpu_leg:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import env_drift
        env_drift.gpu_leg(args.env_drift_gpu_leg, ENV_DRIFT_ENVS, ENV_DRIFT_STEPS)
        sys.exit(0)
    small = small_action_runs(args) if (args.level == "env" and world == 1 and args.extra_scales) else {}
    pol_leg = policy_run(args) if (args.level == "env" and world == 1) else None
    cfg_legs = config_legs(args) if (world == 1 and args.config_legs) else {}
    drift_npz = None
    env_drift_res = None
    if world == 1 and not args.no_cpu_baseline and args.level == "env":
        env_drift_res = env_level_drift(args)
    if world == 1 and not args.no_cpu_baseline:   # the HIP leg of the drift metric, in a child process before this one touches the GPU
        drift_npz = os.path.join(ROOT, "gpurun_out", "bench_drift_gpu_%d.npz" % os.getpid())
        os.makedirs(os.path.dirname(drift_npz), exist_ok=True)
        try:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--drift-gpu-leg", drift_npz, "--model", args.model], timeout=600, check=True,
                           stdout=subprocess.DEVNULL)
        except Exception as e:
            print("bench.py: drift GPU leg failed: %r" % (e,), file=sys.stderr)
            drift_npz = None

    import numpy as np
    import torch
    import torch.distributed as dist
    from mujoco_jaco_amd import workload
    from mujoco_jaco_amd.modelc import blob
    from mujoco_jaco_amd.physics import BatchedMujoco

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    B = args.batch
    fs = args.frame_skip if args.frame_skip is not None else (50 if args.level == "env" else 1)
    M = blob.load(os.path.join(ROOT, "mujoco_jaco_amd", "assets", args.model + ".jacomdl"))
    done_count = torch.zeros((), dtype=torch.int64, device=dev)
    if args.level == "env":
        from mujoco_jaco_amd.env import JacoBatchedEnv
        from mujoco_jaco_amd.sharding import ObsGather, env_seed
        genv = JacoBatchedEnv(num_envs=B, device=local_rank, frame_skip=fs, seed=env_seed(1000, rank), task=args.task, robot_file=args.model,
                              auto_reset=not (args.explicit_reset or args.no_reset))
        env = genv.sim
        if args.hints is not None:
            env.set_option("hints", args.hints)
        genv.reset()
        gen = torch.Generator(device=dev); gen.manual_seed(2000 + rank)
        # a rollout in progress: episode ages spread over [0, task_max_steps) so time-outs (and their resets) arrive at the
        # steady-state rate instead of all together 700 steps after the global reset
        ts = genv.task_state()
        ts[:, 1] = torch.randint(0, genv.task_max_steps, (B,), device=dev, generator=gen).float()
        genv.set_task_state(ts)
        gather = ObsGather(B, 26, dev) if world > 1 else None
        scale = [float(args.action_scale)]
        nact = genv.action_space.shape[0]

        abuf = torch.empty(B, nact, device=dev)
        done_each = torch.zeros(B, dtype=torch.int32, device=dev)
        pol, cur_obs = None, [None]
        if args.policy:
            from mujoco_jaco_amd.policy import HPCPolicy
            path = args.policy if os.path.exists(args.policy) else os.path.join(ROOT, "tests", "golden", "policy_%s.npz" % args.policy)
            pol = HPCPolicy.load(path, device=dev)
            cur_obs[0] = genv.make_observation()

        def step():
            if pol is not None:
                a, _ = pol.predict(cur_obs[0])
            else:
                a = abuf.uniform_(-scale[0], scale[0], generator=gen)   # one launch: U(-1, 1)^nact x action scale
            o, r, d, _ = genv.step(a)
            done_each.add_(d)   # (one elementwise launch; summed after the timed window)
            if not args.no_reset and not genv.auto_reset:
                o = genv.reset(d)  # masked jaco_reset of the finished envs: no host sync; their obs rows become the new episode's first
            # (default: option auto_reset -- the same reset, done inside jaco_step by the wave that finished the episode)
            cur_obs[0] = o
            if world > 1:          # one collective per rollout step: concatenate the observation rows of all shards (issued on a
                gather.start(o)    # side stream: it overlaps the next step's launch set; the timed region ends on a full device sync)
    else:
        env = BatchedMujoco(B, robot_file=args.model, device=local_rank, frame_skip=fs, seed=rank)
        dual = env.nu == 18   # jaco2_dual_torque (two arms, two objects; the d30 build)
        q = torch.tensor(workload.reset_states_dual(M["qpos0"], B, seed=1000 + rank) if dual else workload.reset_states(M["qpos0"], B, seed=1000 + rank),
                         dtype=torch.float32, device=dev)
        ctrl = torch.zeros(B, env.nu, dtype=torch.float32, device=dev)
        env.set_state(q, None, None)
        if args.no_contact:
            env.set_option("disable_contact", 1)
        gathered = torch.empty(world * B, env.nq, device=dev) if world > 1 else None

        # "random-action rollouts": fresh motor torques U(-1, 1) x 0.2 x force range every launch (one launch = `fs` substeps), fingers held at
        # 0.6 -- the same distribution as workload.random_ctrl, drawn on the GPU.  (Constant torques for the whole run, as rounds 1-3 had it,
        # spin the limit-free arm joints up to hundreds of rad/s within the 11 000 substeps of the config-2 leg: a non-physical workload
        # that ended in the velocity quarantine for some envs.)
        arm = [30, 30, 30, 15, 15, 15, 0, 0, 0]
        tq = torch.tensor((arm * 2)[:env.nu], dtype=torch.float32, device=dev) * 0.2
        hold = torch.tensor(([0] * 6 + [0.6] * 3) * 2, dtype=torch.float32, device=dev)[:env.nu]
        cgen = torch.Generator(device=dev); cgen.manual_seed(3000 + rank)

        def step():
            ctrl.copy_((torch.rand(B, env.nu, device=dev, generator=cgen) * 2 - 1) * tq + hold)
            env.send_forces(ctrl, nsub=fs)
            if world > 1:
                qpos, _, _ = env.state_views()
                dist.all_gather_into_tensor(gathered, qpos)

    for kv in args.set_option:
        name, val = kv.split("=")
        env.set_option(name, float(val))

    per_rank, t_loop = {}, [0.0]

    def timed(nsteps):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        torch.cuda.synchronize()
        t_loop[0] = time.perf_counter() - t0   # this rank's own steps, before it waits for the others
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:   # MAX over ranks is the job's time; the spread (and each rank's own loop time, before the closing barrier) explains it
            t = torch.tensor([dt, -dt, t_loop[0], -t_loop[0]], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            per_rank.update({"ms_per_step_max": 1e3 * float(t[0]) / nsteps, "ms_per_step_min": -1e3 * float(t[1]) / nsteps,
                             "own_loop_ms_per_step_max": 1e3 * float(t[2]) / nsteps, "own_loop_ms_per_step_min": -1e3 * float(t[3]) / nsteps})
            dt = float(t[0].item())
        return dt

    for _ in range((args.preroll if args.level == "env" else 0) + args.warmup):
        step()
    done_count.zero_()
    if args.level == "env":
        done_each.zero_()
    env.launch_count()   # (reset the kernel-launch counter)
    env.enable_timing(True)
    if world > 1 and args.level == "env":
        gather.enable_timing(True)
    dt = timed(args.steps)
    gather_ms = gather.gather_time_ms() if (world > 1 and args.level == "env") else None
    step_ms = env.step_time_ms()
    kern_ms, launches = env.kernel_time_ms()
    env.enable_timing(False)
    if args.level == "env":
        done_count.add_(done_each.sum())
    done_fraction = float(done_count.item()) / (B * args.steps)
    launches_per_step = env.launch_count() / float(args.steps)
    flags = int(np.bitwise_or.reduce(env.flags().cpu().numpy().astype(np.uint32)))   # OR over the batch (a max would let a big informational bit hide a small error bit)
    stats = env.stats().float().mean(0).cpu().numpy()
    heavy = float(((env.flags() & 32) != 0).float().mean().item())
    # flag census (untimed, after the timed window, same regime): env steps in which contacts / rows beyond the last capacity tier were
    # dropped (bits 1 | 2 | 4), the state went non-finite (8) or the solver hit its iteration cap (16), counted per step
    flagged = None
    if args.flag_census > 0 and args.level == "env":
        env.clear_flags()
        nfl = torch.zeros((), dtype=torch.int64, device=dev)
        for _ in range(args.flag_census):
            step()
            nfl += ((env.flags() & 31) != 0).sum()
            env.clear_flags()
        flagged = 1e6 * float(nfl.item()) / (B * args.flag_census)

    if rank == 0:
        # algorithmic HBM bytes per env per launch.  SURVEY 8(d): A = 4 (2 nq + 2 nv + n_act + n_obs + 1) + 1 = 489 B for the env
        # step of the full model (state in, state + obs + reward + done out); what this implementation actually has to move
        # in addition: warm start in/out, sensordata, the task row and the controller-cache row in and out.
        if args.level == "env":
            a_survey = 4 * (2 * env.nq + 2 * env.nv + 7 + 26 + 1) + 1
            a_impl = 4 * (4 * env.nq + 6 * env.nv + env.nsensor + 7 + 26 + 1 + 2 * env.L.jaco_task_row_floats() + 2 * 104) + 1   # (state as hi + lo pairs)
        else:
            a_survey = 4 * (2 * env.nq + 2 * env.nv + env.nu)
            a_impl = 4 * (4 * env.nq + 6 * env.nv + env.nu + env.nsensor)
        achieved = a_survey * B / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        # PMC counters cannot be collected inside this run (rocprofv3 --pmc passes are separate runs of the same command,
        # tools/profile_pmc.sh): what is reported is a REPLAY of the committed pass, named as such, with the commit and the kernel
        # time it was captured at -- `traffic` is that pass's fabric bytes per launch
        traffic, replay = None, None
        try:
            pmc_all = json.load(open(PMC_FILE))
            pm = pmc_all.get("%s_B%d_fs%d" % (args.level, B, fs), {})
            traffic = pm.get("hbm_bytes_per_launch")
            replay = {"file": os.path.relpath(PMC_FILE, ROOT), "commit": pmc_all.get("commit"), "kernel_ms": pm.get("kernel_ms"), "hbm_bytes_per_launch": traffic,
                      "valu_issue_frac": (pm["SQ_INSTS_VALU"] / (pm["kernel_ms"] * 1e-3) / VALU_ISSUE_PEAK) if pm.get("SQ_INSTS_VALU") and pm.get("kernel_ms") else None,
                      "wait_frac": (pm["SQ_WAIT_ANY"] / pm["SQ_WAVE_CYCLES"]) if pm.get("SQ_WAIT_ANY") and pm.get("SQ_WAVE_CYCLES") else None}
        except Exception:
            pass
        cfgname = "config2" if (args.no_contact and args.level == "ctrl") else ("config4" if (args.no_reset and fs == 4) else ("two-arm model, sim tier" if env.nu == 18 else
                                                                         ("sibling model, sim tier" if (args.level == "ctrl" and args.model != "jaco2_curtain_torque") else "config3")))
        if args.level != "env":
            workload_desc = "random motor ctrl, ctrl-level jaco_physics_step"
        else:
            source = ("actions from the reference's shipped '%s' policy (deterministic forward pass inside the timed step)" % args.policy) if args.policy \
                else "fresh random actions x %g each step" % args.action_scale
            workload_desc = source + ", env-level jaco_step (OSC + %d substeps + obs/reward/done) + masked reset of finished envs" % fs
        out = {
            "metric": "env-steps/sec at batch 65 536 (full Jaco + gripper + contacts)",
            "value": world * B * args.steps / dt, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d envs/GPU, %s, %s reset distribution, %s" % (cfgname, B, args.model, args.task, workload_desc),
                       "level": args.level, "action_scale": args.action_scale, "preroll_steps": args.preroll if args.level == "env" else 0,
                       "envs_per_gpu": B, "frame_skip": fs, "substeps_per_s": world * B * args.steps * fs / dt,
                       "done_fraction": done_fraction,
                       "small_action_env_steps_per_s": small, "policy_driven": pol_leg,
                       "sharding": "independent env shards per rank" + ("; one all_gather of [B,%d] f32 rows per step" % (26 if args.level == "env" else env.nq) if world > 1 else ""),
                       "collective": ({"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank0_device": torch.cuda.get_device_name(local_rank),
                                       "devices_visible": torch.cuda.device_count(), "overlapped_on_side_stream": args.level == "env"} if world > 1 else None),
                       "mean_contacts": float(stats[0]), "mean_rows": float(stats[1]), "mean_newton_iters": float(stats[2]),
                       "reset": ("none (finished envs frozen)" if args.no_reset else ("inside jaco_step (auto_reset)" if (args.level == "env" and genv.auto_reset) else "masked jaco_reset launch chain after every step")),
                       "error_flags_or": flags & 31, "info_flags_or": flags & ~31, "launches_per_step": launches_per_step,
                       "flagged_env_steps_per_million": flagged, "flag_census_steps": args.flag_census if flagged is not None else 0,
                       "side_leg_flags": LEG_FLAGS,
                       "per_rank": (dict(per_rank, gather_ms_mean=gather_ms[0] if gather_ms else None, gather_ms_max=gather_ms[1] if gather_ms else None,
                                         gather_note="device time of the all_gather on its side stream (rank 0's view); it overlaps the next step's launch set") if world > 1 else None),
                       "config2_env_steps_per_s": (cfg_legs.get("config2") or {}).get("env_steps_per_s"), "config4_env_steps_per_s": (cfg_legs.get("config4") or {}).get("env_steps_per_s"),
                       "config_legs": cfg_legs,
                       "heavy_tier_fraction": heavy},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "jaco_physics_kernel", "kernel_ms": kern_ms, "launches": launches, "step_launch_set_ms": step_ms,
                         "algorithmic_bytes_per_env_launch": a_survey, "implementation_bytes_per_env_launch": a_impl,
                         "pmc_replayed": replay,
                         "note": "latency/occupancy-bound by design (SURVEY 8d): algorithmic traffic is ~0.5 KB per env per launch, so the HBM fraction is ~1e-4 whatever the kernel does; the meaningful ceilings are VALU issue (wave-instructions issued / 1.23e12 per s) and exposed latency (SQ_WAIT_ANY / SQ_WAVE_CYCLES): `pmc_replayed` holds them, and `traffic`, from the committed separate --pmc passes of this command (not measured in this run; see its commit / kernel_ms)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["drift"] = cpu_baseline_and_drift(args.model, fs, drift_npz)
            if out["drift"] is not None:
                out["drift"]["env_level_closed_loop"] = env_drift_res
            if drift_npz and os.path.exists(drift_npz):
                os.remove(drift_npz)
        print(json.dumps(out))
    env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""GPU diagnostics, one file, one subcommand each (round 4: the former one-off tools/gpu_*.py probes):
  python tools/gpu_diag.py <subcommand> [args...]        subcommands: tier-causes, first-contact, parity, touch, async-modes, tail-trace
Each subcommand keeps the argument conventions of the probe it came from (its docstring is the first line of its function).
Diagnostics only: some of them use the oracle (tests/) as the checker; nothing here is on the product path."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))


def cmd_tier_causes():
    """Why envs leave each capacity tier (diagnostic): per-step histogram of the informational bail-cause flags."""

    import os, sys
    import numpy as np, torch
    from mujoco_jaco_amd.env import JacoBatchedEnv
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    scale = float(sys.argv[sys.argv.index("--action-scale") + 1]) if "--action-scale" in sys.argv else 1.0
    env = JacoBatchedEnv(num_envs=B, seed=1000, task="picking")
    env.reset()
    gen = torch.Generator(device=env.device); gen.manual_seed(2000)
    for step in range(n):
        a = (torch.rand(B, 7, device=env.device, generator=gen) * 2 - 1) * scale
        env.sim.clear_flags()
        obs, r, d, _ = env.step(a)
        st = env.sim.stats().cpu().numpy(); fl = env.sim.flags().cpu().numpy().astype(np.uint32)
        out = []
        for t, name in enumerate(("light", "medium", "heavy")):
            c = (fl >> (8 + 3 * t)) & 7
            out.append("%s: contacts %.1f%% rows %.1f%% candidates %.1f%%" % (name, 100 * ((c & 1) != 0).mean(), 100 * ((c & 2) != 0).mean(), 100 * ((c & 4) != 0).mean()))
        print("step %d  " % step + " | ".join(out))
        print("        end of step: contacts pct50 %d pct90 %d max %d  rows pct50 %d pct90 %d max %d" % (
            np.percentile(st[:, 0], 50), np.percentile(st[:, 0], 90), st[:, 0].max(), np.percentile(st[:, 1], 50), np.percentile(st[:, 1], 90), st[:, 1].max()))


def cmd_first_contact():
    """First GPU run: stage-by-stage parity of the HIP kernel against the oracle, then a throughput probe."""

    import os, sys, time
    import numpy as np
    import torch
    from oracle_binding import Oracle
    from mujoco_jaco_amd.physics import BatchedMujoco
    from mujoco_jaco_amd.modelc import blob

    contact = "--contact" in sys.argv
    B = 64
    env = BatchedMujoco(B)
    if not contact:
        env.set_option("disable_contact", 1)
    o = Oracle()
    if not contact:
        o.option("disable_contact", 1)
    M = blob.load(os.path.join(os.path.dirname(__file__), "..", "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.jacomdl"))
    rng = np.random.default_rng(0)
    q = np.tile(M["qpos0"], (B, 1)); v = np.zeros((B, 21)); ctrl = np.zeros((B, 9))
    for e in range(B):
        q[e, :6] = [rng.uniform(.7, 2.5), rng.uniform(3.8, 4), rng.uniform(1, 1.7), rng.uniform(1.8, 2.5), rng.uniform(1, 2.5), rng.uniform(.8, 2.3)]
        q[e, 6:9] = rng.uniform(0.0, 1.5, 3)
        q[e, 9:12] = [rng.uniform(-.1, .1), .65 + rng.uniform(-.08, .02), .1898 if contact else 0.5]
        q[e, 16:18] = [.4 + rng.uniform(-.05, .05), .3 + rng.uniform(-.05, .05)]
        v[e, :9] = rng.normal(size=9) * 0.2
        ctrl[e] = np.concatenate([rng.uniform(-1, 1, 6) * [30, 30, 30, 15, 15, 15] * 0.3, rng.uniform(0, 1.5, 3)])
    dev = env.device
    tq = torch.tensor(q, dtype=torch.float32, device=dev); tv = torch.tensor(v, dtype=torch.float32, device=dev)
    tc = torch.tensor(ctrl, dtype=torch.float32, device=dev)
    env.set_state(tq, tv, torch.zeros_like(tv))
    roots = [b for b in range(1, int(M["nbody"][0])) if M["body_weldid"][b] == b]
    D = env.send_forces_debug(tc, 3, nsub=1)
    o.set("qpos", q[3]); o.set("qvel", v[3]); o.set("ctrl", ctrl[3]); o.set("qacc_warmstart", np.zeros(21)); o.forward()
    print("xpos err", np.abs(D[0:33].reshape(11, 3) - o.get("xpos").reshape(-1, 3)[roots]).max())
    off = 33 + 99
    Mo = o.get("qM").reshape(21, 21); print("M rel err", np.abs(D[off:off + 441].reshape(21, 21) - Mo).max() / np.abs(Mo).max()); off += 441
    for nm in ["qfrc_bias", "qfrc_smooth", "qacc_smooth", "qacc", "qfrc_constraint"]:
        a = o.get(nm); b = D[off:off + 21]; print(nm, "abs err", np.abs(a - b).max(), "max", np.abs(a).max()); off += 24
    print("stats gpu", D[off:off + 4], "oracle ncon/nefc/iter", o.ncon, o.nefc, o.solver_iter)
    # multi-step drift, all envs
    nstep = 200
    qo, vo, wo = q.copy(), v.copy(), np.zeros((B, 21))
    o.step_batch(qo, vo, wo, np.ascontiguousarray(ctrl), nsub=nstep + 1, nthreads=8)
    env.send_forces(tc, nsub=nstep)
    gq, gv, gw = [t.cpu().numpy() for t in env.get_state()]
    print("after", nstep + 1, "steps: max qpos err", np.abs(gq - qo).max(), "arm+finger", np.abs(gq - qo)[:, :9].max(), "qvel err", np.abs(gv - vo).max())
    print("flags", env.flags().cpu().numpy().max(), "stats max", env.stats().cpu().numpy().max(0))
    # throughput probe
    for Bn in (4096, 65536):
        e2 = BatchedMujoco(Bn)
        if not contact:
            e2.set_option("disable_contact", 1)
        c2 = torch.zeros(Bn, 9, device=dev); c2[:, 6:] = 0.6
        if contact:
            q2 = torch.tensor(np.tile(q, (Bn // B, 1)), dtype=torch.float32, device=dev)
            e2.set_state(q2, None, None)
        e2.send_forces(c2, nsub=2); torch.cuda.synchronize()
        t = time.time(); e2.send_forces(c2, nsub=20); torch.cuda.synchronize(); dt = time.time() - t
        print("B", Bn, "substeps/s", Bn * 20 / dt, "us per substep-wave (1 wave)", dt / 20 / (Bn / (256 * 6)) * 1e6, "stats max", e2.stats().cpu().numpy().max(0), "flags", int(e2.flags().max()))
        e2.close()


def cmd_parity():
    """parity"""
    import os, sys
    import numpy as np, torch
    from mujoco_jaco_amd import workload
    from mujoco_jaco_amd.modelc import blob
    from mujoco_jaco_amd.physics import BatchedMujoco
    from oracle_binding import Oracle
    M = blob.load(os.path.join(ROOT, "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.jacomdl"))
    B = 512
    q = workload.reset_states(M["qpos0"], B, seed=21); c = workload.random_ctrl(B, seed=22, scale=0.2)
    o = Oracle()
    v = np.zeros((B, 21)); w = np.zeros((B, 21))
    o.step_batch(q, v, w, np.ascontiguousarray(c), nsub=12, nthreads=16)
    env = BatchedMujoco(B); dev = env.device
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    env.set_state(t(q), t(v), t(w)); env.send_forces(t(c), nsub=1)
    gq, gv, gw = [x.cpu().numpy().astype(np.float64) for x in env.get_state()]
    st = env.stats().cpu().numpy(); fl = env.flags().cpu().numpy()
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    err = np.zeros(B); info = []
    for e in range(B):
        o.set("qpos", f32(q[e])); o.set("qvel", f32(v[e])); o.set("qacc_warmstart", f32(w[e])); o.step(f32(c[e]))
        err[e] = np.abs(gq[e] - o.get("qpos")).max()
        info.append((o.ncon, o.nefc, o.solver_iter, np.abs(gw[e] - o.get("qacc_warmstart")).max(), np.abs(o.get("qacc_warmstart")).max()))
    order = np.argsort(-err)
    print("median %.2e  p90 %.2e p99 %.2e" % (np.median(err), *np.percentile(err, [90, 99])))
    for e in order[:14]:
        print("env %3d err %.2e gpu ncon/nefc/it/cand %s flags %d | oracle ncon/nefc/it %s qacc err %.3g of %.3g" % (e, err[e], st[e], fl[e], info[e][:3], info[e][3], info[e][4]))


def cmd_touch():
    """touch"""
    import os, sys
    import numpy as np, torch
    from mujoco_jaco_amd.modelc import blob, rot
    from mujoco_jaco_amd.physics import BatchedMujoco
    from oracle_binding import Oracle
    M = blob.load(os.path.join(ROOT, "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.jacomdl"))
    names = {}
    for line in open(os.path.join(ROOT, "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.names.txt")):
        k, v = line.strip().split(": ", 1); names[k] = v.split()
    o = Oracle()
    q = M["qpos0"].copy()
    q[:6] = [1.3, 3.85, 1.05, 2.05, 1.5, -1.15]; q[6:9] = 0.6; q[16:18] = [.4, .3]
    o.set("qpos", q); o.forward()
    b = names["body"].index("EE_obj")
    xp = o.get("xpos").reshape(-1, 3)[b]; xq = o.get("xquat").reshape(-1, 4)[b]
    q[9:12] = xp + rot.quat_to_mat(xq) @ np.array([-0.04, 0, 0]); q[12:16] = xq
    B = 4
    env = BatchedMujoco(B); dev = env.device
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    C = np.tile(np.array([0, 0, 0, 0, 0, 0, .8, .8, .8]), (B, 1))
    o.reset(); o.set("qpos", q.astype(np.float32).astype(np.float64))
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    for i in range(6):
        st = [f32(o.get(n)) for n in ("qpos", "qvel", "qacc_warmstart")]
        for n, x in zip(("qpos", "qvel", "qacc_warmstart"), st): o.set(n, x)
        env.set_state(*[t(np.tile(x, (B, 1))) for x in st])
        D = env.send_forces_debug(t(C), 0, nsub=1)
        o.step(C[0])
        if i == 3:
            off = 33 + 99 + 441 + 5 * 24
            nc = int(D[off]); Cg = D[off + 4:off + 4 + 8 * nc].reshape(nc, 8); oc = o.get("contact").reshape(-1, 11)
            gnm = names["geom"]
            for k in range(nc):
                if "thumb_proximal_plane" in (gnm[int(oc[k, 7])], gnm[int(oc[k, 8])]):
                    print("   contact", k, gnm[int(oc[k, 7])], gnm[int(oc[k, 8])], "gpu dist %.6f pos %s n %s" % (Cg[k, 0], Cg[k, 1:4].round(5), Cg[k, 4:7].round(4)),
                          "| oracle dist %.6f pos %s n %s" % (oc[k, 0], oc[k, 1:4].round(5), oc[k, 4:7].round(4)))
            E = D[off + 4 + 8 * 64: off + 4 + 8 * 64 + 4 * o.nefc].reshape(-1, 4)
            print("   force err", np.abs(E[:, 3] - o.get("efc_force")).max())
            base = off + 4 + 8 * 64 + 4 * 256 + 3 * 64
            print("   gpu c_fn", D[base:base + nc].round(2)); print("   gpu sens (dump)", D[base + 64:base + 84].round(2))
            f = o.get("efc_force"); print("   oracle fn", np.array([f[int(r[10]):int(r[10]) + 2 * (int(r[9]) - 1)].sum() for r in oc]).round(2))
        s = env.sensordata().cpu().numpy()[0]; so = o.get("sensordata")
        gq = env.get_state()[0].cpu().numpy()[0]
        print(i, "qpos err %.2e" % np.abs(gq - o.get("qpos")).max(), "stats", env.stats().cpu().numpy()[0], "oracle", o.ncon, o.nefc)
        print("   gpu   ", s.round(2)); print("   oracle", so.round(2))


def cmd_async_modes():
    """Mean step time of the bench loop (fresh random actions, masked resets, no host sync inside the loop) under different execution options, with and without the HIP timing events (diagnostic)."""

    import os, sys, time
    import torch
    from mujoco_jaco_amd.env import JacoBatchedEnv
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    genv = JacoBatchedEnv(num_envs=B, device=0, frame_skip=50, seed=1000, task="picking")
    env = genv.sim
    dev = genv.device
    genv.reset()
    gen = torch.Generator(device=dev); gen.manual_seed(2000)
    ts = genv.task_state(); ts[:, 1] = torch.randint(0, 700, (B,), device=dev, generator=gen).float(); genv.set_task_state(ts)
    def run(k):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(k):
            o, r, d, _ = genv.step(torch.rand(B, 7, device=dev, generator=gen) * 2 - 1); genv.reset(d)
        torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e3
    run(30)
    for name, opts, timing in (("default, timing off", {}, False), ("default, timing on", {}, True), ("default, timing off", {}, False),
                               ("concurrent_heavy 0, timing off", {"concurrent_heavy": 0}, False), ("hints 0, timing off", {"hints": 0}, False),
                               ("hints 1, timing off", {"hints": 1}, False), ("default, timing on", {}, True)):
        for k, v in (("concurrent_heavy", 1), ("hints", 2)): env.set_option(k, v)
        for k, v in opts.items(): env.set_option(k, v)
        env.enable_timing(timing)
        run(5)
        ms = run(n)
        extra = ""
        if timing:
            st = env.step_time_ms(); km, _ = env.kernel_time_ms(); extra = "  launch set %.2f light kernel %.2f" % (st, km)
        env.enable_timing(False)
        print("%-34s %.2f ms/step%s" % (name, ms, extra), flush=True)


def cmd_tail_trace():
    """Per-step tail of the step launch set behind the light kernel (diagnostic): launch-set time minus light-kernel time, with what the tier queues held and how much of it the resident workers took."""

    import ctypes, os, sys
    import torch
    from mujoco_jaco_amd.env import JacoBatchedEnv
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    genv = JacoBatchedEnv(num_envs=B, device=0, frame_skip=50, seed=1000, task="picking")
    env = genv.sim
    dev = genv.device
    genv.reset()
    gen = torch.Generator(device=dev); gen.manual_seed(2000)
    ts = genv.task_state(); ts[:, 1] = torch.randint(0, 700, (B,), device=dev, generator=gen).float(); genv.set_task_state(ts)
    for _ in range(20):
        o, r, d, _ = genv.step(torch.rand(B, 7, device=dev, generator=gen) * 2 - 1); genv.reset(d)
    qw = (ctypes.c_int * 32)()
    for i in range(n):
        a = torch.rand(B, 7, device=dev, generator=gen) * 2 - 1
        env.enable_timing(True)
        o, r, d, _ = genv.step(a)
        torch.cuda.synchronize()
        st = env.step_time_ms(); km, _ = env.kernel_time_ms()
        env.L.jaco_debug_queue_words(env.h, qw, 32)
        env.enable_timing(False)
        genv.reset(d)
        print("step %2d launch set %.3f ms light kernel %.3f ms tail %.3f | queued %s by workers %s resets %d" % (
            i, st, km, st - km, list(qw[0:3]), list(qw[3:6]), int(d.sum().item())))


COMMANDS = {"tier-causes": cmd_tier_causes, "first-contact": cmd_first_contact, "parity": cmd_parity, "touch": cmd_touch, "async-modes": cmd_async_modes, "tail-trace": cmd_tail_trace}

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in COMMANDS:
        print(__doc__); sys.exit(2)
    cmd = sys.argv.pop(1)
    COMMANDS[cmd]()

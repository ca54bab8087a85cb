"""GPU box diagnostic: the scenario of tests/test_gpu_env.py::test_auto_reset_equals_step_plus_masked_reset_8192_envs, reporting WHERE the
in-kernel reset and the explicit reset chain differ (env, step, tensor, columns, flags, step counters) instead of asserting."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mujoco_jaco_amd.env import JacoBatchedEnv
B = 8192
outs = []
for auto in (True, False, True):
    env = JacoBatchedEnv(num_envs=B, task="picking", seed=31, auto_reset=auto, frame_skip=10)
    for kv in os.environ.get("JACO_OPTS", "").split(","):
        if kv:
            env.sim.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    env.reset()
    gen = torch.Generator(device=env.device); gen.manual_seed(3)
    t = env.task_state(); t[:, 1] = torch.randint(693, 699, (B,), device=env.device, generator=gen).float(); env.set_task_state(t)
    rec = []
    save_env = int(os.environ.get("JACO_SAVE_ENV", "-1"))
    saved = {}
    for s in range(6):
        a = torch.rand(B, 7, device=env.device, generator=gen) * 2 - 1
        if save_env >= 0:   # everything needed to replay this env's step elsewhere (e.g. on the host emulator)
            st = env.sim.get_state()
            saved["step%d" % s] = {"qpos": st[0][save_env].cpu().numpy(), "qvel": st[1][save_env].cpu().numpy(), "qacc_ws": st[2][save_env].cpu().numpy(),
                                   "task": env.task_state()[save_env].cpu().numpy(), "marker": env.markers()[save_env].cpu().numpy(), "action": a[save_env].cpu().numpy()}
        o, r, d, _ = env.step(a)
        r, d = r.clone(), d.clone()
        if not auto:
            o = env.reset(d)
        rec.append((o.clone().cpu(), r.cpu(), d.cpu(), env.sim.get_state()[0].clone().cpu(), env.task_state().clone().cpu(), env.sim.flags().clone().cpu(), env.sim.stats().clone().cpu(), env.sim.sensordata().clone().cpu()))
    outs.append(rec)
    if save_env >= 0:
        import numpy as np
        np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "autoreset_env%d_auto%d_%d.npz" % (save_env, int(auto), len(outs))),
                 **{"%s_%s" % (k, kk): vv for k, v in saved.items() for kk, vv in v.items()},
                 sens=torch.stack([r[7][save_env] for r in rec]).numpy(), obs=torch.stack([r[0][save_env] for r in rec]).numpy())
    env.close()
names = ["obs", "reward", "done", "qpos", "task"]
for label, (A, Bm) in (("auto vs explicit", (outs[0], outs[1])), ("auto vs auto (second run)", (outs[0], outs[2]))):
    print("==", label)
    for s, (x, y) in enumerate(zip(A, Bm)):
        for k in range(5):
            if not torch.equal(x[k], y[k]):
                diff = (x[k].float() - y[k].float()).abs()
                envs = torch.nonzero(diff.reshape(B, -1).max(1).values > 0).flatten()
                print("step %d %s: %d envs differ, max %.3e" % (s + 1, names[k], len(envs), diff.max()))
                for e in envs[:6].tolist():
                    cols = torch.nonzero(diff.reshape(B, -1)[e] > 0).flatten().tolist()
                    print("   env %d cols %s  x %s  y %s | done x/y %d/%d  flags x 0x%x y 0x%x  stats x %s y %s  steps x %g y %g  prev-step done %d" % (
                        e, cols[:8], x[k].reshape(B, -1)[e][cols[:4]].tolist(), y[k].reshape(B, -1)[e][cols[:4]].tolist(), int(x[2][e]), int(y[2][e]),
                        int(x[5][e]), int(y[5][e]), x[6][e].tolist(), y[6][e].tolist(), float(x[4][e, 1]), float(y[4][e, 1]), int(A[s - 1][2][e]) if s else -1))
                    sx, sy = x[7][e], y[7][e]
                    nz = torch.nonzero((sx != 0) | (sy != 0)).flatten().tolist()
                    print("      sensordata (index: auto / explicit):", {i: (float(sx[i]), float(sy[i])) for i in nz}, "| bitwise equal:", bool(torch.equal(sx, sy)))
print("done")

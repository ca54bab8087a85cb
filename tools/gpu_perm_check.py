import sys, torch
sys.path.insert(0, "/root/repo")
from mujoco_jaco_amd.env import JacoBatchedEnv
import os
B = int(os.environ.get('PERM_B', 8192)); SCALE = float(os.environ.get('PERM_SCALE', 1.0))
def run(schedule, conc=1):
    env = JacoBatchedEnv(num_envs=B, task="picking", seed=21)
    env.sim.set_option("schedule", schedule); env.sim.set_option("concurrent_heavy", conc)
    env.reset()
    gen = torch.Generator(device=env.device); gen.manual_seed(5)
    for s in range(4):
        a = (torch.rand(B, 7, device=env.device, generator=gen) * 2 - 1) * SCALE
        obs, rew, done, _ = env.step(a)
    q, v, _ = env.sim.get_state()
    return q.clone(), obs.clone(), env.sim.flags().clone(), v.clone(), rew.clone(), done.clone()
ref = run(0, 0)
for name, args in (("sched0 conc0 again", (0, 0)), ("sched0 conc1", (0, 1)), ("sched1 conc0", (1, 0)), ("sched1 conc1", (1, 1))):
    out = run(*args)
    dq = (out[0] - ref[0]).abs().max(1).values
    bad = (dq > 0).nonzero().flatten()
    for nm, k in (("obs", 1), ("qvel", 3), ("rew", 4), ("done", 5)):
        d = (out[k].float() - ref[k].float()).abs().reshape(B, -1).max(1).values
        if (d > 0).any(): print("   ", nm, "differs in", int((d > 0).sum()), "envs, max", float(d.max()), "first idx", (d > 0).nonzero().flatten()[:5].tolist(), "obs cols", (out[1] - ref[1]).abs().max(0).values.nonzero().flatten().tolist() if nm == "obs" else "")
    print(name, "envs differing:", len(bad), "max diff", float(dq.max()), "flags of differing envs:", ref[2][bad][:8].tolist(), "idx", bad[:8].tolist())

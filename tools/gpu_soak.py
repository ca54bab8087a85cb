"""Soak run: many env steps with random actions and masked resets of finished envs; reports flags, finiteness, episode
statistics and throughput (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mujoco_jaco_amd.env import JacoBatchedEnv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
task = sys.argv[3] if len(sys.argv) > 3 else "picking"
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0   # actions are U(-1, 1) times this
auto = "--auto-reset" in sys.argv                            # reset inside jaco_step (option auto_reset) instead of a masked jaco_reset per step
env = JacoBatchedEnv(num_envs=B, task=task, seed=7, auto_reset=auto)
env.reset()
gen = torch.Generator(device=env.device); gen.manual_seed(1)
ndone = nsucc = 0
bad = 0
torch.cuda.synchronize(); t0 = time.perf_counter()
for s in range(n):
    a = (torch.rand(B, env.action_space.shape[0], device=env.device, generator=gen) * 2 - 1) * scale
    obs, rew, done, _ = env.step(a)
    bad += int((~torch.isfinite(obs).all(1)).sum()) + int((~torch.isfinite(rew)).sum())
    if done.any():
        ndone += int(done.sum()); nsucc += int(env.success_flags()[done].sum()) if hasattr(env, "success_flags") else 0
        if not env.auto_reset:
            env.reset(done)
    if s % 50 == 49:
        fl = env.sim.flags()
        print("step %4d  %.1f ms/step  episodes ended %d  non-finite %d  flags: nan %d con-ovf %d efc-ovf %d cand-ovf %d maxiter %d" % (
            s + 1, (time.perf_counter() - t0) / (s + 1) * 1e3, ndone, bad, int(((fl & 8) != 0).sum()), int(((fl & 1) != 0).sum()),
            int(((fl & 2) != 0).sum()), int(((fl & 4) != 0).sum()), int(((fl & 16) != 0).sum())), flush=True)
torch.cuda.synchronize()
print("done: %d steps x %d envs in %.1f s -> %.0f env-steps/s (resets included)" % (n, B, time.perf_counter() - t0, n * B / (time.perf_counter() - t0)))

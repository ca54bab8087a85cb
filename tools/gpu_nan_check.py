"""How often do reward / observation come out non-finite under random actions (diagnostic)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mujoco_jaco_amd.env import JacoBatchedEnv
B = 65536
env = JacoBatchedEnv(num_envs=B, task="picking", seed=21)
env.reset()
gen = torch.Generator(device=env.device); gen.manual_seed(5)
for s in range(12):
    a = torch.rand(B, 7, device=env.device, generator=gen) * 2 - 1
    obs, rew, done, _ = env.step(a)
    bad_r = ~torch.isfinite(rew); bad_o = ~torch.isfinite(obs).all(1)
    print("step %2d non-finite reward %d obs %d  done %d" % (s, int(bad_r.sum()), int(bad_o.sum()), int(done.sum())))
    if bad_r.any():
        i = int(bad_r.nonzero()[0]); print("   env", i, "obs", obs[i, :11].tolist())

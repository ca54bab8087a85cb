"""Row / contact count distribution at the end of an env step for a given action scale (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from mujoco_jaco_amd.env import JacoBatchedEnv
B = 65536
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
genv = JacoBatchedEnv(num_envs=B, device=0, frame_skip=50, seed=1000, task="picking")
env = genv.sim
genv.reset()
gen = torch.Generator(device=env.device); gen.manual_seed(2000)
for i in range(8):
    genv.step((torch.rand(B, 7, device=env.device, generator=gen) * 2 - 1) * scale)
st = env.stats().cpu().numpy()
for name, col in (("ncon", 0), ("nefc", 1)):
    v = st[:, col]
    print(name, "pct [50,75,90,95,99,99.9,100]:", np.percentile(v, [50, 75, 90, 95, 99, 99.9, 100]).astype(int))
print("nefc > 64: %.3f  > 96: %.3f  > 128: %.3f  > 160: %.3f   ncon > 32: %.3f  > 48: %.3f" % tuple(
    [(st[:, 1] > t).mean() for t in (64, 96, 128, 160)] + [(st[:, 0] > t).mean() for t in (32, 48)]))

"""Why envs leave the light tier: overflow flags and row / contact counts of heavy-tier envs per env step (diagnostic)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
from mujoco_jaco_amd.env import JacoBatchedEnv
B = 65536
genv = JacoBatchedEnv(num_envs=B, device=0, frame_skip=50, seed=1000, task="picking")
env = genv.sim
genv.reset()
gen = torch.Generator(device=env.device); gen.manual_seed(2000)
actions = [torch.rand(B, 7, device=env.device, generator=gen) * 2 - 1 for _ in range(4)]
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    env.clear_flags()
    genv.step(actions[i % 4])
    fl = env.flags().cpu().numpy(); st = env.stats().cpu().numpy()
    hv = (fl & 32) != 0
    if i >= 2:
        print("step %d heavy %d  con-ovf %d efc-ovf %d cand-ovf %d | heavy envs: nefc pct [50,90,99,max] %s  ncon %s" % (
            i, hv.sum(), ((fl & 1) != 0).sum(), ((fl & 2) != 0).sum(), ((fl & 4) != 0).sum(),
            np.percentile(st[hv, 1], [50, 90, 99, 100]).astype(int) if hv.any() else None,
            np.percentile(st[hv, 0], [50, 90, 99, 100]).astype(int) if hv.any() else None))
    if i >= 2:
        a = actions[i % 4].cpu().numpy()
        for name, v in (("max|a_xyz|", np.abs(a[:, :3]).max(1)), ("|a_xyz|2", np.linalg.norm(a[:, :3], axis=1)), ("max|a_rot|", np.abs(a[:, 3:6]).max(1))):
            print("    %-10s heavy pct [10,50,90] %s   all %s" % (name, np.round(np.percentile(v[hv], [10, 50, 90]), 2), np.round(np.percentile(v, [10, 50, 90]), 2)))

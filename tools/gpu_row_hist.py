import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
from mujoco_jaco_amd.env import JacoBatchedEnv
B = 65536
env = JacoBatchedEnv(num_envs=B, seed=1000, task="picking")
env.reset()
gen = torch.Generator(device=env.device); gen.manual_seed(2000)
for step in range(12):
    a = torch.rand(B, 7, device=env.device, generator=gen) * 2 - 1
    env.sim.clear_flags()
    obs, r, d, _ = env.step(a)
    st = env.sim.stats().cpu().numpy(); fl = env.sim.flags().cpu().numpy()
    ne, nc = st[:, 1], st[:, 0]
    print("step %2d rows: mean %.1f  >64: %.2f%%  >128: %.2f%%  >256: %.2f%% | contacts >32: %.2f%% >64: %.2f%% | heavy-tier %.2f%%  done %.2f%%" % (
        step, ne.mean(), 100 * (ne > 64).mean(), 100 * (ne > 128).mean(), 100 * (ne >= 256).mean(), 100 * (nc > 32).mean(), 100 * (nc >= 64).mean(),
        100 * ((fl & 32) != 0).mean(), 100 * d.float().mean().item()), "osc-singular %.3f%%" % (100 * ((fl & 64) != 0).mean()))

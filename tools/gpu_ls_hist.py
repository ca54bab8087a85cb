"""Histogram of line-search / Newton iteration counts in the last substep of an env step (diagnostic)."""
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
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    genv.step(actions[i % 4])
st = env.stats().cpu().numpy()
nls = st[:, 3] >> 16; it = st[:, 2]
print("newton iterations:", np.bincount(it, minlength=8)[:10])
print("line-search iterations (sum over the Newton iterations of the last substep):")
h = np.bincount(np.minimum(nls, 60), minlength=61)
for lo, hi in ((0, 1), (1, 2), (2, 3), (3, 4), (4, 6), (6, 10), (10, 20), (20, 40), (40, 61)):
    print("  [%2d,%2d): %6d  (%.2f %%)" % (lo, hi, h[lo:hi].sum(), 100.0 * h[lo:hi].sum() / B))
print("mean nls %.2f; mean nls among envs with nls >= 10: %.1f" % (nls.mean(), nls[nls >= 10].mean() if (nls >= 10).any() else 0))

"""Why envs leave each capacity tier (diagnostic): per-step histogram of the informational bail-cause flags."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
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

"""Time library variants (different register budgets) on the bench workload; one process per variant."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from mujoco_jaco_amd import workload
from mujoco_jaco_amd.modelc import blob
from mujoco_jaco_amd.physics import BatchedMujoco
B = 65536
M = blob.load(os.path.join(%r, "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.jacomdl"))
env = BatchedMujoco(B); dev = env.device
env.set_state(torch.tensor(workload.reset_states(M["qpos0"], B, seed=1000), dtype=torch.float32, device=dev), None, None)
c = torch.tensor(workload.random_ctrl(B, seed=2000, scale=0.2), dtype=torch.float32, device=dev)
for _ in range(3): env.send_forces(c, nsub=1)
torch.cuda.synchronize(); t = time.time()
for _ in range(10): env.send_forces(c, nsub=1)
torch.cuda.synchronize(); dt = time.time() - t
print(os.environ.get("JACO_ENV_LIB"), "substeps/s %%.3g" %% (B * 10 / dt), "flags", int(env.flags().max()), "stats mean", env.stats().float().mean(0).cpu().numpy())
''' % (ROOT, ROOT)
for lib in sys.argv[1:]:
    env = dict(os.environ, JACO_ENV_LIB=lib)
    subprocess.run([sys.executable, "-c", code], env=env)

import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from mujoco_jaco_amd import workload
from mujoco_jaco_amd.modelc import blob
from mujoco_jaco_amd.physics import BatchedMujoco
from oracle_binding import Oracle
M = blob.load(os.path.join(ROOT, "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.jacomdl"))
B = 512
q = workload.reset_states(M["qpos0"], B, seed=21); c = workload.random_ctrl(B, seed=22, scale=0.2)
o = Oracle()
v = np.zeros((B, 21)); w = np.zeros((B, 21))
o.step_batch(q, v, w, np.ascontiguousarray(c), nsub=12, nthreads=16)
env = BatchedMujoco(B); dev = env.device
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
env.set_state(t(q), t(v), t(w)); env.send_forces(t(c), nsub=1)
gq, gv, gw = [x.cpu().numpy().astype(np.float64) for x in env.get_state()]
st = env.stats().cpu().numpy(); fl = env.flags().cpu().numpy()
f32 = lambda a: a.astype(np.float32).astype(np.float64)
err = np.zeros(B); info = []
for e in range(B):
    o.set("qpos", f32(q[e])); o.set("qvel", f32(v[e])); o.set("qacc_warmstart", f32(w[e])); o.step(f32(c[e]))
    err[e] = np.abs(gq[e] - o.get("qpos")).max()
    info.append((o.ncon, o.nefc, o.solver_iter, np.abs(gw[e] - o.get("qacc_warmstart")).max(), np.abs(o.get("qacc_warmstart")).max()))
order = np.argsort(-err)
print("median %.2e  p90 %.2e p99 %.2e" % (np.median(err), *np.percentile(err, [90, 99])))
for e in order[:14]:
    print("env %3d err %.2e gpu ncon/nefc/it/cand %s flags %d | oracle ncon/nefc/it %s qacc err %.3g of %.3g" % (e, err[e], st[e], fl[e], info[e][:3], info[e][3], info[e][4]))

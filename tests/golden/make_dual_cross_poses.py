"""Poses of jaco2_dual_torque.xml in which the two arms touch each other (test inputs, not reference outputs): a two-stage random search
on the fp64 oracle -- arm poses whose hand comes within 12 cm of the mid-plane x = 0 above z = 0.3, then pairs of them whose hands are
within 13 cm and whose contact count is moderate (13..36: objects on their holders = 12, plus arm-arm / arm-object contacts).
Writes tests/golden/dual_cross_poses.npz (qpos rows, fp32-representable).  Needs oracle/ built; no reference code involved."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_binding import Oracle

o = Oracle("jaco2_dual_torque")
names = {}
for line in open(os.path.join(ROOT, "mujoco_jaco_amd", "assets", "jaco2_dual_torque.names.txt")):
    k, v = line.strip().split(": ", 1); names[k] = v.split()
ee1, ee2 = names["body"].index("EE_1"), names["body"].index("EE_2")
rng = np.random.default_rng(4)
q0 = o.get("qpos").copy(); q0[18:21] = [-0.5, 0.6, 0.2001]; q0[25:28] = [0.5, 0.6, 0.2001]
samp = lambda: np.array([rng.uniform(0, 6.28), rng.uniform(3.5, 5.0), rng.uniform(0.6, 2.5), rng.uniform(0, 6.28), rng.uniform(0, 3), rng.uniform(0, 6.28)])
A, B = [], []
for t in range(3000):
    a = samp(); q = q0.copy(); q[0:6] = a; o.set("qpos", q); o.forward(); p = o.get("xpos").reshape(-1, 3)[ee1].copy()
    if abs(p[0]) < 0.12 and p[2] > 0.3: A.append((a, p))
    b = samp(); q = q0.copy(); q[9:15] = b; o.set("qpos", q); o.forward(); p = o.get("xpos").reshape(-1, 3)[ee2].copy()
    if abs(p[0]) < 0.12 and p[2] > 0.3: B.append((b, p))
found = []
for a, pa in A:
    for b, pb in B:
        if np.linalg.norm(pa - pb) < 0.13:
            q = q0.copy(); q[0:6] = a; q[9:15] = b
            q = q.astype(np.float32).astype(np.float64)
            o.set("qpos", q); o.forward()
            if 12 < o.ncon <= 36: found.append((o.ncon, q.copy()))
    if len(found) >= 12: break
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dual_cross_poses.npz"), qpos=np.array([f[1] for f in found]), ncon=np.array([f[0] for f in found]))
print("wrote", len(found), "poses, contact counts", [f[0] for f in found])

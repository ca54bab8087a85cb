"""Poses of jaco2_curtain_torque_sensor.xml in which the arm touches one of the static cylinders (the blocker's posts / rod,
xml:62-74, or the object holder's stem / disc, xml:278-281) -- test inputs, not reference outputs: a random search over arm poses on the
fp64 oracle, keeping ten with a shallow (< 8 mm) hull-cylinder contact, then four in which a finger part carrying a touch site presses on a
cylinder (non-zero touch readings after one step).  The object rests on the holder's disc in all of them.
Writes tests/golden/sensor_post_poses.npz (qpos rows, fp32-representable).  Needs oracle/ built; no reference code involved."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_binding import Oracle
from mujoco_jaco_amd.modelc import blob as blobmod

name = "jaco2_curtain_torque_sensor"
M = blobmod.loads(open(os.path.join(ROOT, "mujoco_jaco_amd", "assets", name + ".jacomdl"), "rb").read())
gtype = M["geom_type"]
o = Oracle(name)
rng = np.random.default_rng(11)
q0 = o.get("qpos").copy(); q0[12:15] = [0.0, 0.65, 0.4401]
found = []
for t in range(60000):
    q = q0.copy()
    q[0:6] = [rng.uniform(0, 6.28), rng.uniform(1.2, 5.0), rng.uniform(0.6, 5.6), rng.uniform(0, 6.28), rng.uniform(0, 6.28), rng.uniform(0, 6.28)]
    q[6:12] = rng.uniform([0, 0, 0, 0, 0, 0], [1.1, 0.4, 1.1, 0.4, 1.1, 0.4])
    q = q.astype(np.float32).astype(np.float64)
    o.set("qpos", q); o.forward()
    if not (1 < o.ncon <= 6): continue
    C = o.get("contact").reshape(-1, 11)
    cyl = [(c[0], int(c[7]), int(c[8])) for c in C if 5 in (gtype[int(c[7])], gtype[int(c[8])]) and 7 in (gtype[int(c[7])], gtype[int(c[8])])]
    if cyl and all(-0.008 < c[0] for c in C):
        found.append((o.ncon, q.copy(), cyl[0][1:]))
        if len(found) >= 10: break
# ... and four more in which a finger part that carries a touch site presses on a cylinder: one step gives a non-zero touch reading
rng = np.random.default_rng(12)
touch, ctrl = [], np.zeros(9)
for t2 in range(200000):
    q = q0.copy()
    q[0:6] = [rng.uniform(0, 6.28), rng.uniform(1.2, 5.0), rng.uniform(0.6, 5.6), rng.uniform(0, 6.28), rng.uniform(0, 6.28), rng.uniform(0, 6.28)]
    q[6:12:2] = rng.uniform(0.2, 1.1, 3); q[7:12:2] = rng.uniform(0, 0.4, 3)
    q = q.astype(np.float32).astype(np.float64)
    o.set("qpos", q); o.forward()
    if not (1 < o.ncon <= 8): continue
    C = o.get("contact").reshape(-1, 11)
    if not all(-0.006 < c[0] for c in C) or not any(5 in (gtype[int(c[7])], gtype[int(c[8])]) for c in C): continue
    o.set("qvel", np.zeros(18)); o.set("qacc_warmstart", np.zeros(18))
    ctrl[6:9] = q[6:12:2]
    o.step(ctrl, n=1)
    if o.get("sensordata").max() > 0:
        touch.append((o.ncon, q.copy(), float(o.get("sensordata").max())))
        if len(touch) >= 4: break
found = [(f[0], f[1]) for f in found] + [(f[0], f[1]) for f in touch]
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sensor_post_poses.npz"), qpos=np.array([f[1] for f in found]), ncon=np.array([f[0] for f in found]),
                    ntouch=np.array([len(touch)]))
print("wrote", len(found), "poses (the last", len(touch), "with touch readings", ["%.2f" % f[2] for f in touch], "); contacts", [f[0] for f in found])

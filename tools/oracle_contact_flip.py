"""CPU only (fp64 oracle): how often does a contact of the smoke workload's env `env` jump when the state before substep `k` is perturbed by
U(-eps, eps) in the arm / finger angles?  (profiles/r04_smoke_knife_edge.txt: env 62, k 6, eps 3e-7 -> 87 of 400 draws move one hull-box contact by 2 mm.)

  python tools/oracle_contact_flip.py [env] [k] [eps] [draws]
"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mujoco_jaco_amd import workload
from mujoco_jaco_amd.modelc import blob
from oracle_binding import Oracle

i = int(sys.argv[1]) if len(sys.argv) > 1 else 62
k = int(sys.argv[2]) if len(sys.argv) > 2 else 6
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 3e-7
draws = int(sys.argv[4]) if len(sys.argv) > 4 else 400
M = blob.load(os.path.join(ROOT, "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.jacomdl"))
q = workload.reset_states(M["qpos0"], 64, seed=7, f32_draws=True)
c = workload.random_ctrl(64, seed=8, scale=0.2).astype(np.float32).astype(np.float64)
o = Oracle()
o.set("qpos", q[i]); o.set("qvel", np.zeros(21)); o.set("qacc_warmstart", np.zeros(21))
o.step(c[i], n=k)
qs, vs, ws = o.get("qpos").copy(), o.get("qvel").copy(), o.get("qacc_warmstart").copy()


def contacts(qp):
    o.set("qpos", qp); o.set("qvel", vs); o.set("qacc_warmstart", ws); o.set("ctrl", c[i]); o.forward()
    return o.get("contact").reshape(-1, 11).copy()


C0 = contacts(qs)
rng = np.random.default_rng(1)
flips, worst, shown = 0, 0.0, 0
for t in range(draws):
    qp = qs.copy(); qp[:9] += rng.uniform(-1, 1, 9) * eps
    C = contacts(qp)
    if len(C) != len(C0):
        flips += 1; continue
    dp, dn = np.abs(C[:, 1:4] - C0[:, 1:4]).max(1), np.abs(C[:, 4:7] - C0[:, 4:7]).max(1)
    worst = max(worst, dp.max())
    if dp.max() > 1e-4:
        flips += 1
        j = int(dp.argmax())
        if shown < 5:
            shown += 1
            print("draw %d: contact %d (geoms %d-%d, condim %d) pos diff %.2e normal diff %.2e dist %.6f vs %.6f" % (t, j, int(C[j, 7]), int(C[j, 8]), int(C[j, 9]), dp[j], dn[j], C[j, 0], C0[j, 0]))
print("env %d before substep %d: %d contacts; %d of %d draws (eps %.0e) move a contact by more than 1e-4; worst %.2e" % (i, k + 1, len(C0), flips, draws, eps, worst))

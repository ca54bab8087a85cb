"""Diagnostic for the smoke workload's most violent env (a reset with the hand inside the pedestal): the oracle is run `k` substeps, the GPU
is handed that state and both take the next substep; contact lists (dist / pos / normal), qacc and the state after the step side by side.

  python tools/gpu_smoke_contacts.py [env] [k]
"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from mujoco_jaco_amd import workload
from mujoco_jaco_amd.modelc import blob
from mujoco_jaco_amd.physics import BatchedMujoco
from oracle_binding import Oracle

env_i = int(sys.argv[1]) if len(sys.argv) > 1 else 62
k0 = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B = 64
M = blob.load(os.path.join(ROOT, "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.jacomdl"))
q = workload.reset_states(M["qpos0"], B, seed=7, f32_draws=True)
c = workload.random_ctrl(B, seed=8, scale=0.2).astype(np.float32).astype(np.float64)
f32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
o = Oracle()
o.set("qpos", q[env_i]); o.set("qvel", np.zeros(21)); o.set("qacc_warmstart", np.zeros(21))
o.step(c[env_i], n=k0)
for k in range(k0, k0 + 3):
    qs, vs, ws = f32(o.get("qpos")), f32(o.get("qvel")), f32(o.get("qacc_warmstart"))
    o.set("qpos", qs); o.set("qvel", vs); o.set("qacc_warmstart", ws)   # both sides start the substep from the same fp32 numbers
    env = BatchedMujoco(B, device=0)
    dev = env.device
    t = lambda a: torch.tensor(np.tile(a, (B, 1)), dtype=torch.float32, device=dev)
    env.set_state(t(qs), t(vs), t(ws))
    D = env.send_forces_debug(t(c[env_i]), 0, nsub=1)
    gq, gv, _ = [x.cpu().numpy().astype(np.float64)[0] for x in env.get_state()]
    o.step(c[env_i], n=1)
    off = 33 + 99 + 441
    qacc_g = D[off + 3 * 24: off + 4 * 24][:21]
    off += 5 * 24
    nc, ne = int(D[off]), int(D[off + 1])
    C = D[off + 4:off + 4 + 8 * min(nc, 64)].reshape(-1, 8)
    oc = o.get("contact").reshape(-1, 11)
    print("== substep %d: gpu contacts/rows %d/%d, oracle %d/%d; state after the step: qpos diff %.2e qvel diff %.2e (dof %d)" % (
        k + 1, nc, ne, o.ncon, o.nefc, np.abs(gq - o.get("qpos")).max(), np.abs(gv - o.get("qvel")).max(), int(np.abs(gv - o.get("qvel")).argmax())))
    print("   qacc: oracle max |.| %.3e, diff max %.3e" % (np.abs(o.get("qacc")).max(), np.abs(qacc_g - o.get("qacc")).max()))
    if nc == o.ncon:
        dd, dp, dn = np.abs(C[:, 0] - oc[:nc, 0]), np.abs(C[:, 1:4] - oc[:nc, 1:4]).max(1), np.abs(C[:, 4:7] - oc[:nc, 4:7]).max(1)
        w = np.argsort(-dn)[:4]
        for i in w:
            print("   contact %2d geoms %d-%d dim %d: dist gpu %.6f oracle %.6f | pos diff %.2e | normal diff %.2e  (gpu %s oracle %s)" % (
                i, int(oc[i, 7]), int(oc[i, 8]), int(oc[i, 9]), C[i, 0], oc[i, 0], dp[i], dn[i], np.round(C[i, 4:7], 5), np.round(oc[i, 4:7], 5)))
    env.close()

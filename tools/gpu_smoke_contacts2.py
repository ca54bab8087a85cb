"""As gpu_smoke_contacts.py, but the other way round: the GPU free-runs the smoke workload in 1-substep launches; before every substep from
`k0` on the oracle is handed the GPU's own (qpos, qvel, qacc_warmstart) of env `env`, then both take the substep (GPU: the continuing env, with
its hints / tier routing as they are) and are compared: contacts, qacc, state after the step.

  python tools/gpu_smoke_contacts2.py [env] [k0] [nsteps] [name=value ...]
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

e_i = int(sys.argv[1]) if len(sys.argv) > 1 else 62
k0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 4
B = 64
M = blob.load(os.path.join(ROOT, "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.jacomdl"))
q = workload.reset_states(M["qpos0"], B, seed=7, f32_draws=True)
c = workload.random_ctrl(B, seed=8, scale=0.2).astype(np.float32).astype(np.float64)
env = BatchedMujoco(B, device=0)
for kv in sys.argv[4:]:
    k, v = kv.split("="); env.set_option(k, float(v))
dev = env.device
env.set_state(torch.tensor(q, dtype=torch.float32, device=dev), None, None)
ct = torch.tensor(c, dtype=torch.float32, device=dev)
o = Oracle()
for k in range(k0):
    env.send_forces(ct, nsub=1)
for k in range(k0, k0 + ns):
    gq, gv, gw = [x.cpu().numpy().astype(np.float64)[e_i] for x in env.get_state()]
    o.set("qpos", gq); o.set("qvel", gv); o.set("qacc_warmstart", gw)
    fl0 = int(env.flags()[e_i])
    D = env.send_forces_debug(ct, e_i, nsub=1)
    o.step(c[e_i], n=1)
    gq1, gv1, _ = [x.cpu().numpy().astype(np.float64)[e_i] for x in env.get_state()]
    off = 33 + 99 + 441
    qacc_g = D[off + 3 * 24: off + 4 * 24][:21]
    off += 5 * 24
    nc, ne, it = int(D[off]), int(D[off + 1]), int(D[off + 2])
    C = D[off + 4:off + 4 + 8 * min(nc, 64)].reshape(-1, 8)
    oc = o.get("contact").reshape(-1, 11)
    dv = gv1 - o.get("qvel")
    print("== substep %d: gpu contacts/rows/iters %d/%d/%d, oracle %d/%d/%d; after the step qpos diff %.2e qvel diff %.2e (dof %d) flags before 0x%x after 0x%x" % (
        k + 1, nc, ne, it, o.ncon, o.nefc, o.solver_iter, np.abs(gq1 - o.get("qpos")).max(), np.abs(dv).max(), int(np.abs(dv).argmax()), fl0, int(env.flags()[e_i])))
    print("   qacc: oracle max |.| %.3e, diff max %.3e at dof %d;  qacc diff per dof %s" % (np.abs(o.get("qacc")).max(), np.abs(qacc_g - o.get("qacc")).max(),
          int(np.abs(qacc_g - o.get("qacc")).argmax()), np.array2string(qacc_g - o.get("qacc"), precision=2)))
    if nc == o.ncon:
        dd, dp, dn = np.abs(C[:, 0] - oc[:nc, 0]), np.abs(C[:, 1:4] - oc[:nc, 1:4]).max(1), np.abs(C[:, 4:7] - oc[:nc, 4:7]).max(1)
        print("   contacts: dist diff max %.2e, pos diff max %.2e, normal diff max %.2e" % (dd.max(), dp.max(), dn.max()))

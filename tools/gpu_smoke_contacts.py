"""GPU box diagnostic for the smoke workload's violent envs (resets with the hand inside the pedestal): kernel and oracle take single substeps
from a common state and are compared -- contact lists (dist / pos / normal), qacc and the state after the step side by side.  Two modes:

  python tools/gpu_smoke_contacts.py oracle-leads [env] [k]                       the oracle runs k substeps, the GPU is handed that state, 3 steps
  python tools/gpu_smoke_contacts.py gpu-leads [env] [k0] [nsteps] [name=value ...]   the GPU free-runs in 1-substep launches (its own hints / tier
                                                                                  routing); before every substep from k0 on the oracle gets the GPU's state
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

mode = sys.argv[1] if len(sys.argv) > 1 else "oracle-leads"
assert mode in ("oracle-leads", "gpu-leads"), mode
sys.argv = sys.argv[:1] + sys.argv[2:]


def oracle_leads():
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


def gpu_leads():
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


(oracle_leads if mode == "oracle-leads" else gpu_leads)()

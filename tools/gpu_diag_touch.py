import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from mujoco_jaco_amd.modelc import blob, rot
from mujoco_jaco_amd.physics import BatchedMujoco
from oracle_binding import Oracle
M = blob.load(os.path.join(ROOT, "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.jacomdl"))
names = {}
for line in open(os.path.join(ROOT, "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.names.txt")):
    k, v = line.strip().split(": ", 1); names[k] = v.split()
o = Oracle()
q = M["qpos0"].copy()
q[:6] = [1.3, 3.85, 1.05, 2.05, 1.5, -1.15]; q[6:9] = 0.6; q[16:18] = [.4, .3]
o.set("qpos", q); o.forward()
b = names["body"].index("EE_obj")
xp = o.get("xpos").reshape(-1, 3)[b]; xq = o.get("xquat").reshape(-1, 4)[b]
q[9:12] = xp + rot.quat_to_mat(xq) @ np.array([-0.04, 0, 0]); q[12:16] = xq
B = 4
env = BatchedMujoco(B); dev = env.device
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
C = np.tile(np.array([0, 0, 0, 0, 0, 0, .8, .8, .8]), (B, 1))
o.reset(); o.set("qpos", q.astype(np.float32).astype(np.float64))
f32 = lambda a: a.astype(np.float32).astype(np.float64)
for i in range(6):
    st = [f32(o.get(n)) for n in ("qpos", "qvel", "qacc_warmstart")]
    for n, x in zip(("qpos", "qvel", "qacc_warmstart"), st): o.set(n, x)
    env.set_state(*[t(np.tile(x, (B, 1))) for x in st])
    D = env.send_forces_debug(t(C), 0, nsub=1)
    o.step(C[0])
    if i == 3:
        off = 33 + 99 + 441 + 5 * 24
        nc = int(D[off]); Cg = D[off + 4:off + 4 + 8 * nc].reshape(nc, 8); oc = o.get("contact").reshape(-1, 11)
        gnm = names["geom"]
        for k in range(nc):
            if "thumb_proximal_plane" in (gnm[int(oc[k, 7])], gnm[int(oc[k, 8])]):
                print("   contact", k, gnm[int(oc[k, 7])], gnm[int(oc[k, 8])], "gpu dist %.6f pos %s n %s" % (Cg[k, 0], Cg[k, 1:4].round(5), Cg[k, 4:7].round(4)),
                      "| oracle dist %.6f pos %s n %s" % (oc[k, 0], oc[k, 1:4].round(5), oc[k, 4:7].round(4)))
        E = D[off + 4 + 8 * 64: off + 4 + 8 * 64 + 4 * o.nefc].reshape(-1, 4)
        print("   force err", np.abs(E[:, 3] - o.get("efc_force")).max())
        base = off + 4 + 8 * 64 + 4 * 256 + 3 * 64
        print("   gpu c_fn", D[base:base + nc].round(2)); print("   gpu sens (dump)", D[base + 64:base + 84].round(2))
        f = o.get("efc_force"); print("   oracle fn", np.array([f[int(r[10]):int(r[10]) + 2 * (int(r[9]) - 1)].sum() for r in oc]).round(2))
    s = env.sensordata().cpu().numpy()[0]; so = o.get("sensordata")
    gq = env.get_state()[0].cpu().numpy()[0]
    print(i, "qpos err %.2e" % np.abs(gq - o.get("qpos")).max(), "stats", env.stats().cpu().numpy()[0], "oracle", o.ncon, o.nefc)
    print("   gpu   ", s.round(2)); print("   oracle", so.round(2))

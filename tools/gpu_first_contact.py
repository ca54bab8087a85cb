"""First GPU run: stage-by-stage parity of the HIP kernel against the oracle, then a throughput probe."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from oracle_binding import Oracle
from mujoco_jaco_amd.physics import BatchedMujoco
from mujoco_jaco_amd.modelc import blob

contact = "--contact" in sys.argv
B = 64
env = BatchedMujoco(B)
if not contact:
    env.set_option("disable_contact", 1)
o = Oracle()
if not contact:
    o.option("disable_contact", 1)
M = blob.load(os.path.join(os.path.dirname(__file__), "..", "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.jacomdl"))
rng = np.random.default_rng(0)
q = np.tile(M["qpos0"], (B, 1)); v = np.zeros((B, 21)); ctrl = np.zeros((B, 9))
for e in range(B):
    q[e, :6] = [rng.uniform(.7, 2.5), rng.uniform(3.8, 4), rng.uniform(1, 1.7), rng.uniform(1.8, 2.5), rng.uniform(1, 2.5), rng.uniform(.8, 2.3)]
    q[e, 6:9] = rng.uniform(0.0, 1.5, 3)
    q[e, 9:12] = [rng.uniform(-.1, .1), .65 + rng.uniform(-.08, .02), .1898 if contact else 0.5]
    q[e, 16:18] = [.4 + rng.uniform(-.05, .05), .3 + rng.uniform(-.05, .05)]
    v[e, :9] = rng.normal(size=9) * 0.2
    ctrl[e] = np.concatenate([rng.uniform(-1, 1, 6) * [30, 30, 30, 15, 15, 15] * 0.3, rng.uniform(0, 1.5, 3)])
dev = env.device
tq = torch.tensor(q, dtype=torch.float32, device=dev); tv = torch.tensor(v, dtype=torch.float32, device=dev)
tc = torch.tensor(ctrl, dtype=torch.float32, device=dev)
env.set_state(tq, tv, torch.zeros_like(tv))
roots = [b for b in range(1, int(M["nbody"][0])) if M["body_weldid"][b] == b]
D = env.send_forces_debug(tc, 3, nsub=1)
o.set("qpos", q[3]); o.set("qvel", v[3]); o.set("ctrl", ctrl[3]); o.set("qacc_warmstart", np.zeros(21)); o.forward()
print("xpos err", np.abs(D[0:33].reshape(11, 3) - o.get("xpos").reshape(-1, 3)[roots]).max())
off = 33 + 99
Mo = o.get("qM").reshape(21, 21); print("M rel err", np.abs(D[off:off + 441].reshape(21, 21) - Mo).max() / np.abs(Mo).max()); off += 441
for nm in ["qfrc_bias", "qfrc_smooth", "qacc_smooth", "qacc", "qfrc_constraint"]:
    a = o.get(nm); b = D[off:off + 21]; print(nm, "abs err", np.abs(a - b).max(), "max", np.abs(a).max()); off += 24
print("stats gpu", D[off:off + 4], "oracle ncon/nefc/iter", o.ncon, o.nefc, o.solver_iter)
# multi-step drift, all envs
nstep = 200
qo, vo, wo = q.copy(), v.copy(), np.zeros((B, 21))
o.step_batch(qo, vo, wo, np.ascontiguousarray(ctrl), nsub=nstep + 1, nthreads=8)
env.send_forces(tc, nsub=nstep)
gq, gv, gw = [t.cpu().numpy() for t in env.get_state()]
print("after", nstep + 1, "steps: max qpos err", np.abs(gq - qo).max(), "arm+finger", np.abs(gq - qo)[:, :9].max(), "qvel err", np.abs(gv - vo).max())
print("flags", env.flags().cpu().numpy().max(), "stats max", env.stats().cpu().numpy().max(0))
# throughput probe
for Bn in (4096, 65536):
    e2 = BatchedMujoco(Bn)
    if not contact:
        e2.set_option("disable_contact", 1)
    c2 = torch.zeros(Bn, 9, device=dev); c2[:, 6:] = 0.6
    if contact:
        q2 = torch.tensor(np.tile(q, (Bn // B, 1)), dtype=torch.float32, device=dev)
        e2.set_state(q2, None, None)
    e2.send_forces(c2, nsub=2); torch.cuda.synchronize()
    t = time.time(); e2.send_forces(c2, nsub=20); torch.cuda.synchronize(); dt = time.time() - t
    print("B", Bn, "substeps/s", Bn * 20 / dt, "us per substep-wave (1 wave)", dt / 20 / (Bn / (256 * 6)) * 1e6, "stats max", e2.stats().cpu().numpy().max(0), "flags", int(e2.flags().max()))
    e2.close()

"""The smoke workload (64 envs x 20 substeps, __graft_entry__.smoke) substep by substep against the fp64 oracle, for the env with the largest
error: per-substep qpos error and contact / row counts of both sides, free-running (no re-synchronisation), under execution options.

  python tools/gpu_smoke_trace.py [name=value ...]      e.g. pair_list=0 sep_cache=0
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

B, nsub = 64, 20
M = blob.load(os.path.join(ROOT, "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.jacomdl"))
q = workload.reset_states(M["qpos0"], B, seed=7, f32_draws=True)
c = workload.random_ctrl(B, seed=8, scale=0.2).astype(np.float32).astype(np.float64)
o = Oracle()
for chunk in (20, 1):
    env = BatchedMujoco(B, device=0)
    for kv in sys.argv[1:]:
        k, v = kv.split("="); env.set_option(k, float(v))
    dev = env.device
    env.set_state(torch.tensor(q, dtype=torch.float32, device=dev), None, None)
    qo, vo, wo = q.copy(), np.zeros((B, 21)), np.zeros((B, 21))
    st = np.zeros((B, 4), np.int32)
    ct = torch.tensor(c, dtype=torch.float32, device=dev)
    print("== %d launches of %d substeps, options %s" % (nsub // chunk, chunk, sys.argv[1:]))
    for k in range(nsub // chunk):
        env.send_forces(ct, nsub=chunk)
        o.step_batch(qo, vo, wo, np.ascontiguousarray(c), nsub=chunk, nthreads=8, stats=st)
        gq = env.get_state()[0].cpu().numpy().astype(np.float64)
        gs = env.stats().cpu().numpy()
        err = np.abs(gq - qo).max(axis=1)
        w = int(err.argmax())
        print("after %2d substeps: max err %.2e (env %d: gpu contacts/rows %d/%d, oracle %d/%d, flags 0x%x) median %.2e | env 62: %.2e gpu %d/%d oracle %d/%d" % (
            (k + 1) * chunk, err.max(), w, gs[w, 0], gs[w, 1], st[w, 0], st[w, 1], int(env.flags()[w]), np.median(err), err[62], gs[62, 0], gs[62, 1], st[62, 0], st[62, 1]))
    env.close()

"""Drift of the HIP path against the fp64 oracle over 1 000 physics steps (the BASELINE metric's second half, with the
oracle standing in for the MuJoCo that is not available): ctrl level (constant random torques, 1 000 substeps) for 256
envs of the picking reset distribution, and the arm-only model (config 2, no contacts) for 256 envs.  Per-env results go to
gpurun_out/gpu_drift.npz for tools/drift_attribution.py."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from mujoco_jaco_amd import workload, _lib
from mujoco_jaco_amd.modelc import blob
from mujoco_jaco_amd.physics import BatchedMujoco
from oracle_binding import Oracle

def run(model, B, nsub, contact, scale, save=None, comp=1):
    M = blob.load(_lib.model_path(model))
    q = workload.reset_states(M["qpos0"], B, seed=41, f32_draws=True)
    nu = int(M["nu"][0]); nv = int(M["nv"][0])
    c = workload.random_ctrl(B, seed=42, scale=scale)[:, :nu]
    env = BatchedMujoco(B, robot_file=model)
    env.set_option("compensated", comp)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=env.device)
    env.set_state(t(q), None, None)
    o = Oracle(model)
    if not contact:
        env.set_option("disable_contact", 1); o.option("disable_contact", 1)
    out = {}
    qo, vo, wo = np.ascontiguousarray(q), np.zeros((B, nv)), np.zeros((B, nv))
    cc = np.ascontiguousarray(c.astype(np.float32).astype(np.float64))
    done = 0
    for mark in (100, 300, 1000):
        env.send_forces(t(c), nsub=mark - done)
        o.step_batch(qo, vo, wo, cc, nsub=mark - done, nthreads=len(os.sched_getaffinity(0)))   # in place
        done = mark
        err = np.abs(env.get_state()[0].cpu().numpy().astype(np.float64) - qo).max(1)
        out[mark] = err
        print("%-24s comp %d %4d substeps: max-abs qpos error median %.2e  p90 %.2e  p99 %.2e  max %.2e  (<= 1e-4: %.1f %%)" % (
            model, comp, mark, np.median(err), *np.percentile(err, [90, 99]), err.max(), 100 * np.mean(err <= 1e-4)))
    if save:
        np.savez(save, flags=env.flags().cpu().numpy(), **{"err_%d" % k: v for k, v in out.items()})
    return out

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for comp in (0, 1):   # 0: plain fp32 state (rounds 1-2); 1: qpos / qvel carried as hi + lo floats (the default)
    run("jaco2_reaching_torque", B, 1000, False, 0.2, comp=comp)
    run("jaco2_curtain_torque", B, 1000, True, 0.2, save=os.path.join(ROOT, "gpurun_out", "gpu_drift%s.npz" % ("" if comp else "_plain")), comp=comp)

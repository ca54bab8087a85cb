"""Control experiment for the 1 000-step drift metric (CPU only; VERDICT r01 "next" item 1a).

Question: when the fp32 HIP path parts from the fp64 oracle after a few hundred steps, is that the HIP path's arithmetic
or the dynamics (unactuated impacts, stick/slip, contact on/off switching) amplifying ANY last-bit difference?
Controls, all run by the SAME fp64 code on the SAME envs / ctrl as tools/gpu_drift.py:
  A  fp64 oracle                                   (the reference trajectory)
  B  fp64 oracle whose state (qpos, qvel, qacc_warmstart) is rounded to fp32 after every step
  C  fp64 oracle started from a state 1 ulp(fp32) away in the six arm joints
  D  fp64 oracle handed the pedestal height as the fp32 number 0.09f instead of 0.09 (3.6e-9 higher): the pedestal's bottom
     face sits exactly on the floor plane by construction (0.09 + 0.07 - 0.16), so this decides whether its 4 floor contacts
     exist in the very first step -- the knife edge every reset of the reference starts on
  E  fp64 oracle whose STATE stays fp64 while every forward pass (kinematics .. qacc) is evaluated at the fp32 rounding of it: the
     ceiling for an fp32 engine that carries qpos / qvel compensated (hi + lo floats) -- rounding perturbs each evaluation but
     never accumulates in the state (VERDICT r02 "next" item 1a)
B, C, D and E contain no fp32 arithmetic at all: every force, contact and solve is fp64.  Whatever divergence they show is the
floor for any engine that carries fp32 state (B) or is handed inputs that differ in the last fp32 bit (C).
Writes per-env errors at 100 / 300 / 1000 steps + the oracle's max contact / row counts to an .npz for the attribution.
"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mujoco_jaco_amd import workload, _lib
from mujoco_jaco_amd.modelc import blob
from oracle_binding import Oracle

MARKS = (100, 300, 1000)


def run(model, B, contact, scale, variant, seed=41):
    M = blob.load(_lib.model_path(model))
    nu, nv = int(M["nu"][0]), int(M["nv"][0])
    q = workload.reset_states(M["qpos0"], B, seed=seed, f32_draws=True)
    c = np.ascontiguousarray(workload.random_ctrl(B, seed=seed + 1, scale=scale)[:, :nu].astype(np.float32).astype(np.float64))
    o = Oracle(model)
    if not contact:
        o.option("disable_contact", 1)
    if variant == "B":
        o.option("round_state", 1)
    if variant == "E":
        o.option("round_state", 2)
    if variant == "F":
        o.option("round_state", 3)
    if variant == "C":
        q[:, :6] = np.nextafter(q[:, :6].astype(np.float32), np.float32(np.inf)).astype(np.float64)
    if variant == "D" and q.shape[1] >= 23:
        q[:, 18] = np.float64(np.float32(q[:, 18]))   # pedestal height 0.09 -> 0.09f: its bottom face leaves the floor plane by 3.6 nm
    q = np.ascontiguousarray(q)
    v, w = np.zeros((B, nv)), np.zeros((B, nv))
    out, done = {}, 0
    stats = np.zeros((B, 4), np.int32)
    smax = np.zeros((B, 4), np.int32)
    for mark in MARKS:
        o.step_batch(q, v, w, c, nsub=mark - done, nthreads=len(os.sched_getaffinity(0)), stats=stats)
        smax = np.maximum(smax, stats)
        done = mark
        out[mark] = q.copy()
    return out, smax


def summarize(tag, err):
    for mark in MARKS:
        e = err[mark]
        print("%-44s %4d steps: median %.2e  p90 %.2e  p99 %.2e  max %.2e  (<= 1e-4: %5.1f %%)" % (
            tag, mark, np.median(e), *np.percentile(e, [90, 99]), e.max(), 100 * np.mean(e <= 1e-4)), flush=True)


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    save = {}
    for model, contact in (("jaco2_reaching_torque", False), ("jaco2_curtain_torque", True)):
        ref, smax = run(model, B, contact, 0.2, "A")
        save[model + "_maxcon"] = smax[:, 0]; save[model + "_maxefc"] = smax[:, 1]
        for variant, what in (("B", "fp64 oracle, fp32-rounded state"), ("C", "fp64 oracle, +1 ulp(fp32) initial arm angles"),
                              ("D", "fp64 oracle, pedestal height 0.09f instead of 0.09"),
                              ("E", "fp64 state, forward pass sees its fp32 rounding"),
                              ("F", "as E, pedestal height seen exactly (no knife edge)")):
            if variant in "DF" and not contact:
                continue
            got, _ = run(model, B, contact, 0.2, variant)
            err = {k: np.abs(got[k] - ref[k]).max(1) for k in MARKS}
            summarize("%s | %s" % (model, what), err)
            for k in MARKS:
                save["%s_%s_%d" % (model, variant, k)] = err[k]
        for k in MARKS:
            save["%s_A_qpos_%d" % (model, k)] = ref[k]
    np.savez(os.path.join(ROOT, "gpurun_out", "drift_control.npz"), **save)

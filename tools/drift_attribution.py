"""Per-env attribution of the 1 000-step drift (CPU only; VERDICT r01 "next" item 1b).

Runs, for every env of tools/gpu_drift.py's workload, THREE trajectories in lockstep:
  orc   the fp64 oracle, free running
  emu   the fp32 kernel source (host build under the wavefront emulator, tests/emu), free running
  syn   the fp32 kernel restarted from the oracle's state at every step (single-step error: no accumulation)
and records, per step: max-abs qpos error of emu and of syn, both sides' contact / row counts, kernel flags.
From that, per env:
  onset      first step where the free-running error exceeds 1e-5
  cause      what the single-step comparison shows in the 3 steps up to the onset:
               overflow   a capacity flag was raised (rows / contacts / candidates dropped)
               contactset the two sides disagree on the number of contacts or rows (a contact switching on/off one step apart)
               narrow     same contact set, but single-step error > 1e-5 (contact geometry / solver difference)
               smooth     single-step error stays < 1e-5: the growth is the dynamics amplifying the accumulated 1e-7s
This is a diagnostic: it needs the oracle and the emulator (test infrastructure), never the product library.
"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiprocessing as mp
import numpy as np

MODEL = "jaco2_curtain_torque"
NSTEP = 1000


def one_env(args):
    k, q0, c = args
    from emu_binding import EmuEnv
    from oracle_binding import Oracle
    o = Oracle(MODEL); e = EmuEnv(MODEL); s = EmuEnv(MODEL)
    o.reset(); o.set("qpos", q0); o.set("ctrl", c)
    e.qpos[0] = q0
    rec = np.zeros((NSTEP, 8))
    for t in range(NSTEP):
        st = [o.get(n) for n in ("qpos", "qvel", "qacc_warmstart")]
        s.qpos[0], s.qvel[0], s.qacc_ws[0] = st
        s.flags[0] = 0
        s.step(c)
        e.step(c)
        o.step()
        qo = o.get("qpos")
        rec[t] = (np.abs(e.qpos[0] - qo).max(), np.abs(s.qpos[0] - qo).max(), o.ncon, o.nefc, s.stats[0, 0], s.stats[0, 1], s.flags[0], e.flags[0])
        if not np.isfinite(rec[t, 0]) or rec[t, 0] > 10:
            rec[t:] = rec[t]
            break
    return k, rec


def classify(rec):
    """(onset step, cause) of an env's free-running divergence; onset = first step with error > 1e-5."""
    err, syn = rec[:, 0], rec[:, 1]
    hit = np.nonzero(err > 1e-5)[0]
    if len(hit) == 0:
        return -1, "clean"
    t = int(hit[0])
    # the error usually crosses 1e-5 a little after the event that seeded it: look back over the growth phase
    w = slice(max(0, t - 40), t + 1)
    if (rec[w, 6].astype(int) & 7).any() or (rec[w, 7].astype(int) & 7).any():
        return t, "overflow"
    if (rec[w, 2] != rec[w, 4]).any() or (rec[w, 3] != rec[w, 5]).any():
        return t, "contactset"
    if (syn[w] > 1e-5).any():
        return t, "narrow"
    return t, "smooth"


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
    from mujoco_jaco_amd import workload, _lib
    from mujoco_jaco_amd.modelc import blob
    M = blob.load(_lib.model_path(MODEL))
    q = workload.reset_states(M["qpos0"], B, seed=41, f32_draws=True)
    c = workload.random_ctrl(B, seed=42, scale=0.2).astype(np.float32).astype(np.float64)
    if "--report" in sys.argv:   # re-print the tables from the saved per-step records
        rec = np.load(os.path.join(ROOT, "gpurun_out", "drift_attribution.npy"))
        B = rec.shape[0]
    else:
        with mp.Pool(os.cpu_count()) as pool:
            res = pool.map(one_env, [(k, q[k], c[k]) for k in range(B)], chunksize=1)
        rec = np.zeros((B, NSTEP, 8))
        for k, r in res:
            rec[k] = r
        np.save(os.path.join(ROOT, "gpurun_out", "drift_attribution.npy"), rec)
    final = rec[:, -1, 0]
    print("free-running fp32 kernel (emulated) vs fp64 oracle, %d envs x %d steps: <= 1e-4: %.1f %%, median %.2e" % (B, NSTEP, 100 * np.mean(final <= 1e-4), np.median(final)))
    for mark in (100, 300, 1000):
        e = rec[:, mark - 1, 0]
        print("  %4d steps: median %.2e p90 %.2e p99 %.2e max %.2e (<= 1e-4: %.1f %%)" % (mark, np.median(e), *np.percentile(e, [90, 99]), e.max(), 100 * np.mean(e <= 1e-4)))
    syn = rec[:, :, 1]
    print("single-step error (restart from the oracle state every step): median of per-env max %.2e, p99 %.2e, max %.2e; steps > 1e-5: %d of %d" % (
        np.median(syn.max(1)), np.percentile(syn.max(1), 99), syn.max(), int((syn > 1e-5).sum()), syn.size))
    ctl = None
    try:
        ctl = np.load(os.path.join(ROOT, "gpurun_out", "drift_control.npz"))
    except Exception:
        pass
    gpu = None
    try:
        gpu = np.load(os.path.join(ROOT, "gpurun_out", "gpu_drift.npz"))
    except Exception:
        pass
    mism = (rec[:, :, 2] != rec[:, :, 4]) | (rec[:, :, 3] != rec[:, :, 5])
    big = syn > 1e-5
    print("single-step errors > 1e-5: %d steps; with a contact / row count mismatch in that step: %d; with a capacity flag: %d; neither: %d (max %.2e)" % (
        int(big.sum()), int((big & mism).sum()), int((big & ((rec[:, :, 6].astype(int) & 7) != 0)).sum()),
        int((big & ~mism & ((rec[:, :, 6].astype(int) & 7) == 0)).sum()), syn[big & ~mism].max() if (big & ~mism).any() else 0.0))
    clean = ~mism & ((rec[:, :, 6].astype(int) & 7) == 0)
    print("single-step error over the steps with identical contact sets and no capacity flag: max %.2e, p99.9 %.2e" % (syn[clean].max(), np.percentile(syn[clean], 99.9)))
    causes = {}
    for k in range(B):
        t, why = classify(rec[k])
        causes.setdefault(why, []).append((k, t, final[k]))
    print("cause of the first departure (> 1e-5) of the free-running fp32 kernel from the oracle, per env:")
    for why, lst in sorted(causes.items()):
        f = np.array([x[2] for x in lst])
        line = "  %-11s %3d envs; onset median step %s; final error median %.2e; > 1e-4 at the end: %d" % (
            why, len(lst), int(np.median([x[1] for x in lst])) if why != "clean" else "-", np.median(f), int((f > 1e-4).sum()))
        if ctl is not None:
            ks = [x[0] for x in lst]
            cb, cc = ctl[MODEL + "_B_1000"][ks], ctl[MODEL + "_C_1000"][ks]
            line += "; of these the fp64 controls lose (> 1e-4): rounded-state %d, +1 ulp %d, either %d" % (int((cb > 1e-4).sum()), int((cc > 1e-4).sum()), int(((cb > 1e-4) | (cc > 1e-4)).sum()))
        print(line)
    if ctl is not None:
        for mark in (100, 300, 1000):
            e = rec[:, mark - 1, 0]; cb = ctl["%s_B_%d" % (MODEL, mark)]; cc = ctl["%s_C_%d" % (MODEL, mark)]
            print("%4d steps, fraction <= 1e-4: fp32 kernel %.1f %% | fp64 control, fp32-rounded state %.1f %% | fp64 control, +1 ulp start %.1f %%;  median %.2e | %.2e | %.2e" % (
                mark, 100 * np.mean(e <= 1e-4), 100 * np.mean(cb <= 1e-4), 100 * np.mean(cc <= 1e-4), np.median(e), np.median(cb), np.median(cc)))
    if gpu is not None:
        for mark in (100, 300, 1000):
            g = gpu["err_%d" % mark]; e = rec[:len(g), mark - 1, 0]
            print("%4d steps, MI355X run of the same envs: <= 1e-4 %.1f %% (emulated kernel %.1f %%), median %.2e (%.2e); envs lost by both %d, only GPU %d, only emulation %d" % (
                mark, 100 * np.mean(g <= 1e-4), 100 * np.mean(e <= 1e-4), np.median(g), np.median(e), int(((g > 1e-4) & (e > 1e-4)).sum()), int(((g > 1e-4) & (e <= 1e-4)).sum()), int(((g <= 1e-4) & (e > 1e-4)).sum())))

"""TEST-INFRASTRUCTURE TOOL (CPU, fp64 oracle env only): WHY does an env of the closed-loop drift run (tools/env_drift.py) part from the oracle?

For every env beyond 1e-4 at the last mark, and every env with a touch-class mismatch, the tool re-runs the ORACLE env alone from the run's own
inputs and asks whether the fp64 oracle parts from ITSELF when its state at the start of the onset step is perturbed by the size of an fp32
rounding (U(-1, 1) x 2e-7 on the arm / finger angles, `draws` times):

  * how many perturbed twins end beyond 1e-4 of the unperturbed run (and their median / max end error),
  * the first substep at which a twin's contact LIST (geom pairs, in order) differs from the unperturbed run's, which contact appears /
    disappears (geom names, its depth at that moment) -- a grazing contact switching on or off one substep earlier or later,
  * or, without any difference in the contact lists, the growth of the perturbation per env step (smooth amplification through stiff contacts),

An env whose oracle twins all stay within 1e-5 while the HIP env parts by > 1e-4 would be a discrepancy of the kernel itself: listed as UNEXPLAINED.

  python tools/env_drift_causes.py <gpu.npz> <oracle.npz> [draws] [nproc] [control envs]  > profiles/r05_env_drift_causes.txt
"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np

import env_drift

EPS = 2e-7
_W = {}


def _names():
    names = {}
    for line in open(os.path.join(ROOT, "mujoco_jaco_amd", "assets", env_drift.MODEL + ".names.txt")):
        k, v = line.strip().split(": ", 1)
        names[k] = v.split()
    return names


def _geom_label(names, M, g):
    n = names["geom"][g]
    return n if n != "-" else "geom%d(%s)" % (g, names["body"][int(M["geom_bodyid"][g])])


class Twin:
    """OracleEnv with a per-substep record of the contact list."""

    def __init__(self, names, q0k, round_state=0):
        from oracle_env import OracleEnv
        self.oe = OracleEnv(names)
        if round_state:
            self.oe.o.option("round_state", round_state)   # 3: fp64 state, every forward pass evaluated at its fp32 rounding (oracle/jaco_oracle.c orc_step)
        self.oe.obj_goal = q0k[9:12].astype(np.float32).astype(np.float64)
        self.oe.dest_goal = np.array([q0k[16], q0k[17], 0.3468]).astype(np.float32).astype(np.float64)
        self.oe.set_state(q0k)

    def snapshot(self):
        o, oe = self.oe.o, self.oe
        return dict(qpos=o.get("qpos").copy(), qvel=o.get("qvel").copy(), ws=o.get("qacc_warmstart").copy(), mp=o.get("mocap_pos").copy(), mq=o.get("mocap_quat").copy(),
                    grip=oe.grip, steps=oe.steps, episodes=oe.episodes)

    def restore(self, s, dq=None):
        o, oe = self.oe.o, self.oe
        q = s["qpos"].copy()
        if dq is not None:
            q[:9] += dq
        o.set("qpos", q); o.set("qvel", s["qvel"]); o.set("qacc_warmstart", s["ws"]); o.set("mocap_pos", s["mp"]); o.set("mocap_quat", s["mq"])
        o.forward()
        oe.grip, oe.steps, oe.episodes = s["grip"], s["steps"], s["episodes"]

    def run(self, act, noise, first, last, record=False):
        """env steps first .. last - 1; returns qpos after each, and (record) per substep the contact pair list + depth list."""
        import glue
        oe, o = self.oe, self.oe.o
        qs, rec = [], []
        for s in range(first, last):
            if record:
                # the same loop as OracleEnv.step, unrolled to look at every substep (contacts of the step's own forward pass)
                orig_step = o.step

                def spy(ctrl=None, n=1, _orig=orig_step):
                    _orig(ctrl, n)
                    C = o.get("contact").reshape(-1, 11)
                    rec.append((s, [(int(c[7]), int(c[8])) for c in C], [float(c[0]) for c in C]))
                o.step = spy
                try:
                    _, _, d, _ = oe.step(act[s].astype(np.float64), noise[s + 1].astype(np.float64))
                finally:
                    del o.step
            else:
                _, _, d, _ = oe.step(act[s].astype(np.float64), noise[s + 1].astype(np.float64))
            qs.append(o.get("qpos").copy())
            if d:
                break
        return np.array(qs), rec


def emulated_step(k, snap, start, a, nz, names, q0k):
    """The fp32 kernel (host emulation of the unmodified kernel source, tests/emu) and the oracle take env steps `start`, `start + 1` from the
    SAME state (the oracle's snapshot, derived quantities refreshed by a forward pass on both sides): the error a single env step of fp32
    evaluation leaves, with the step's contact / row counts."""
    from emu_binding import EmuJacoEnv
    tw = Twin(names, q0k)
    tw.restore(snap)
    e = EmuJacoEnv()
    e.qpos[0], e.qvel[0], e.qacc_ws[0] = snap["qpos"], snap["qvel"], snap["ws"]
    e.task[0, 0] = e.task[0, 16] = snap["grip"]; e.task[0, 1] = snap["steps"]; e.task[0, 2] = snap["episodes"]
    e.task[0, 4:7] = tw.oe.obj_goal; e.task[0, 7:10] = tw.oe.dest_goal
    e.forward(nz[start][None])
    out, errs = [], []
    gq = _W.get("gpu_qpos")
    for s in range(start, len(a)):
        e.env_step(a[s], nz[s + 1][None])
        qo, _ = tw.run(a, nz, s, s + 1)
        err = np.abs(e.qpos[0].astype(np.float64) - qo[-1])
        hip = np.abs(gq[s, k].astype(np.float64) - qo[-1]).max() if gq is not None else float("nan")
        errs.append(err.max())
        if s < start + 2 or s == len(a) - 1:
            out.append("step %d: emulated %.1e (coordinate %d) / HIP %.1e; %d contacts, %d rows at its end%s" % (
                s + 1, err.max(), int(err.argmax()), hip, e.stats[0, 0], e.stats[0, 1], ", bigger tier" if e.flags[0] & 32 else ""))
        if e.done[0]:
            break
    tag = "the emulated kernel REPRODUCES the divergence: fp32 evaluation of this state's contact problem" if max(errs) > 1e-4 else (
        "the emulated kernel stays with the oracle (%.1e): the MI355X run's rounding (v_rcp / v_sqrt at 1 ulp, MFMA summation order) took another branch of the same sensitivity" % max(errs))
    return "fp32 kernel emulated on the host vs oracle from the SAME state at the start of step %d, free-running: %s -> %s" % (start + 1, "; ".join(out), tag)


def analyse(job):
    k, onset, kind = job
    q0, act, noise, names, M, nstep, draws = _W["q0"], _W["act"], _W["noise"], _W["names"], _W["M"], _W["nstep"], _W["draws"]
    a, nz = act[:, k], noise[:, k]
    base = Twin(names, q0[k])
    start = max(onset - 1, 0)            # perturb at the start of the step BEFORE the onset step: the onset is where the error first shows
    if start > 0:
        base.run(a, nz, 0, start)
    snap = base.snapshot()
    base.restore(snap)                   # (the twins restart from the snapshot through a forward pass: the unperturbed run does the same)
    qb, recb = base.run(a, nz, start, nstep, record=True)
    rng = np.random.default_rng(1000 + k)
    ends, flips, growth = [], [], []
    for t in range(draws):
        tw = Twin(names, q0[k])
        tw.restore(snap, rng.uniform(-1, 1, 9) * EPS)
        qt, rect = tw.run(a, nz, start, nstep, record=True)
        n = min(len(qt), len(qb))
        e = np.abs(qt[:n] - qb[:n]).max(1)
        ends.append(e[-1])
        growth.append(e)
        first = None
        for j, (rb, rt) in enumerate(zip(recb, rect)):
            if rb[1] != rt[1]:
                sb, st = set(rb[1]), set(rt[1])
                only = list(sb ^ st) or [p for p in rb[1] if rb[1].count(p) != rt[1].count(p)]
                pair = only[0]
                side, lst, dep = ("unperturbed", rb[1], rb[2]) if rb[1].count(pair) > rt[1].count(pair) else ("perturbed", rt[1], rt[2])
                depth = [d for p, d in zip(lst, dep) if p == pair]
                first = (rb[0], j - sum(1 for r in recb[:j] if r[0] < rb[0]), pair, side, min(depth) if depth else 0.0, len(rb[1]), len(rt[1]))
                break
        flips.append(first)
    ends = np.array(ends)
    nbad = int((ends > 1e-4).sum())
    with_flip = [f for f, e in zip(flips, ends) if f is not None]
    lines = []
    head = "env %5d (%s, HIP-vs-oracle onset in env step %d): %d of %d oracle twins perturbed by %.0e at the start of step %d end beyond 1e-4 (median %.1e, max %.1e)" % (
        k, kind, onset + 1, nbad, draws, EPS, start + 1, np.median(ends), ends.max())
    if with_flip:
        f = min(with_flip, key=lambda x: (x[0], x[1]))
        g1, g2 = f[2]
        cause = "contact list differs in %d twins, first in env step %d substep %d: %s <-> %s present on the %s side only, depth %.2e m (%d vs %d contacts)" % (
            len(with_flip), f[0] + 1, f[1] + 1, _geom_label(names, M, g1), _geom_label(names, M, g2), f[3], -f[4], f[5], f[6])
    else:
        g = np.array([gr[:min(len(x) for x in growth)] for gr in growth])
        per_step = np.median(g[:, 1:] / np.maximum(g[:, :-1], 1e-300), axis=0) if g.shape[1] > 1 else np.array([1.0])
        cause = "no difference in any twin's contact list: smooth amplification, x%.1f per env step at most (median over twins), %d contacts" % (per_step.max(), len(recb[0][1]) if recb else 0)
    emu = ""
    if nbad == 0:
        emu = "\n            " + emulated_step(k, snap, start, a, nz, names, q0[k])
    verdict = "EXPLAINED (the fp64 oracle parts from itself)" if nbad > 0 else ("sensitive (twins reach %.1e)" % ends.max() if ends.max() > 1e-5 else "UNEXPLAINED by this test")
    lines.append(head)
    lines.append("            " + cause + " -> " + verdict + emu)
    return k, nbad, ends.max(), bool(with_flip), "\n".join(lines)


def control_one(k):
    """Env k of the drift workload on the fp64 oracle twice: as it is, and with every forward pass evaluated at the fp32 rounding of its fp64
    state (no fp32 arithmetic anywhere): the end error of the pair and, when it is beyond 1e-4, the first substep whose contact lists differ."""
    q0, act, noise, names, M, nstep = _W["q0"], _W["act"], _W["noise"], _W["names"], _W["M"], _W["nstep"]
    a, nz = act[:, k], noise[:, k]
    qb, recb = Twin(names, q0[k]).run(a, nz, 0, nstep, record=True)
    qt, rect = Twin(names, q0[k], round_state=3).run(a, nz, 0, nstep, record=True)
    n = min(len(qb), len(qt))
    e = np.abs(qb[:n] - qt[:n]).max(1)
    msg = None
    if e[-1] > 1e-4:
        onset = int(np.argmax(e > 1e-5))
        first = "no contact-list difference before the end"
        for j, (rb, rt) in enumerate(zip(recb, rect)):
            if rb[1] != rt[1]:
                sb, st = set(rb[1]), set(rt[1])
                only = list(sb ^ st) or [p for p in rb[1] if rb[1].count(p) != rt[1].count(p)]
                g1, g2 = only[0]
                lst, dep = (rb[1], rb[2]) if rb[1].count(only[0]) > rt[1].count(only[0]) else (rt[1], rt[2])
                depth = min([d for p, d in zip(lst, dep) if p == only[0]] or [0.0])
                first = "contact lists first differ in env step %d substep %d: %s <-> %s, depth %.2e m (%d vs %d contacts)" % (
                    rb[0] + 1, j % 50 + 1, _geom_label(names, M, g1), _geom_label(names, M, g2), -depth, len(rb[1]), len(rt[1]))
                break
        lead = int(np.abs(qb[onset] - qt[onset]).argmax())
        msg = "  control env %5d: end error %.1e, onset env step %d (coordinate %d: %s); %s" % (
            k, e[-1], onset + 1, lead, "arm" if lead < 6 else ("fingers" if lead < 9 else ("object" if lead < 16 else "pedestal")), first)
    return k, float(e[-1]), msg


def control(nenv, nproc):
    """The closed-loop counterpart of bench.py's ctrl-level control: how often does the fp64 oracle env part from ITSELF by more than 1e-4 over the
    run when nothing but the evaluation point of its forward passes is rounded to fp32?  That share is the floor of any fp32 engine on this workload."""
    import multiprocessing as mp
    with mp.get_context("fork").Pool(nproc) as pool:
        res = pool.map(control_one, range(nenv), chunksize=4)
    ends = np.array([r[1] for r in res])
    print("control: fp64 oracle env vs the same oracle with fp32-rounded evaluation points, %d envs x %d env steps: <= 1e-4 at the end: %.2f %% (median %.1e, p90 %.1e, max %.1e); %d envs beyond 1e-4:" % (
        nenv, _W["nstep"], 100 * np.mean(ends <= 1e-4), np.median(ends), np.percentile(ends, 90), ends.max(), int((ends > 1e-4).sum())))
    for r in res:
        if r[2]:
            print(r[2])


def main():
    import multiprocessing as mp
    from mujoco_jaco_amd.modelc import blob
    g, r = dict(np.load(sys.argv[1])), dict(np.load(sys.argv[2]))
    draws = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    nproc = int(sys.argv[4]) if len(sys.argv) > 4 else max(1, (os.cpu_count() or 2) - 1)
    nstep, B = g["qpos"].shape[:2]
    q0, act, noise = env_drift.inputs(B, nstep)
    M = blob.load(os.path.join(ROOT, "mujoco_jaco_amd", "assets", env_drift.MODEL + ".jacomdl"))
    err = np.abs(g["qpos"].astype(np.float64) - r["qpos"]).max(2)
    dead = np.cumsum(g["done"].astype(bool) | r["done"].astype(bool), 0).astype(bool)
    live_end = ~dead[-1]
    bad = [int(k) for k in np.where(live_end & (err[-1] > 1e-4))[0]]
    tm = (g["obs"][..., 0] != r["obs"][..., 0]) & ~dead
    touch_envs = [int(k) for k in np.where(tm.any(0))[0]]
    jobs = [(k, int(np.argmax(err[:, k] > 1e-5)), "beyond 1e-4 at the end") for k in bad]
    jobs += [(k, int(np.argmax(tm[:, k])), "touch-class mismatch in step %d" % (int(np.argmax(tm[:, k])) + 1)) for k in touch_envs if k not in bad]
    _W.update(q0=q0, act=act, noise=noise, names=_names(), M=M, nstep=nstep, draws=draws, gpu_qpos=g["qpos"])
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    print("# tools/env_drift_causes.py: %d envs beyond 1e-4 after %d env steps + %d more with a touch-class mismatch (of %d envs); %d perturbed oracle twins each, eps %.0e" % (
        len(bad), nstep, len(jobs) - len(bad), B, draws, EPS), flush=True)
    with mp.get_context("fork").Pool(nproc) as pool:
        res = pool.map(analyse, jobs, chunksize=1)
    expl = sum(1 for x in res if x[1] > 0)
    sens = sum(1 for x in res if x[1] == 0 and x[2] > 1e-5)
    for x in res:
        print(x[4])
    if len(sys.argv) > 5:
        control(int(sys.argv[5]), nproc)
    print("summary: %d envs analysed; %d explained (the fp64 oracle itself ends beyond 1e-4 under an fp32-rounding-size perturbation), %d sensitive (twins beyond 1e-5), %d unexplained; "
          "%d with a contact switching on / off at a different substep" % (len(res), expl, sens, len(res) - expl - sens, sum(1 for x in res if x[3])))


if __name__ == "__main__":
    main()

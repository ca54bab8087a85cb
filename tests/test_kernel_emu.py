"""CPU tier: the *unmodified* HIP kernel source, compiled for the host against the lockstep 64-lane wavefront
emulator (tests/emu), checked against the fp64 oracle.  This is development/test infrastructure: the product
library has no CPU path.  Tolerances are fp32-vs-fp64 single-step bounds (state re-synchronised every step);
free-running drift is measured on the GPU tier.
"""
import numpy as np
import pytest

from emu_binding import EmuEnv
from mujoco_jaco_amd import workload
from oracle_binding import Oracle


def _sync_step(o, e, ctrl, **kw):
    st = [o.get(n) for n in ("qpos", "qvel", "qacc_warmstart")]
    e.qpos[0], e.qvel[0], e.qacc_ws[0] = st
    e.step(ctrl, **kw)
    o.step(ctrl)
    return np.abs(o.get("qpos") - e.qpos[0]).max(), np.abs(o.get("qvel") - e.qvel[0]).max()


def test_smooth_dynamics_single_step(model_arrays):
    o = Oracle(); o.option("disable_contact", 1)
    e = EmuEnv()
    rng = np.random.default_rng(0)
    for _ in range(4):
        q = model_arrays["qpos0"].copy(); q[:9] += rng.uniform(-1, 1, 9) * 0.5
        q[9:12] = [0, .65, .5]; q[12:16] = rng.normal(size=4); q[12:16] /= np.linalg.norm(q[12:16]); q[16:19] = [.4, .3, .6]
        o.set("qpos", q); o.set("qvel", rng.normal(size=21) * 0.3); o.set("qacc_warmstart", np.zeros(21))
        ctrl = np.concatenate([rng.uniform(-1, 1, 6) * 5, rng.uniform(0, 1.5, 3)])
        eq, ev = _sync_step(o, e, ctrl, disable_contact=True)
        assert eq < 1e-6 and ev < 1e-3   # qvel: finger dofs have 1e-5 kg m^2 inertia, accelerations ~1e3 rad/s^2


def test_contact_pipeline_matches_oracle_on_reset_distribution(model_arrays):
    o = Oracle(); e = EmuEnv()
    q = workload.reset_states(model_arrays["qpos0"], 6, seed=11)
    c = workload.random_ctrl(6, seed=12, scale=0.2)
    worst = 0.0
    for k in range(6):
        o.reset(); o.set("qpos", q[k])
        for i in range(25):
            eq, ev = _sync_step(o, e, c[k])
            assert (e.stats[0, 0], e.stats[0, 1]) == (o.ncon, o.nefc), (k, i)   # same contacts, same rows
            worst = max(worst, eq)
    assert worst < 2e-6 and e.flags[0] == 0


def test_in_hand_grasp_with_hull_contacts(model_arrays, names):
    """Placing-style reset (env_mujoco_util.py:106-117): object put between the fingers -> ~45 contacts / 230 rows,
    condim-6 hull contacts through MPR, touch sensors."""
    from mujoco_jaco_amd.modelc import rot
    o = Oracle(); e = EmuEnv()
    q = model_arrays["qpos0"].copy()
    q[:6] = [1.3, 3.85, 1.05, 2.05, 1.5, -1.15]; q[6:9] = 0.6; q[16:18] = [.4, .3]
    o.set("qpos", q); o.forward()
    b = names["body"].index("EE_obj")
    xp = o.get("xpos").reshape(-1, 3)[b]; xq = o.get("xquat").reshape(-1, 4)[b]
    q[9:12] = xp + rot.quat_to_mat(xq) @ np.array([-0.04, 0, 0]); q[12:16] = xq
    o.set("qpos", q)
    errs, sens = [], []
    for i in range(60):
        g = min(1.0, 0.6 + 0.004 * i)
        eq, _ = _sync_step(o, e, np.array([0, 0, 0, 0, 0, 0, g, g, g]))
        errs.append(eq)
        if (e.stats[0, 0], e.stats[0, 1]) == (o.ncon, o.nefc):
            sens.append(np.abs(o.get("sensordata") - e.sensordata[0]).max() / max(1.0, o.get("sensordata").max()))
    assert (e.flags[0] & 31) == 0   # bit 32 only says the 256-row tier was used
    assert np.median(errs) < 1e-6 and max(errs) < 2e-4      # a grazing contact may flip for one step
    assert np.median(sens) < 1e-3


def test_box_box_contact_lists_match_oracle(model_arrays):
    """collide_box_box4 (up to four box-box pairs per pass, collision.h) against the oracle's box-box, contact by contact and in order:
    the object box dropped in random attitudes onto the pedestal's top, its rim (edge-edge contacts) and next to the holder, so that a
    substep holds one to three box-box pairs (object-pedestal, object-holder, pedestal-floor is plane-box) in face and edge regimes."""
    from mujoco_jaco_amd.modelc import rot
    rng = np.random.default_rng(5)
    o = Oracle(); e = EmuEnv()
    nd = JDBG = None
    L = e.L
    off_ncon = 3 * 11 + 9 * 11 + 21 * 21 + 5 * 24
    seen_edge = seen_face = multi = 0
    for trial in range(40):
        q = model_arrays["qpos0"].copy()
        q[16:19] = [0.0, 0.345, 0.0899]                 # pedestal next to the holder (0, 0.6), 0.1 mm into the floor (at 0.09 exactly the
                                                        # presence of its floor contacts hangs on the rounding of 0.09 + 0.07 - 0.16)
        ang = rng.uniform(-np.pi, np.pi, 3) * (0.15 if trial % 3 == 0 else 1.0)
        qq = rot.euler_to_quat(ang) if hasattr(rot, "euler_to_quat") else None
        if qq is None:
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax); th = np.linalg.norm(ang)
            qq = np.concatenate([[np.cos(th / 2)], np.sin(th / 2) * ax])
        where = trial % 4
        top = 0.09 + 0.07 + 0.16                         # pedestal top
        if where == 0: pos = [rng.uniform(-0.06, 0.06), 0.345 + rng.uniform(-0.06, 0.06), top + 0.03]          # on the top face
        elif where == 1: pos = [0.1 + rng.uniform(-0.01, 0.01), 0.345 + rng.uniform(-0.05, 0.05), top + 0.02]  # over the rim
        elif where == 2: pos = [rng.uniform(-0.05, 0.05), 0.447 + rng.uniform(-0.004, 0.004), 0.17 + 0.03]     # in the gap: holder top and pedestal side
        else: pos = [rng.uniform(-0.1, 0.1), 0.6 + rng.uniform(-0.1, 0.1), 0.17 + 0.035]                      # on the holder
        q[9:12] = pos; q[12:16] = qq
        qf = q.astype(np.float32).astype(np.float64)
        o.reset(); o.set("qpos", qf); o.forward()
        e.qpos[0] = qf; e.qvel[0] = 0; e.qacc_ws[0] = 0
        e.step(np.zeros(9), nsub=1, dbg_env=0)
        D = e.dbg
        ncon = int(D[off_ncon])
        oc = o.get("contact").reshape(-1, 11)
        assert ncon == o.ncon == len(oc), (trial, ncon, o.ncon)
        C = D[off_ncon + 4:off_ncon + 4 + 8 * ncon].reshape(ncon, 8)
        if ncon:
            assert np.abs(C[:, 0] - oc[:, 0]).max() < 2e-6 and np.abs(C[:, 1:4] - oc[:, 1:4]).max() < 2e-6, trial
            assert np.abs(C[:, 4:7] - oc[:, 4:7]).max() < 2e-5, trial
        pairs = {(int(a), int(b)) for a, b in oc[:, 7:9]}
        multi += len(pairs) >= 3
        per_pair = [int(((oc[:, 7] == a) & (oc[:, 8] == b)).sum()) for a, b in pairs]
        seen_edge += any(n == 1 for n in per_pair); seen_face += any(n >= 3 for n in per_pair)
    assert seen_edge >= 3 and seen_face >= 10 and multi >= 5, (seen_edge, seen_face, multi)


def test_arm_only_model_config2():
    o = Oracle("jaco2_reaching_torque"); o.option("disable_contact", 1)
    e = EmuEnv("jaco2_reaching_torque")
    assert (e.nq, e.nv) == (9, 9)
    rng = np.random.default_rng(3)
    q = e.M["qpos0"].copy(); q[:6] = [1.5, 3.9, 1.3, 2.0, 2.0, 1.5]
    o.set("qpos", q)
    for i in range(20):
        eq, ev = _sync_step(o, e, np.concatenate([rng.uniform(-1, 1, 6) * [30, 30, 30, 15, 15, 15], [0.6] * 3]), disable_contact=True)
        assert eq < 1e-6


def test_bitwise_deterministic(model_arrays):
    q = workload.reset_states(model_arrays["qpos0"], 1, seed=5)[0]
    out = []
    for _ in range(2):
        e = EmuEnv(); e.qpos[0] = q
        e.step(workload.random_ctrl(1, seed=6)[0], nsub=5)
        out.append((e.qpos.copy(), e.qvel.copy()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_huge_tier_keeps_every_row_of_a_hand_in_pedestal_reset(model_arrays):
    """A picking reset that spawns the hand inside the pedestal (1 % of draws): 64 contacts / 292 constraint rows in the first
    step, beyond the 256-row heavy tier.  The 512-row tier keeps them all (MuJoCo's njmax is 8000): same counts as the oracle,
    no capacity flag, and the violent ejection stays on the oracle's trajectory."""
    q = workload.reset_states(model_arrays["qpos0"], 256, seed=41, f32_draws=True)[200]
    c = workload.random_ctrl(256, seed=42, scale=0.2)[200].astype(np.float32).astype(np.float64)
    o = Oracle(); e = EmuEnv()
    o.reset(); o.set("qpos", q); o.set("ctrl", c); e.qpos[0] = q
    seen, worst = 0, 0.0
    for t in range(12):
        e.step(c); o.step()
        assert (e.stats[0, 0], e.stats[0, 1]) == (o.ncon, o.nefc), t
        seen = max(seen, o.nefc)
        worst = max(worst, np.abs(e.qpos[0] - o.get("qpos")).max())
    # free-running over 12 substeps of a ~300-row ejection (accelerations ~1e4 rad/s^2): measured 1.1e-5 with body-space rows (dense rows of
    # rounds 1-4: below 1e-5); bound 3x
    print("hand-in-pedestal ejection, 12 free-running substeps: worst qpos error %.2e" % worst)
    assert worst < 3.4e-5, worst
    assert seen > 256 and (e.flags[0] & 7) == 0


def test_jaco2_torque_model_on_the_d12_build_matches_oracle():
    """Sibling MJCF jaco2_torque.xml: 6 arm + 6 finger hinges in one tree, the distal finger joints sprung and damped
    (xml:109-133), no free bodies.  The same kernel sources built for that layout (-DJNB=12 -DJNV=12 ..., jaco/model_dev.h) against
    the fp64 oracle: free motion under motor torques with the fingers driven into their joint limits (limit rows)."""
    from emu_binding import EmuEnv
    from oracle_binding import Oracle
    e = EmuEnv("jaco2_torque", 1)
    o = Oracle("jaco2_torque")
    assert (e.nq, e.nv, e.nu) == (12, 12, 9)
    q = e.M["qpos0"].copy(); q[:6] = [1.2, 3.9, 1.3, 2.0, 1.5, 1.0]; q[6:12] = [1.15, -0.45, 0.8, 0.1, 1.0, 0.43]       # thumb proximal / distal and pinky distal start beyond their limits (1.1, -0.4, 0.4)
    q = q.astype(np.float32).astype(np.float64)
    c = np.array([3., -5., 2., 1., -1., 0.5, 0.0, 0.0, 0.0])
    e.qpos[0] = q
    o.set("qpos", q); o.set("qvel", np.zeros(12)); o.set("qacc_warmstart", np.zeros(12))
    rows = 0
    for k in range(6):
        e.step(c, nsub=25)
        o.step(c, n=25)
        assert (e.stats[0, 0], e.stats[0, 1]) == (o.ncon, o.nefc) and not (e.flags[0] & 31)
        rows = max(rows, o.nefc)
        assert np.abs(e.qpos[0] - o.get("qpos")).max() < 2e-6 and np.abs(e.qvel[0] - o.get("qvel")).max() < 2e-4, k
    assert rows >= 1                                                # the scenario really exercised limit rows
    assert np.abs(o.get("qfrc_passive")[6:]).max() > 0.005           # ... and the springs / dampers are at work at the end


def test_sensor_model_with_cylinder_geoms_on_the_d12_build_matches_oracle():
    """Sibling MJCF jaco2_curtain_torque_sensor.xml: the 12-hinge arm of jaco2_torque.xml + one free object + 61 touch sensors, and the
    only cylinder geoms among the reference's loadable assets (xml:62-74 the blocker's posts and rod, the rod turned by `zaxis`; xml:278-281
    the object holder's stem and disc).  Cylinders go through the convex-convex path (support function: rim point towards the direction's
    radial part).  Stepped by the d12 build (13 bodies / 18 dofs in blocks 12 + 6) against the fp64 oracle: (A) the object box dropped,
    slightly tilted, onto the holder's disc -- box-cylinder contact, 300 substeps free-running; (B) poses in which an arm hull touches a
    post / the stem / the disc (tests/golden/sensor_post_poses.npz), single steps re-synchronised: same contact and row counts."""
    import os
    from emu_binding import EmuEnv
    from oracle_binding import Oracle
    o = Oracle("jaco2_curtain_torque_sensor"); e = EmuEnv("jaco2_curtain_torque_sensor")
    assert (e.nq, e.nv, e.nu, e.ns) == (19, 18, 9, 61)
    assert sorted(np.unique(e.M["geom_type"]).tolist()) == [0, 2, 5, 6, 7]
    q = e.M["qpos0"].copy(); q[12:15] = [0.02, 0.66, 0.445]; q[15:19] = np.array([1, 0.05, -0.03, 0.1]) / np.linalg.norm([1, 0.05, -0.03, 0.1])
    q = q.astype(np.float32).astype(np.float64)
    e.qpos[0] = q; o.set("qpos", q); o.set("qvel", np.zeros(18)); o.set("qacc_warmstart", np.zeros(18))
    c = np.array([1.0, -2.0, 1.0, 0.5, -0.5, 0.2, 0.3, 0.3, 0.3])
    touched = 0
    for k in range(12):                                                     # (A)
        e.step(c, nsub=25); o.step(c, n=25)
        assert (e.stats[0, 0], e.stats[0, 1]) == (o.ncon, o.nefc) and not (e.flags[0] & 31), (k, e.stats[0], o.ncon, o.nefc)
        touched += o.ncon
        assert np.abs(e.qpos[0] - o.get("qpos")).max() < 2e-6 and np.abs(e.qvel[0] - o.get("qvel")).max() < 2e-4, k
    assert touched >= 10
    P = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sensor_post_poses.npz"))["qpos"]
    rng = np.random.default_rng(3)
    errs = []
    touch = 0.0
    for q in np.concatenate([P[:6], P[10:]]):                               # (B) (the last four: a finger part with a touch site on a cylinder)
        o.reset(); o.set("qpos", q); o.set("qvel", np.zeros(18)); o.set("qacc_warmstart", np.zeros(18))
        e.flags[0] = 0
        ctrl = np.concatenate([rng.uniform(-1, 1, 6) * 3, q[6:12:2]])
        for i in range(3):
            eq, ev = _sync_step(o, e, ctrl)
            assert (e.stats[0, 0], e.stats[0, 1]) == (o.ncon, o.nefc) and o.ncon >= 2, (i, e.stats[0], o.ncon, o.nefc)
            errs.append(eq)
            so = o.get("sensordata")
            # same sensors respond; the reading is one contact's share of a load that two near-redundant contacts of one finger part carry
            # (condim 6 + condim 3 on the same cylinder): an ill-conditioned split, measured up to 4 % apart in fp32 (0.77 vs 0.74)
            assert np.array_equal(so > 0, e.sensordata[0] > 0) and np.abs(so - e.sensordata[0]).max() < 0.05 * max(1.0, so.max()), (i, so.max())
            touch = max(touch, so.max())
        assert (e.flags[0] & 31) == 0
    assert touch > 1.0                                                      # the touch stage saw cylinder contacts
    print("sensor model, arm-on-cylinder poses: single-step qpos error median %.1e max %.1e" % (np.median(errs), max(errs)))
    assert np.median(errs) < 1e-6 and max(errs) < 1e-4


def test_curtain_old_model_on_the_d12_build_matches_oracle():
    """Sibling MJCF jaco2_curtain_torque_old.xml: the 12-hinge arm + one free object among 22 static boxes, 20 touch sensors; its rest pose
    has the distal finger joints beyond their range (limit rows from the first step).  d12 build, 60 single steps re-synchronised with the
    oracle, the object lying on the floor: same contact and row counts every step."""
    o = Oracle("jaco2_curtain_torque_old"); e = EmuEnv("jaco2_curtain_torque_old")
    assert (e.nq, e.nv, e.nu, e.ns) == (19, 18, 9, 20)
    q = e.M["qpos0"].copy(); q[12:15] = [0.3, 0.3, 0.0301]
    q = q.astype(np.float32).astype(np.float64)
    o.set("qpos", q); o.set("qvel", np.zeros(18)); o.set("qacc_warmstart", np.zeros(18))
    c = np.array([3., -5., 2., 1., -1., 0.5, 0.5, 0.5, 0.5])
    worst, rows = 0.0, 0
    for k in range(60):
        eq, ev = _sync_step(o, e, c)
        assert (e.stats[0, 0], e.stats[0, 1]) == (o.ncon, o.nefc), (k, e.stats[0], o.ncon, o.nefc)
        worst, rows = max(worst, eq), max(rows, o.nefc)
    assert (e.flags[0] & 31) == 0 and rows >= 19 and worst < 3e-6   # measured 1.1e-6 (finger joints under limit rows: |qacc| ~ 1e3)


def test_contact_free_kernel_with_joint_limit_rows():
    """The contact-free instantiation's Newton solve (stage_newton_limits: every row a joint-limit row held by its dof's lane) against
    the oracle: arm-only model driven into the limits of joints 1 and 2 (range [0.87, 5.41] / [0.33, 5.95]) and of the finger joints
    (servo command beyond their range), single steps re-synchronised with the oracle, row counts equal."""
    o = Oracle("jaco2_reaching_torque"); o.option("disable_contact", 1)
    e = EmuEnv("jaco2_reaching_torque")
    q = e.M["qpos0"].copy(); q[:6] = [1.5, 0.88, 5.94, 2.0, 2.0, 1.5]; q[6:9] = [1.5, 0.01, 0.7]
    v = np.zeros(9); v[1] = -3.0; v[2] = 2.5; v[6] = 4.0; v[7] = -5.0
    o.set("qpos", q); o.set("qvel", v)
    ctrl = np.array([5.0, -30.0, 30.0, 3.0, -2.0, 1.0, 1.6, -0.2, 0.7])
    rows, worst_q, worst_v = 0, 0.0, 0.0
    for i in range(40):
        eq, ev = _sync_step(o, e, ctrl, disable_contact=True)
        assert e.stats[0, 1] == o.nefc and e.stats[0, 0] == 0, (i, e.stats[0], o.nefc)
        rows = max(rows, o.nefc); worst_q = max(worst_q, eq); worst_v = max(worst_v, ev)
    assert rows >= 3 and e.flags[0] == 0
    assert worst_q < 1e-6 and worst_v < 2e-3, (worst_q, worst_v)


def test_separating_direction_cache_does_not_change_results(model_arrays, names):
    """Hull narrowphase with the per-pair separating-direction cache (collision.h): only provable misses are skipped, so a free run of
    the in-hand grasp scenario (fingers closing onto the object: hull hits, near misses that come and go) must give the same state,
    contact / row counts and sensor values BIT FOR BIT with the cache on and off -- while spending fewer support queries."""
    import ctypes
    from mujoco_jaco_amd.modelc import rot
    o = Oracle()
    q = model_arrays["qpos0"].copy()
    q[:6] = [1.3, 3.85, 1.05, 2.05, 1.5, -1.15]; q[6:9] = 0.6; q[16:18] = [.4, .3]
    o.set("qpos", q); o.forward()
    b = names["body"].index("EE_obj")
    xp = o.get("xpos").reshape(-1, 3)[b]; xq = o.get("xquat").reshape(-1, 4)[b]
    q[9:12] = xp + rot.quat_to_mat(xq) @ np.array([-0.04, 0, 0]); q[12:16] = xq
    runs, queries = [], []
    for on in (1, 0):
        e = EmuEnv()
        e.L.emu_set_sep_cache.argtypes = [ctypes.c_int]
        e.L.emu_get_counter.argtypes = [ctypes.c_int, ctypes.c_int]; e.L.emu_get_counter.restype = ctypes.c_long
        e.L.emu_set_sep_cache(on)
        try:
            e.qpos[0] = q
            for i in (3, 4, 5, 6, 7): e.L.emu_get_counter(i, 1)
            trace = []
            for i in range(50):
                g = min(1.0, 0.6 + 0.006 * i)
                e.step(np.array([0.5, -0.3, 0.2, 0.1, 0, 0, g, g, g]), nsub=1)
                trace.append((e.qpos.copy(), e.qvel.copy(), e.stats[:, :2].copy(), e.sensordata.copy()))
            runs.append(trace)
            queries.append((e.L.emu_get_counter(3, 1), e.L.emu_get_counter(7, 1)))
        finally:
            e.L.emu_set_sep_cache(1)
    for ta, tb in zip(*runs):
        for x, y in zip(ta, tb):
            assert np.array_equal(x, y)
    (calls_on, q_on), (calls_off, q_off) = queries
    print("hull pairs: %d narrowphase calls; support queries %d with the cache, %d without" % (calls_on, q_on, q_off))
    assert calls_on == calls_off and q_on < q_off   # (this scenario is dominated by hits; the saving under the shipped policy is ~a quarter of all queries: tools/mpr_query_stats.py kernel)


def test_dual_arm_model_on_the_d30_build_matches_oracle():
    """Sibling MJCF jaco2_dual_torque.xml (xml:48-49 includes the two arms): 20 fused bodies, 30 dofs (two 9-dof arm trees in
    one dof block + two free objects), 106 geoms, 3 332 pairs, 18 actuators, 40 touch sensors -- stepped by the d30 build of the same kernel
    sources (frame, geom and pair stages in several 64-lane passes).  Single steps re-synchronised with the oracle: (A) free fall of the
    objects and torque-driven arms, (B) objects resting on their holders (12 contacts / 48 rows) under random arm poses, (C) poses in which
    the two arms touch each other (tests/golden/dual_cross_poses.npz: 13..47 contacts, up to ~340 rows incl. condim-6 hull contacts that
    couple both arm trees; medium / heavy / huge tiers): same contact and row counts as the oracle in every step."""
    import os
    o = Oracle("jaco2_dual_torque"); e = EmuEnv("jaco2_dual_torque")
    assert (e.nq, e.nv, e.nu, e.ns) == (32, 30, 18, 40)
    rng = np.random.default_rng(1)

    def run(q, nsteps, ctrl):
        o.reset(); o.set("qpos", q); o.set("qvel", np.zeros(30)); o.set("qacc_warmstart", np.zeros(30))
        e.flags[0] = 0
        wq = wv = 0.0
        maxrows = 0
        for i in range(nsteps):
            eq, ev = _sync_step(o, e, ctrl)
            assert (e.stats[0, 0], e.stats[0, 1]) == (o.ncon, o.nefc), (i, e.stats[0], o.ncon, o.nefc)
            wq, wv, maxrows = max(wq, eq), max(wv, ev), max(maxrows, o.nefc)
        assert (e.flags[0] & 31) == 0
        return wq, wv, maxrows

    q0 = e.M["qpos0"].copy()
    ctrl = np.concatenate([rng.uniform(-1, 1, 6) * 5, [0.6] * 3, rng.uniform(-1, 1, 6) * 5, [1.0] * 3])
    wq, wv, _ = run(q0, 6, ctrl)                                            # (A)
    assert wq < 1e-6 and wv < 2e-4
    lo, hi = np.array([0.7, 3.8, 1.0, 1.8, 1.0, 0.8]), np.array([2.5, 4.0, 1.7, 2.5, 2.5, 2.3])
    for k in range(2):                                                      # (B)
        q = q0.copy(); q[18:21] = [-0.5, 0.6, 0.2001]; q[25:28] = [0.5, 0.6, 0.2001]
        q[0:6] = rng.uniform(lo, hi); q[9:15] = rng.uniform(lo, hi)
        wq, wv, rows = run(q, 8, np.concatenate([rng.uniform(-1, 1, 6) * 8, [0.8] * 3, rng.uniform(-1, 1, 6) * 8, [0.6] * 3]))
        assert wq < 1e-6 and wv < 2e-4 and rows == 48
    P = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dual_cross_poses.npz"))["qpos"]
    errs, most = [], 0
    for q in P[:6]:                                                         # (C)
        wq, wv, rows = run(q, 6, np.concatenate([rng.uniform(-1, 1, 6) * 3, [0.7] * 3, rng.uniform(-1, 1, 6) * 3, [0.7] * 3]))
        errs.append(wq); most = max(most, rows)
    print("dual arm, arm-on-arm poses: single-step qpos error per pose", ["%.1e" % x for x in errs], "most rows", most)
    # the geoms of these poses overlap by 1-7 cm (|qacc| ~ 5e3-2e4 rad/s^2): single-step errors of 2e-7 .. 2e-6, one pose / torque draw in
    # a dozen an order above that (same contact set, fp32 solve of ~300 ill-conditioned rows)
    assert np.median(errs) < 2e-6 and max(errs) < 1e-4
    assert most > 128                                                       # the bigger tiers carried it


def test_pair_list_on_the_d30_build_keeps_floor_contacts():
    """d30 layout (3 584-pair table: pair-list entries pack 12 + 7 + 7 bits, the plane flag sits at bit 26 where the default layout has it at
    bit 22).  A list pass over a chunk with floor pairs must test them in plane form: both objects lie on the floor, the arms hang over
    them; 40 free-running substeps with the pair list on / off and the separating-direction cache on / off must agree BIT FOR BIT and keep the
    objects' floor contacts.  (A chunk's plane flag was once read from the default layout's bit 22 = bit 3 of g2 here; this scene alone does not
    expose that -- chunk 0 also holds entries with that bit set -- the fix is the shared expression JPL_KBITS + 2 JPL_GBITS; the test is the d30
    build's on / off regression for both execution options.)"""
    import ctypes
    runs = []
    for pl_on, sc_on in ((1, 1), (0, 1), (1, 0)):
        e = EmuEnv("jaco2_dual_torque")
        e.L.emu_set_pair_list.argtypes = [ctypes.c_int]; e.L.emu_set_sep_cache.argtypes = [ctypes.c_int]
        e.L.emu_get_counter.argtypes = [ctypes.c_int, ctypes.c_int]; e.L.emu_get_counter.restype = ctypes.c_long
        e.L.emu_set_pair_list(pl_on); e.L.emu_set_sep_cache(sc_on)
        e.L.emu_get_counter(0, 1); e.L.emu_get_counter(1, 1)
        try:
            q = e.M["qpos0"].copy()
            q[18:21] = [-0.5, 1.0, 0.03]; q[21:25] = [1, 0, 0, 0]
            q[25:28] = [0.5, 1.0, 0.03]; q[28:32] = [1, 0, 0, 0]
            e.qpos[0] = q
            ctrl = np.concatenate([[2, -3, 1, 0.5, -0.5, 0.2], [0.8] * 3, [-2, 3, -1, 0.5, 0.5, -0.2], [0.6] * 3]).astype(np.float32)
            trace = []
            for _ in range(2):
                e.step(ctrl, nsub=20)
                trace.append((e.qpos.copy(), e.qvel.copy(), e.stats.copy(), e.flags.copy(), e.sensordata.copy()))
            runs.append((trace, e.L.emu_get_counter(0, 1), e.L.emu_get_counter(1, 1)))
        finally:
            e.L.emu_set_pair_list(1); e.L.emu_set_sep_cache(1)
    (ta, full_on, list_on), (tb, _, list_off), (tc, _, _) = runs
    for other in (tb, tc):
        for x, y in zip(ta, other):
            for u, v in zip(x, y):
                assert np.array_equal(u, v)
    assert list_off == 0 and list_on > 30, (full_on, list_on)        # the list carried nearly every substep
    assert ta[-1][2][0, 3] >= 8                                      # both objects still rest on the floor (4 contacts each) at the end
    assert abs(ta[-1][0][0, 20] - 0.03) < 1e-4 and abs(ta[-1][0][0, 27] - 0.03) < 1e-4


def test_wrench_rows_build_option_matches_oracle(model_arrays, names):
    """The body-space ("wrench") constraint rows (physics_kernel.h, -DJACO_WRENCH=1: a row as its 6-vector wrench + two body ids, every product
    with J formed from it) are an A/B build option, measured slower on MI355X and not shipped (profiles/r05_ab_wrench_rows.txt).  This keeps the
    option honest: the same kernel sources built with it, single steps re-synchronised with the oracle -- reset distribution (plane-box, box-box,
    limit rows) and the in-hand grasp (condim-6 hull contacts, ~230 rows through the bigger tiers): same contact / row counts, same bounds as the
    dense rows."""
    from mujoco_jaco_amd.modelc import rot
    o = Oracle(); e = EmuEnv(layout="_wrench")
    q = workload.reset_states(model_arrays["qpos0"], 3, seed=11)
    c = workload.random_ctrl(3, seed=12, scale=0.2)
    worst = 0.0
    for k in range(3):
        o.reset(); o.set("qpos", q[k])
        for i in range(12):
            eq, ev = _sync_step(o, e, c[k])
            assert (e.stats[0, 0], e.stats[0, 1]) == (o.ncon, o.nefc), (k, i)
            worst = max(worst, eq)
    assert worst < 2e-6 and e.flags[0] == 0
    qg = model_arrays["qpos0"].copy()
    qg[:6] = [1.3, 3.85, 1.05, 2.05, 1.5, -1.15]; qg[6:9] = 0.6; qg[16:18] = [.4, .3]
    o.reset(); o.set("qpos", qg); o.forward()
    b = names["body"].index("EE_obj")
    xp = o.get("xpos").reshape(-1, 3)[b]; xq = o.get("xquat").reshape(-1, 4)[b]
    qg[9:12] = xp + rot.quat_to_mat(xq) @ np.array([-0.04, 0, 0]); qg[12:16] = xq
    o.set("qpos", qg); o.set("qvel", np.zeros(21)); o.set("qacc_warmstart", np.zeros(21))
    most = 0
    for i in range(6):
        eq, ev = _sync_step(o, e, np.concatenate([np.zeros(6), [1.0] * 3]))
        assert (e.stats[0, 0], e.stats[0, 1]) == (o.ncon, o.nefc), i
        most = max(most, o.nefc)
        assert eq < 5e-6
    assert most > 128 and (e.flags[0] & 15) == 0


def test_newton_look_ahead_stop_saves_hessian_builds_and_keeps_the_solution(model_arrays):
    """The solver ends a solve when the NEXT round's improvement -- exactly (1 - alpha)^2 times this round's when the step switched no row on or
    off -- is below MuJoCo's tolerance (physics_kernel.h newton_next_round_is_idle).  Against the build without the rule
    (-DJACO_NEWTON_LOOKAHEAD=0, on demand): fewer matrix-core Hessian builds per constrained solve, and the solution (qacc, compared with the fp64
    oracle's after every synchronised substep of the contact workload) is as close to the oracle's as before."""
    import ctypes
    res = {}
    q = workload.reset_states(model_arrays["qpos0"], 6, seed=31)
    c = workload.random_ctrl(6, seed=32, scale=0.2)
    for layout in ("", "_nolook"):
        o = Oracle(); e = EmuEnv(layout=layout)
        e.L.emu_get_counter.argtypes = [ctypes.c_int, ctypes.c_int]; e.L.emu_get_counter.restype = ctypes.c_long
        e.L.emu_get_counter(8, 1); e.L.emu_get_counter(9, 1)
        errs, solves = [], 0
        for k in range(6):
            o.reset(); o.set("qpos", q[k])
            for i in range(25):
                _sync_step(o, e, c[k])
                a = o.get("qacc_warmstart")
                errs.append(np.abs(a - e.qacc_ws[0]).max() / max(1.0, np.abs(a).max()))
                solves += int(e.stats[0, 2])
        res[layout] = (e.L.emu_get_counter(8, 1), e.L.emu_get_counter(9, 1), solves, float(np.median(errs)), float(np.max(errs)))
    print("constrained solves, Hessian builds, Newton steps, qacc error vs oracle (median, max): with the look-ahead stop %s, without %s" % (res[""], res["_nolook"]))
    (n1, b1, s1, med1, max1), (n0, b0, s0, med0, max0) = res[""], res["_nolook"]
    assert n1 == n0 == 150
    assert b1 <= b0 - 0.9 * n0, (b1, b0)     # one Hessian build fewer per solve (measured on this transient workload -- the object is being pushed out of its spawn overlap: 320 against 470)
    assert med1 <= 2 * med0 + 1e-7 and max1 <= 2 * max0 + 1e-6, (med1, med0, max1, max0)


def test_mpr_pairs_do_not_change_results(model_arrays, names):
    """Build option -DJACO_MPR_PAIRS=1 (measured on MI355X: no gain, not shipped; kept honest here): hull candidates go through MPR two at a time,
    one per half wave (collision.h mpr_pair2; option "mpr_pairs"), their results parked for the in-order contact loop.  Per pair the
    arithmetic, its order and the tie-breaking are those of the one-pair routine, so a free run of the in-hand grasp scenario (fingers closing onto
    the object: ~10 hull pairs per substep, hits, near misses, cached separating directions coming and going) must give the same state, contact /
    row counts and sensor values BIT FOR BIT with the option on and off -- and the two-pair routine must really have run."""
    import ctypes
    from mujoco_jaco_amd.modelc import rot
    o = Oracle()
    q = model_arrays["qpos0"].copy()
    q[:6] = [1.3, 3.85, 1.05, 2.05, 1.5, -1.15]; q[6:9] = 0.6; q[16:18] = [.4, .3]
    o.set("qpos", q); o.forward()
    b = names["body"].index("EE_obj")
    xp = o.get("xpos").reshape(-1, 3)[b]; xq = o.get("xquat").reshape(-1, 4)[b]
    q[9:12] = xp + rot.quat_to_mat(xq) @ np.array([-0.04, 0, 0]); q[12:16] = xq
    for cache in (1, 0):
        runs, passes, calls = [], [], []
        for on in (1, 0):
            e = EmuEnv(layout="_mprpairs")   # (the build option's own emulator library, made on demand)
            e.L.emu_set_mpr_pairs.argtypes = [ctypes.c_int]; e.L.emu_set_sep_cache.argtypes = [ctypes.c_int]
            e.L.emu_get_counter.argtypes = [ctypes.c_int, ctypes.c_int]; e.L.emu_get_counter.restype = ctypes.c_long
            e.L.emu_set_mpr_pairs(on); e.L.emu_set_sep_cache(cache)
            try:
                e.qpos[0] = q
                for i in (3, 4, 10): e.L.emu_get_counter(i, 1)
                trace = []
                for i in range(40):
                    g = min(1.0, 0.6 + 0.006 * i)
                    e.step(np.array([0.5, -0.3, 0.2, 0.1, 0, 0, g, g, g]), nsub=1)
                    trace.append((e.qpos.copy(), e.qvel.copy(), e.stats[:, :2].copy(), e.sensordata.copy(), e.flags.copy()))
                runs.append(trace)
                passes.append(e.L.emu_get_counter(10, 1)); calls.append((e.L.emu_get_counter(3, 1), e.L.emu_get_counter(4, 1)))
            finally:
                e.L.emu_set_mpr_pairs(1); e.L.emu_set_sep_cache(1)
        for i, (ta, tb) in enumerate(zip(*runs)):
            for x, y in zip(ta, tb):
                assert np.array_equal(x, y), (cache, i)
        print("separating-direction cache %d: %d passes of the two-pair routine over 40 substeps; MPR runs / hits %s with it, %s without" % (cache, passes[0], calls[0], calls[1]))
        assert passes[0] > 80 and passes[1] == 0 and calls[0] == calls[1]

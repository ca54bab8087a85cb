"""Golden vectors for the `releasing`, `carrying` and `pushing` task branches, from the reference's OWN Python (same stand-ins as
make_glue_vectors.py; build container only).  Pins: _get_terminal_inspection (releasing env_mujoco_util.py:551-566 incl. the object
velocity test, carrying :549-550, pushing :583-584; all 3-tuples in the reference), _get_reward (0 for all three, :432-439), the
ranges of _create_init_angle (:181-189: carrying shares the placing / grasping branch, releasing has nine values incl. the fingers
at 0.6; pushing has no branch there -- the call raises, recorded as such), the action width of JacoMujocoEnv (env_mujoco.py:79-89)
and its episode length (:18-23).  Writes tests/golden/glue_vectors_releasing.npz (data only)."""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_glue_vectors import install_stubs, make_util, random_scene  # noqa: E402


def main(out):
    warnings.simplefilter("ignore")
    install_stubs()
    rng = np.random.default_rng(20260404)
    N = 160
    rec = {k: [] for k in ("ee", "eeq", "obj", "q2", "touch", "nsteps", "dest_goal", "objvel", "term_rel", "term_carry", "term_push", "reward")}
    for k in range(N):
        u = make_util("releasing", rng)
        ee, q, obj = random_scene(u, rng, near=(k % 4 == 0))
        touch = int(rng.integers(0, 4))
        vel = rng.normal(size=3) * 0.02
        if k % 3 == 1:     # object let go over the pedestal: success (slow, close), still moving, wrong place
            off = rng.choice([0.01, 0.03, 0.05, 0.08])
            ang = rng.uniform(0, 2 * np.pi)
            obj = np.array([u.dest_goal[0][0] + off * np.cos(ang), u.dest_goal[0][1] + off * np.sin(ang), rng.choice([0.34, 0.349, 0.36, 0.15])])
            touch = int(rng.choice([0, 0, 0, 2]))
            vel = rng.normal(size=3) * rng.choice([0.001, 0.004, 0.02])
        if k % 11 == 5:
            obj[2] = 0.05   # dropped
        u.interface.xyz["object_body"] = obj
        u.interface.objvel = vel
        u.interface.q[2] = rng.choice([1.3, np.pi + rng.uniform(-.12, .12)], p=[0.85, 0.15])
        u.touch_index = touch
        u.num_episodes = int(rng.integers(0, 450))
        u._JacoMujocoEnvUtil__get_gripper_pose()
        n0 = u.num_episodes
        rec["ee"].append(ee); rec["eeq"].append(q); rec["obj"].append(obj); rec["q2"].append(u.interface.q[2]); rec["touch"].append(touch)
        rec["nsteps"].append(n0); rec["dest_goal"].append(u.dest_goal[0]); rec["objvel"].append(vel)
        rw = []
        for task, key in (("releasing", "term_rel"), ("carrying", "term_carry"), ("pushing", "term_push")):
            u.task = task; u.num_episodes = n0
            t = u._get_terminal_inspection()
            rec[key].append(np.array([float(t[0]), float(t[1]), float(t[2]), float(len(t))]))
            rw.append(float(u._get_reward()))
        rec["reward"].append(rw)
    G = {"r_" + k: np.array(v) for k, v in rec.items()}
    # _create_init_angle: ranges over many draws of the reference's own function (global numpy RNG seeded)
    for task in ("releasing", "carrying", "placing"):
        u = make_util(task, rng)
        np.random.seed(7)
        A = np.array([u._create_init_angle() for _ in range(4096)], dtype=np.float64)
        G["init_%s_min" % task], G["init_%s_max" % task] = A.min(0), A.max(0)
    u = make_util("pushing", rng)
    try:
        u._create_init_angle()
        G["init_pushing_raises"] = np.array([0])
    except UnboundLocalError:
        G["init_pushing_raises"] = np.array([1])
    # JacoMujocoEnv's action width and episode length per task (env_mujoco.py:18-23,79-89), from its own constructor logic
    from env_script import env_mujoco
    widths, lengths = [], []
    for task in ("carrying", "releasing", "pushing"):
        e = object.__new__(env_mujoco.JacoMujocoEnv)
        e.n_robots = 1

        def fake_super_init(self, **kw):
            pass
        orig = env_mujoco.JacoMujocoEnvUtil.__init__
        env_mujoco.JacoMujocoEnvUtil.__init__ = fake_super_init
        import builtins
        real_open = builtins.open
        builtins.open = lambda *a, **k: real_open(os.devnull, "w")   # the constructor opens a csv logger under ./logger_csv
        try:
            env_mujoco.JacoMujocoEnv.__init__(e, task=task)
        finally:
            env_mujoco.JacoMujocoEnvUtil.__init__ = orig
            builtins.open = real_open
        widths.append(len(e.act_max)); lengths.append(e.task_max_steps)
    G["act_width"], G["task_max_steps"] = np.array(widths), np.array(lengths)
    np.savez_compressed(out, **G)
    print("wrote", out, {k: v.shape for k, v in G.items()})
    print("releasing bonus values:", np.unique(np.round(G["r_term_rel"][:, 1], 0), return_counts=True), "tuple lengths", np.unique(G["r_term_rel"][:, 3]),
          "pushing init raises:", G["init_pushing_raises"], "act widths", widths, "episode lengths", lengths)


if __name__ == "__main__":
    main(os.path.join(os.path.dirname(os.path.abspath(__file__)), "glue_vectors_releasing.npz"))

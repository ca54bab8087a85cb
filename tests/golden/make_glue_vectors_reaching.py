"""Golden vectors for the `reaching` task branches, from the reference's OWN Python (same stand-ins as make_glue_vectors.py;
build container only).  Pins: _get_reward (reaching, env_mujoco_util.py:314-351), _get_terminal_inspection (reaching,
:504-520; a 3-tuple in the reference), __sample_goal's reaching goal incl. the float16 cast (:192-207), and the
rulebased_subgoal = False branch of _get_observation (:255-270).  Writes tests/golden/glue_vectors_reaching.npz (data only)."""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_glue_vectors import install_stubs, make_util, random_scene  # noqa: E402


def main(out):
    warnings.simplefilter("ignore")
    install_stubs()
    rng = np.random.default_rng(20260202)
    N = 96
    rec = {k: [] for k in ("ee", "eeq", "goal", "q2", "nsteps", "reward", "term", "obs", "grip", "obj", "dest_goal", "sens")}
    for k in range(N):
        u = make_util("reaching", rng)
        ee, q, obj = random_scene(u, rng)
        if k % 4 == 0:    # near the goal: the success / angle branches
            u.reaching_goal[0][:3] = ee + rng.normal(size=3) * 0.01
            from scipy.spatial.transform import Rotation
            e = Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_euler("XYZ")
            u.reaching_goal[0][3:] = e + rng.normal(size=3) * rng.choice([0.05, 0.5])
        if k % 5 == 1:    # low / close to the base: the penalty branches
            ee = np.array([rng.uniform(-.08, .08), rng.uniform(-.08, .08), rng.uniform(0.02, 0.25)])
            u.interface.xyz["EE"] = ee
        if k % 7 == 2:    # non-canonical goal pitch, as __sample_goal produces (beta = acos(.) * sign can exceed pi/2)
            u.reaching_goal[0][4] = rng.uniform(1.7, 3.0) * rng.choice([-1, 1])
        u.interface.q[2] = rng.choice([1.3, np.pi + rng.uniform(-.12, .12)], p=[0.8, 0.2])
        u.num_episodes = int(rng.integers(0, 400))
        u._JacoMujocoEnvUtil__get_gripper_pose()
        rec["ee"].append(ee); rec["eeq"].append(q); rec["goal"].append(u.reaching_goal[0].copy()); rec["q2"].append(u.interface.q[2])
        rec["nsteps"].append(u.num_episodes); rec["obj"].append(obj); rec["dest_goal"].append(u.dest_goal[0]); rec["grip"].append(u.gripper_angle[0])
        rec["reward"].append(u._get_reward())
        t = u._get_terminal_inspection()
        rec["term"].append(np.array([float(t[0]), float(t[1]), float(t[2])]))
        u.rulebased_subgoal = False
        sens = rng.choice([0.0, 0.0, 0.0, 1.0], 20)
        u.interface.sensors = {"EE_touch": sens[0], **{"%d_touch" % i: sens[1 + i] for i in range(19)}}
        rec["sens"].append(sens)
        rec["obs"].append(u._get_observation())
    G = {"r_" + k: np.array(v) for k, v in rec.items()}
    # ---- __sample_goal: replay the global numpy RNG to recover the draws the reference consumed, in its order
    from env_script import env_mujoco_util as ref
    draws, goals = [], []
    for k in range(64):
        u = make_util("reaching", rng)
        u.goal_buffer = None
        u.interface.set_dest_xyz = lambda xy: None
        seed = 7000 + k
        np.random.seed(seed)
        rg, og, dg = u._JacoMujocoEnvUtil__sample_goal()
        np.random.seed(seed)
        mx = ref.uniform(0.3, 0.42); sx = ref.choice([-1, 1]); my = ref.uniform(0.3, 0.42); sy = ref.choice([-1, 1]); z = ref.uniform(0.3, 0.5)
        gamma = ref.uniform(-0.1, 0.1)
        draws.append([mx, sx, my, sy, z, gamma]); goals.append(rg[0])
    G["g_draws"], G["g_goal"] = np.array(draws), np.array(goals, dtype=np.float64)
    np.savez_compressed(out, **G)
    print("wrote", out, {k: v.shape for k, v in G.items()})


if __name__ == "__main__":
    main(os.path.join(os.path.dirname(os.path.abspath(__file__)), "glue_vectors_reaching.npz"))

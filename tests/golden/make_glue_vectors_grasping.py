"""Golden vectors for the `grasping` and `pickAndplace` task branches, from the reference's OWN Python (same stand-ins as
make_glue_vectors.py; build container only).  Pins: _get_reward (grasping, env_mujoco_util.py:352-391), _get_terminal_inspection
(grasping :521-536, pickAndplace :585-600; 3-tuples in the reference, pickAndplace with its `picked` flag), and the quantities the
grasping reset's pre-reach loops compute (:124-129 target orientation incl. the float16 cast, :142-150 distance / quaternion
difference).  Writes tests/golden/glue_vectors_grasping.npz (data only)."""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_glue_vectors import install_stubs, make_util, random_scene, _Transformations  # noqa: E402


def main(out):
    warnings.simplefilter("ignore")
    install_stubs()
    rng = np.random.default_rng(20260303)
    N = 128
    rec = {k: [] for k in ("ee", "eeq", "obj", "q2", "touch", "nsteps", "dest_goal", "reward", "term_grasp", "picked_in", "term_pp", "picked_out",
                           "obj_goal", "gamma", "reach_ori", "goal_ori", "dist", "angdiff")}
    for k in range(N):
        u = make_util("grasping", rng)
        ee, q, obj = random_scene(u, rng, near=(k % 2 == 0))
        touch = int(rng.integers(0, 4))
        if k % 5 == 1:     # lifted with a grasp: success branches
            obj = ee + rng.normal(size=3) * 0.02; obj[2] = rng.choice([0.27, 0.30]); touch = int(rng.choice([1, 3]))
        if k % 5 == 2:     # released on the pedestal: pickAndplace success / near misses
            obj = np.array([u.dest_goal[0][0] + rng.choice([0.01, 0.03, 0.05]), u.dest_goal[0][1], rng.choice([0.34, 0.36])]); touch = 0
        if k % 7 == 3:
            obj[2] = 0.05   # dropped
        u.interface.xyz["object_body"] = obj
        u.interface.q[2] = rng.choice([1.3, np.pi + rng.uniform(-.12, .12)], p=[0.85, 0.15])
        u.touch_index = touch
        u.num_episodes = int(rng.integers(0, 1000))
        u._JacoMujocoEnvUtil__get_gripper_pose()
        n0 = u.num_episodes
        rec["ee"].append(ee); rec["eeq"].append(q); rec["obj"].append(obj); rec["q2"].append(u.interface.q[2]); rec["touch"].append(touch)
        rec["nsteps"].append(n0); rec["dest_goal"].append(u.dest_goal[0])
        rec["reward"].append(u._get_reward())
        t = u._get_terminal_inspection()
        rec["term_grasp"].append(np.array([float(t[0]), float(t[1]), float(t[2])]))
        u.task = "pickAndplace"; u.num_episodes = n0
        u.picked = bool(rng.integers(0, 2))
        rec["picked_in"].append(float(u.picked))
        t = u._get_terminal_inspection()
        rec["term_pp"].append(np.array([float(t[0]), float(t[1]), float(t[2])]))
        rec["picked_out"].append(float(u.picked))
        # pre-reach quantities (the statements of :124-129 and :142-150 evaluated by the reference's own numpy / transformations stand-in)
        obj_goal, gamma = u.obj_goal[0], rng.uniform(-0.1, 0.1)
        x, y, z = obj_goal - u.gripper_pose[0][:3]
        alpha = -np.arcsin(y / np.sqrt(y ** 2 + z ** 2)) * np.sign(x)
        beta = np.arccos(x / np.linalg.norm([x, y, z])) * np.sign(x)
        reach_goal_ori = np.array([alpha, beta, gamma], dtype=np.float16)
        T = _Transformations
        grip = T.unit_vector(T.quaternion_from_euler(*u.gripper_pose[0][3:6], axes="rxyz"))
        tar = T.unit_vector(T.quaternion_from_euler(*u.reaching_goal[0][3:6], axes="rxyz"))
        rec["obj_goal"].append(obj_goal); rec["gamma"].append(gamma); rec["reach_ori"].append(reach_goal_ori.astype(np.float64))
        rec["goal_ori"].append(u.reaching_goal[0][3:6].copy())
        rec["dist"].append(np.linalg.norm(u.gripper_pose[0][:3] - obj_goal)); rec["angdiff"].append(np.linalg.norm(grip - tar))
    G = {"g_" + k: np.array(v) for k, v in rec.items()}
    G["task_max_steps"] = np.array([500, 1200])   # env_mujoco.py:18-23: grasping, pickAndplace
    np.savez_compressed(out, **G)
    print("wrote", out, {k: v.shape for k, v in G.items()})
    print("grasp done/succ-bonus counts:", np.unique(G["g_term_grasp"][:, 1].round(0) > 100, return_counts=True), "pp bonus values:", np.unique(G["g_term_pp"][:, 1]))


if __name__ == "__main__":
    main(os.path.join(os.path.dirname(os.path.abspath(__file__)), "glue_vectors_grasping.npz"))

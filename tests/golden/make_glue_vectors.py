"""Generate glue-tier golden vectors by running the reference's OWN Python (read-only import from
/root/reference) behind stand-ins for the third-party modules that are absent from this image
(abr_control, mujoco_py, gym).  Only *data* (inputs + outputs) is written to tests/golden/glue_vectors.npz;
no reference source travels.  Run here (needs /root/reference); the GPU box only sees the .npz.

Pins: _get_rotation, _get_touch, _get_reward (picking), _get_terminal_inspection (picking, placing) incl. the
step-count timeout of JacoMujocoEnv.terminal_inspection, _get_rulebased_subgoal / _get_observation (with the global
numpy RNG seeded so the 6 uniform draws per call are reproducible), _take_action (target pose, gripper ramp, mocap writes).
"""
import sys
import types
import warnings

import numpy as np
from scipy.spatial.transform import Rotation

REF = "/root/reference"


# ---- stand-ins for absent third-party modules -------------------------------------------------------------
def _wxyz_to_rot(q):
    q = np.asarray(q, dtype=np.float64)
    return Rotation.from_quat([q[1], q[2], q[3], q[0]])


class _Transformations:  # abr_control.utils.transformations == Gohlke; 'rxyz' == SciPy intrinsic 'XYZ' 
    @staticmethod
    def euler_from_quaternion(q, axes="rxyz"):
        assert axes == "rxyz"
        return tuple(_wxyz_to_rot(q).as_euler("XYZ"))

    @staticmethod
    def quaternion_from_euler(a, b, c, axes="rxyz"):
        assert axes == "rxyz"
        x, y, z, w = Rotation.from_euler("XYZ", [a, b, c]).as_quat()
        return np.array([w, x, y, z])

    @staticmethod
    def unit_vector(v):
        v = np.asarray(v, dtype=np.float64)
        return v / np.linalg.norm(v)


def install_stubs():
    abr = types.ModuleType("abr_control"); ctr = types.ModuleType("abr_control.controllers"); utl = types.ModuleType("abr_control.utils")
    ctr.OSC = object
    utl.transformations = _Transformations
    abr.controllers, abr.utils = ctr, utl
    mjp = types.ModuleType("mujoco_py"); gen = types.ModuleType("mujoco_py.generated"); gen.const = types.SimpleNamespace()
    mjp.generated = gen
    gym = types.ModuleType("gym")
    gym.spaces = types.SimpleNamespace(Box=lambda lo, hi, dtype=None: types.SimpleNamespace(low=lo, high=hi, shape=np.shape(lo)))
    for n, m in {"abr_control": abr, "abr_control.controllers": ctr, "abr_control.utils": utl, "mujoco_py": mjp,
                 "mujoco_py.generated": gen, "mujoco_py.generated.const": gen.const, "gym": gym}.items():
        sys.modules[n] = m
    if REF not in sys.path:
        sys.path.insert(0, REF)


class FakeInterface:
    """What the glue reads from the sim: body poses, sensors, joint feedback; records mocap writes."""

    def __init__(self):
        self.xyz, self.quat, self.sensors, self.q, self.objvel = {}, {}, {}, np.zeros(6), np.zeros(3)
        self.mocap = {}
        outer = self

        class _Data:
            def get_sensor(self, name):
                return outer.sensors.get(name, 0.0)
        self.sim = types.SimpleNamespace(data=_Data())

    def get_xyz(self, name, object_type="body"):
        return np.copy(self.xyz[name])

    def get_orientation(self, name, object_type="body"):
        return np.copy(self.quat[name])

    def get_feedback(self):
        return {"q": np.copy(self.q), "dq": np.zeros(6)}

    def get_obj_vel(self, idx=0):
        return np.copy(self.objvel)

    def set_mocap_xyz(self, name, xyz):
        self.mocap[name + "_pos"] = np.array(xyz, dtype=np.float64)

    def set_mocap_orientation(self, name, quat):
        self.mocap[name + "_quat"] = np.array(quat, dtype=np.float64)


def make_util(task, rng):
    from env_script.env_mujoco_util import JacoMujocoEnvUtil
    u = object.__new__(JacoMujocoEnvUtil)
    u.n_robots, u.task, u.skip_frames = 1, task, 50
    u.interface = FakeInterface()
    u.object_z = 0.1898
    u.rulebased_subgoal, u.subgoal_obs = True, False
    u.reward_method, u.reward_module = None, None
    u.gripper_angle = np.ones(1) * 0.6
    u.gripper_angle_array = np.zeros((1, 50))
    u.gripper_iter, u.touch_index, u.num_episodes, u.action_in, u.picked = 0, 0, 0, False, False
    u.base_position = np.array([[0.0, 0.0, 0.157]])
    return u


def random_scene(u, rng, near=False):
    ee = np.array([rng.uniform(-.3, .3), rng.uniform(.3, .7), rng.uniform(.15, .5)])
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    obj = ee + rng.normal(size=3) * 0.03 if near else np.array([rng.uniform(-.1, .1), .65 + rng.uniform(-.08, .02), rng.choice([0.1998, 0.05, 0.28, 0.2, 0.34])])
    it = u.interface
    it.xyz["EE"], it.quat["EE"], it.xyz["object_body"] = ee, q, obj
    qo = rng.normal(size=4); it.quat["object_body"] = qo / np.linalg.norm(qo)
    it.q = np.array([rng.uniform(.7, 2.5), rng.uniform(3.8, 4), rng.choice([rng.uniform(1, 1.7), np.pi + rng.uniform(-.12, .12)]), 2.0, 1.5, 1.0])
    it.objvel = rng.normal(size=3) * 0.02
    u.obj_goal = np.array([[rng.uniform(-.1, .1), .65 + rng.uniform(-.08, .02), 0.1898]])
    u.dest_goal = np.array([[.4 + rng.uniform(-.05, .05), .3 + rng.uniform(-.05, .05), .3468]])
    u.reaching_goal = np.array([np.concatenate([rng.uniform(-.4, .4, 3), rng.uniform(-1, 1, 3)])])
    u.gripper_angle = np.array([rng.uniform(0.6, 1.0)])
    return ee, q, obj


def main(out):
    warnings.simplefilter("ignore")
    install_stubs()
    rng = np.random.default_rng(20260101)
    G = {}
    # ---- _get_rotation
    u = make_util("picking", rng)
    rin = np.concatenate([rng.uniform(-3, 3, (32, 3)), rng.normal(size=(32, 3))], 1); inv = rng.integers(0, 2, 32)
    G["rot_in"], G["rot_inv"] = rin, inv
    G["rot_out"] = np.array([u._get_rotation(r[0], r[1], r[2], r[3:6], bool(i)) for r, i in zip(rin, inv)])
    # ---- _get_touch: sensordata order of the XML: EE_touch, 0_touch ... 18_touch
    S = np.zeros((64, 20))
    for k in range(64):
        n = rng.integers(0, 4)
        idx = rng.integers(0, 20, n)
        S[k, idx] = rng.choice([0.0009, 0.0011, 0.5, 3.0], n)
    out_t = []
    for s in S:
        u.interface.sensors = {"EE_touch": s[0], **{"%d_touch" % i: s[1 + i] for i in range(19)}}
        out_t.append(u._get_touch())
    G["touch_sens"], G["touch_class"] = S, np.array(out_t)
    # ---- reward (picking), terminal (picking / placing), observation, subgoal, take_action
    N = 96
    keys = ["ee", "eeq", "obj", "q2", "objvel", "obj_goal", "dest_goal", "grip", "touch", "nsteps", "reward", "term_pick", "term_place",
            "noise", "obs", "sub_pos", "sub_ori", "act", "target", "grip_after", "ramp", "mocap_sub_pos", "mocap_sub_quat", "mocap_hand_pos", "mocap_hand_quat", "noise_act"]
    rec = {k: [] for k in keys}
    for k in range(N):
        u = make_util("picking", rng)
        ee, q, obj = random_scene(u, rng, near=(k % 3 == 0))
        touch = int(rng.integers(0, 4))
        if k % 6 == 1:   # object released on the pedestal: placing success / wrong-place branches
            off = rng.choice([0.005, 0.05])
            obj = np.array([u.dest_goal[0][0] + off, u.dest_goal[0][1], rng.choice([0.34, 0.15])])
            u.interface.xyz["object_body"] = obj
            touch = 0
            u.interface.q[2] = 1.3
        u.touch_index = touch
        u.num_episodes = int(rng.integers(0, 600))
        u._JacoMujocoEnvUtil__get_gripper_pose()
        rec["ee"].append(ee); rec["eeq"].append(q); rec["obj"].append(obj); rec["q2"].append(u.interface.q[2]); rec["objvel"].append(u.interface.objvel)
        rec["obj_goal"].append(u.obj_goal[0]); rec["dest_goal"].append(u.dest_goal[0]); rec["grip"].append(u.gripper_angle[0]); rec["touch"].append(touch)
        rec["nsteps"].append(u.num_episodes)
        rec["reward"].append(u._get_reward())
        n0 = u.num_episodes
        rec["term_pick"].append(np.array(u._get_terminal_inspection(), dtype=np.float64))
        u.task = "placing"; u.num_episodes = n0
        rec["term_place"].append(np.array(u._get_terminal_inspection(), dtype=np.float64))
        u.task = "picking"
        # observation with the 6 uniform draws of _get_rulebased_subgoal made reproducible
        seed = 1000 + k
        np.random.seed(seed); noise = np.random.uniform(size=6)
        sens = rng.choice([0.0, 0.0, 0.0, 1.0], 20)
        u.interface.sensors = {"EE_touch": sens[0], **{"%d_touch" % i: sens[1 + i] for i in range(19)}}
        np.random.seed(seed)
        obs = u._get_observation()
        rec["noise"].append(np.concatenate([noise, sens])); rec["obs"].append(obs)
        np.random.seed(seed)
        sp, so = u._get_rulebased_subgoal()
        rec["sub_pos"].append(sp); rec["sub_ori"].append(so)
        # take_action
        a = rng.uniform(-1, 1, 7)
        g0 = u.gripper_angle[0]
        np.random.seed(seed + 5000); na = np.random.uniform(size=6); np.random.seed(seed + 5000)
        u._take_action(a)
        rec["act"].append(np.concatenate([a, [g0]])); rec["target"].append(u.target_pos.copy()); rec["grip_after"].append(u.gripper_angle[0])
        rec["ramp"].append(u.gripper_angle_array[0].copy()); rec["noise_act"].append(na)
        m = u.interface.mocap
        rec["mocap_sub_pos"].append(m["subgoal_reach_pos"]); rec["mocap_sub_quat"].append(m["subgoal_reach_quat"])
        rec["mocap_hand_pos"].append(m["hand_pos"]); rec["mocap_hand_quat"].append(m["hand_quat"])
    for k, v in rec.items():
        G["s_" + k] = np.array(v)
    # ---- JacoMujocoEnv.terminal_inspection timeout (env_mujoco.py:144-150): picking/placing 700 steps
    G["timeout_steps"] = np.array([698, 699, 700, 701])
    G["timeout_expect_done_at"] = np.array([700])   # current_steps is incremented first; < task_max_steps continues
    np.savez_compressed(out, **G)
    print("wrote", out, {k: v.shape for k, v in G.items() if k.startswith("s_") is False})


if __name__ == "__main__":
    import os
    main(os.path.join(os.path.dirname(os.path.abspath(__file__)), "glue_vectors.npz"))

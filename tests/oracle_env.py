"""TEST INFRASTRUCTURE: fp64 single-env restatement of JacoMujocoEnv.step (env_script/env_mujoco.py:116-139) on top of
the physics oracle and oracle/glue.py, including the one-substep staleness of everything the controller and the
observation read from mjData."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import glue  # noqa: E402
from oracle_binding import Oracle  # noqa: E402


def _mocap_id(names, body):
    """mocap id = rank of the body among the mocap bodies, in body order (MuJoCo numbering); read from the compiled model."""
    from mujoco_jaco_amd.modelc import blob
    M = blob.load(os.path.join(ROOT, "mujoco_jaco_amd", "assets", "jaco2_curtain_torque.jacomdl"))
    return int(M["body_mocapid"][names["body"].index(body)])


class OracleEnv:
    def __init__(self, names, task="picking", frame_skip=50):
        self.o = Oracle()
        self.task, self.frame_skip = task, frame_skip
        self.ee = names["body"].index("EE")
        self.obj = names["body"].index("object_body")
        self.ee_obj = names["body"].index("EE_obj") if "EE_obj" in names["body"] else -1
        self.base = np.array([0.0, 0.0, 0.157])
        # mocap ids of the two markers _take_action moves (their contype-8 geoms collide with the EE axis sticks)
        self.mk_hand, self.mk_sub = _mocap_id(names, "hand"), _mocap_id(names, "subgoal_reach")
        self.grip, self.steps, self.episodes = 0.6, 0, 0
        self.obj_goal, self.dest_goal = np.zeros(3), np.zeros(3)
        self.reach_goal = np.zeros(6)          # task 'reaching' / the rulebased_subgoal = False observation branch
        self.rulebased = True
        self.picked = [False]                  # task 'pickAndplace' (self.picked)
        self.target = None                     # task 'grasping': the EE target the reset's pre-reach leaves behind (self.target_pos)

    def set_state(self, qpos, qvel=None, qacc_ws=None):
        o = self.o
        o.set("qpos", qpos); o.set("qvel", np.zeros(o.nv) if qvel is None else qvel)
        o.set("qacc_warmstart", np.zeros(o.nv) if qacc_ws is None else qacc_ws)
        o.forward()   # sim.forward(): derived quantities of the current state

    def _ee(self):
        o = self.o
        return o.get("xpos").reshape(-1, 3)[self.ee].copy(), o.get("xquat").reshape(-1, 4)[self.ee].copy()

    def observe(self, noise6):
        o = self.o
        pe, qe = self._ee()
        obj = o.get("xpos").reshape(-1, 3)[self.obj]
        touch = glue.touch_class(o.get("sensordata"))
        return glue.observation(self.task, touch, pe, qe, self.grip, obj, self.dest_goal, self.obj_goal, noise6,
                                reach_goal=None if self.rulebased else self.reach_goal), touch, pe, qe, obj

    def step(self, action, noise12):
        o = self.o
        pe, qe = self._ee()
        # set_mocap_*("subgoal_reach", rule-based sub-goal), then the new target, then set_mocap_*("hand", target)
        obj_y = o.get("xpos").reshape(-1, 3)[self.obj][1]
        spos, sori = glue.rulebased_subgoal(self.task, pe, self.obj_goal, obj_y, self.dest_goal, np.asarray(noise12[:6], np.float64))
        target, grip, ramp = glue.take_action(pe, qe, action, self.grip, self.frame_skip)
        self.grip = grip
        mp, mq = o.get("mocap_pos").reshape(-1, 3).copy(), o.get("mocap_quat").reshape(-1, 4).copy()
        if self.rulebased:
            mp[self.mk_sub], mq[self.mk_sub] = spos, glue.quat_from_euler(*sori)
        mp[self.mk_hand], mq[self.mk_hand] = target[:3], glue.quat_from_euler(*target[3:6])
        o.set("mocap_pos", mp.reshape(-1)); o.set("mocap_quat", mq.reshape(-1))
        for k in range(self.frame_skip):
            q_dq = o.get("qvel")[:6]
            jp, jr = o.jac_body_com(self.ee)                       # stale: last forward pass
            J = np.vstack([jp[:, :6], jr[:, :6]])
            M = o.get("qM").reshape(o.nv, o.nv)[:6, :6]
            bias = o.get("qfrc_bias")[:6]
            pe, qe = self._ee()
            u = glue.osc_generate(q_dq, target, J, M, bias, pe, qe)
            o.step(np.concatenate([u, [ramp[k]] * 3]))
        obs, touch, pe, qe, obj = self.observe(noise12[6:12])
        rew = glue.reward_picking(pe, qe, obj, touch) if self.task == "picking" else (
            glue.reward_reaching(pe, qe, self.reach_goal, self.base) if self.task == "reaching" else (
                glue.reward_grasping(pe, qe, obj, touch) if self.task == "grasping" else 0.0))
        done, bonus, wb, succ = glue.env_terminal(self.task, self.steps, o.get("qpos")[2], pe, obj, self.dest_goal, touch, self.episodes, self.base,
                                                  ee_quat=qe, reach_goal=self.reach_goal, picked=self.picked, obj_vel=o.get("qvel")[9:12])
        self.steps += 1
        if self.steps < (700 if self.task in ("picking", "placing") else (1200 if self.task == "pickAndplace" else 500)):
            self.episodes += 1
        return obs, rew + bonus, done, succ

    def _osc(self, target):
        o = self.o
        jp, jr = o.jac_body_com(self.ee)
        J = np.vstack([jp[:, :6], jr[:, :6]])
        M = o.get("qM").reshape(o.nv, o.nv)[:6, :6]
        pe, qe = self._ee()
        return glue.osc_generate(o.get("qvel")[:6], target, J, M, o.get("qfrc_bias")[:6], pe, qe)

    def grasping_prereach(self, gamma, cap=4000):
        """The pre-reach loops of the grasping reset (env_mujoco_util.py:123-170) from the current state (set_state has run sim.forward()):
        returns the number of substeps taken.  Loop 1 calls stop_obj (free-body velocities zeroed + sim.forward(): the controller reads
        the *current* state's M, J, bias) before every _step_simulation; loop 2 steps with what the previous sim.step left in mjData."""
        o = self.o
        pe, qe = self._ee()
        ori = glue.grasp_reach_ori(pe, self.obj_goal, gamma)
        target = np.concatenate([self.obj_goal, ori])
        self.grip, n, g3 = 0.6, 0, [0.6] * 3
        while n < cap:
            v = o.get("qvel").copy(); v[9:] = 0
            o.set("qvel", v); o.forward()
            o.step(np.concatenate([self._osc(target), g3])); n += 1
            pe, qe = self._ee()
            dist, ang = glue.grasp_prereach_conditions(pe, qe, self.obj_goal, self.reach_goal[3:6])
            if dist < 0.2:
                target = np.concatenate([pe, ori])
                break
            if ang < np.pi / 6:
                break
        while n < cap:
            o.step(np.concatenate([self._osc(target), g3])); n += 1
            pe, qe = self._ee()
            target = np.concatenate([self.obj_goal, ori])
            if np.linalg.norm(pe - self.obj_goal) < 0.15:
                break
        self.target = target
        return n

    def placing_hold(self, nsub=150):
        """The object part of the placing reset (env_mujoco_util.py:106-117): object into the grasp frame, `nsub` controlled
        substeps (gripper command 0.6) each followed by set_obj_xyz, i.e. re-pin + zero free-body velocities + sim.forward(),
        so the controller always reads the *current* state's M, J, bias (no staleness inside this loop)."""
        o = self.o
        po = o.get("xpos").reshape(-1, 3)[self.ee_obj].copy()
        qo = o.get("xquat").reshape(-1, 4)[self.ee_obj].copy()
        eul = glue.euler_from_quat(qo)
        quat = glue.quat_from_euler(*eul)
        pos = po + glue.get_rotation(eul[0], eul[1], eul[2], [-0.04, 0, 0])
        pe, qe = self._ee()

        def pin():
            q, v = o.get("qpos").copy(), o.get("qvel").copy()
            q[9:12], q[12:16] = pos, quat
            v[9:] = 0
            o.set("qpos", q); o.set("qvel", v)
            o.forward()
        pin()
        pe, qe = self._ee()
        target = np.concatenate([pe, glue.euler_from_quat(qe)])
        self.grip = 0.6
        for _ in range(nsub):
            jp, jr = o.jac_body_com(self.ee)
            J = np.vstack([jp[:, :6], jr[:, :6]])
            M = o.get("qM").reshape(o.nv, o.nv)[:6, :6]
            pe, qe = self._ee()
            u = glue.osc_generate(o.get("qvel")[:6], target, J, M, o.get("qfrc_bias")[:6], pe, qe)
            o.step(np.concatenate([u, [0.6] * 3]))
            pin()
        return pos, quat, target

"""Gym-style batched Jaco environment: drop-in for the env path of the reference.

Mirrors /root/reference/env_script/env_mujoco.py `JacoMujocoEnv` (not a gym.Env subclass there either):
same attribute names (`observation_space`, `action_space`, `metadata`, `max_steps`, `task_max_steps`, `skip_frames`,
`state_shape`), same methods (`reset`, `step(action, weight=None, subgoal=None, id=None)`, `get_state_shape`,
`get_num_observation`, `get_num_action`, `get_action_bound`, `get_wb`, `seed`, `close`), same kwargs
(`task`, `robot_file`, `n_robots`, `seed`, `visualize`, ...), plus `num_envs`, `device`, `frame_skip`.
`num_envs == 1` returns the reference's unbatched types (float32[26] ndarray, float, bool, {0: 0}) so the evaluation
loop at main.py:254-260 runs unchanged; otherwise observations/rewards/dones are torch tensors on the GPU.
All arithmetic happens in libjaco_env.so (jaco_reset / jaco_step); there is no CPU path.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .physics import BatchedMujoco, JacoError

# picking / placing run end to end in the reference; the terminations of reaching / grasping / pickAndplace return 3-tuples there that
# env_mujoco.py:125 cannot unpack (env_mujoco_util.py:504-536,585-600): built here with the missing success flag added.
# carrying / pushing end every episode in their first step (`return True, 0, wb`, :549-550,583-584); releasing has a real rule (:551-566) and
# its own in-hand reset (:106-117,186-189).  Their terminations are 3-tuples in the reference as well.  'pushing' cannot even be reset
# there (_create_init_angle has no branch for it: UnboundLocalError); here it starts from the picking pose range.
TASK_IDS = {"picking": 0, "placing": 1, "reaching": 2, "grasping": 3, "pickAndplace": 4, "carrying": 5, "releasing": 6, "pushing": 7}


class Box:
    """Minimal stand-in for gym.spaces.Box (gym is not a dependency of this package)."""

    def __init__(self, low, high, dtype=np.float32):
        self.low, self.high = np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype)
        self.shape, self.dtype = self.low.shape, np.dtype(dtype)

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class JacoBatchedEnv:
    def __init__(self, num_envs=1, device=0, frame_skip=50, seed=0, **kwargs):
        self.task = kwargs.get("task", "picking")
        if self.task not in TASK_IDS:
            raise NotImplementedError("task %r: supported tasks are %s" % (self.task, sorted(TASK_IDS)))
        # observation / marker branch: the rule-based sub-goal (env_mujoco_util.py:240-254,607-609) is what main.py:44-45,183-184,
        # 223-224 always selects (the default here); rulebased_subgoal=False puts the reaching goal into obs[17:23] (:255-270).
        # subgoal_obs=True builds a 23-vector in the reference (:226-239) and fails its own shape assert (env_mujoco.py:154-157).
        if kwargs.get("subgoal_obs", False):
            raise NotImplementedError("subgoal_obs=True: the reference's own observation assert (23 != 26 entries) rejects this branch")
        self.rulebased_subgoal = bool(kwargs.get("rulebased_subgoal", True))
        # kwargs of the reference that need host-side Python per env and have no batched counterpart here: said out loud instead of ignored
        if kwargs.get("reward_method", None) is not None or kwargs.get("reward_module", None) is not None:
            raise NotImplementedError("reward_method / reward_module (env_mujoco_util.py:69-70,442-443: a Python callable evaluated per step) are not "
                                      "supported: the reward is computed inside the step kernel")
        self.n_robots = kwargs.get("n_robots", 1)
        if self.n_robots not in (1, 2):
            raise NotImplementedError("n_robots = %r: the reference ships models for one and two arms" % (self.n_robots,))
        # robot_file=None picks 'jaco2' + ['', '_dual', '_tri'][n_robots - 1] in the reference (env_mujoco_util.py:19-28), files its assets
        # directory does not hold; the shipped two-arm model is jaco2_dual_torque.xml
        robot_file = kwargs.get("robot_file", None) or ("jaco2_curtain_torque" if self.n_robots == 1 else "jaco2_dual_torque")
        self.num_envs = int(num_envs)
        self.sim = BatchedMujoco(self.num_envs, robot_file=robot_file, device=device, frame_skip=frame_skip,
                                 task=TASK_IDS[self.task], seed=int(seed))
        self.L, self.h, self.device = self.sim.L, self.sim.h, self.sim.device
        # Two arms: the sim tier (self.sim: send_forces / get_state / set_state on the two-arm build of the library) is what exists.  The
        # reference's env loop is single-robot as well -- _step_simulation stacks one gripper command onto the controller output
        # (env_mujoco_util.py:73-83: 15 values for the dual model's 18 controls) -- so reset() / step() have no working reference to follow.
        self.sim_tier_only = self.n_robots != 1
        if not self.rulebased_subgoal:
            self.sim.set_option("obs_mode", 1)
        # init_buffer (env_mujoco_util.py:46,208-212): recorded rows the reaching goal of every reset is drawn from (row[1:4] position,
        # row[4:7] orientation) instead of being sampled
        self.goal_buffer = None
        if kwargs.get("init_buffer", None) is not None:
            self.set_init_buffer(kwargs["init_buffer"])
        # auto_reset=True (batched rollouts): an env whose step ends its episode is reset inside that very jaco_step call -- sim.reset(),
        # the draws of _reset, sim.forward() -- by the wavefront that finished it; step() then returns the terminal step's reward and
        # done flag together with the FIRST observation of the new episode (and the task row, success flag included, is the new
        # episode's: read success as `reward > 100 and done`, main.py:262).  Tasks whose reset is more than draws + forward pass
        # (placing: 150-substep hold; grasping: pre-reach loops) keep the explicit reset(mask).
        want_auto = bool(kwargs.get("auto_reset", False))
        self.auto_reset = want_auto and self.task in ("picking", "reaching", "pickAndplace", "pushing")
        if want_auto and not self.auto_reset:
            import warnings
            warnings.warn("auto_reset is not available for task %r (its reset is more than draws + forward pass): call reset(done) after step()" % self.task)
        if self.auto_reset:
            self.sim.set_option("auto_reset", 1)
        self._subgoal = None
        # ---- RL setup (env_mujoco.py:15-93)
        self.current_steps = 0
        self.max_steps = 2500
        self.task_max_steps = 700 if self.task in ("picking", "placing") else (1200 if self.task == "pickAndplace" else 500)   # env_mujoco.py:18-23
        self.skip_frames = int(frame_skip)
        obs_max = np.hstack([[3], [1] * 25]).astype(np.float32).repeat(self.n_robots)   # (.repeat(n_robots), env_mujoco.py:57-58)
        self.observation_space = Box(-obs_max, obs_max, dtype=np.float32)
        try:   # env_mujoco.py:64-67: a state generator may dictate the state shape
            self.state_shape = kwargs["stateGen"].get_state_shape()
        except Exception:
            self.state_shape = self.observation_space.shape[0]
        self.pose_action_space_max = 1
        nact = 6 if self.task in ("reaching", "pushing") else 7   # env_mujoco.py:79-89
        self.act_max = np.ones(nact).repeat(self.n_robots)      # env_mujoco.py:86-88
        self.act_min = -np.ones(nact).repeat(self.n_robots)
        self.action_space = Box(self.act_min, self.act_max, dtype=np.float32)
        self.wb = 0
        self.metadata = None
        self._obs = torch.zeros(self.num_envs, 26, device=self.device)
        self._rew = torch.zeros(self.num_envs, device=self.device)
        self._done = torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device)
        self._noise = None
        self._amin = torch.tensor(self.act_min, dtype=torch.float32, device=self.device)
        self._amax = torch.tensor(self.act_max, dtype=torch.float32, device=self.device)

    # ---- helpers
    def _p(self, t):
        return ctypes.c_void_p(t.data_ptr())

    def _out(self, obs=None, rew=None, done=None):
        if self.num_envs != 1:
            return obs, rew, done
        o = obs[0].cpu().numpy().astype(np.float32) if obs is not None else None
        return o, (float(rew[0].item()) if rew is not None else None), (bool(done[0].item()) if done is not None else None)

    def set_noise(self, noise):
        """Inject the 12 uniform draws per env that the rule-based sub-goal consumes each step (tests); None = internal RNG."""
        if noise is None:
            self._noise = None
            self.sim._chk(self.L.jaco_set_noise(self.h, None))
        else:
            self._noise = noise.to(self.device, torch.float32).contiguous()
            assert self._noise.shape == (self.num_envs, 12)
            self.sim._chk(self.L.jaco_set_noise(self.h, self._p(self._noise)))

    # ---- reference surface
    def _env_tier(self):
        if self.sim_tier_only:
            raise NotImplementedError("n_robots = 2: the two-arm model is stepped at the sim-interface tier (env.sim.send_forces / get_state / "
                                      "set_state); the reference's env loop (env_mujoco_util.py:73-83) drives one arm")

    def reset(self, mask=None):
        self._env_tier()
        self.current_steps = 0
        m = self._mask(mask)
        self.sim._chk(self.L.jaco_reset(self.h, self._p(m) if m is not None else None, self._p(self._obs), self.sim._stream()))
        return self._out(self._obs)[0]

    def markers(self):
        """[num_envs, 2, 12] poses (position, rotation matrix) of the "hand" and "subgoal_reach" markers that step() moves
        (the reference's set_mocap_xyz / set_mocap_orientation, env_mujoco_util.py:613-615,644-646)."""
        t = torch.empty(self.num_envs, 2, 12, dtype=torch.float32, device=self.device)
        self.sim._chk(self.L.jaco_get_markers(self.h, self._p(t), self.sim._stream()))
        return t

    def set_markers(self, t):
        t = t.to(self.device, torch.float32).reshape(self.num_envs, 2, 12).contiguous()
        self.sim._chk(self.L.jaco_set_markers(self.h, self._p(t), self.sim._stream()))

    def _placing_hold(self, mask=None, nsub=150):
        """The held part of the placing reset on its own (env_mujoco_util.py:106-117); reset() runs it for task 'placing'."""
        m = self._mask(mask)
        self.sim._chk(self.L.jaco_placing_hold(self.h, self._p(m) if m is not None else None, int(nsub), self.sim._stream()))

    def _grasping_prereach(self, mask=None, max_substeps=4000):
        """The pre-reach loops of the grasping reset on their own (env_mujoco_util.py:123-170); reset() runs them for task 'grasping'."""
        m = self._mask(mask)
        self.sim._chk(self.L.jaco_grasping_prereach(self.h, self._p(m) if m is not None else None, int(max_substeps), self._p(self._obs), self.sim._stream()))
        return self._out(self._obs)[0]

    def _mask(self, mask):
        if mask is None:
            return None
        m = torch.as_tensor(mask, device=self.device)
        if m.numel() != self.num_envs:
            raise ValueError("mask must have one entry per env (%d), got shape %s" % (self.num_envs, tuple(m.shape)))
        return m.reshape(self.num_envs).to(torch.uint8).contiguous()

    def _action(self, action):
        """Validated [num_envs, nact] fp32 device tensor (the kernel reads nact floats per env unconditionally).  The np.clip of
        env_mujoco.py:117 happens inside the kernel (env_logic.h take_action): no elementwise launch per step for a tensor that is
        already on the device in the right layout."""
        a = torch.as_tensor(action, dtype=torch.float32, device=self.device)
        nact = self.action_space.shape[0]
        if a.numel() != self.num_envs * nact or (a.dim() > 1 and a.shape[-1] != nact):
            raise ValueError("action must have shape (%d, %d), got %s" % (self.num_envs, nact, tuple(a.shape)))
        return a.reshape(self.num_envs, nact).contiguous()

    def take_action(self, a, weight=None, subgoal=None, id=None):
        """env_mujoco.py:158-159 -> _take_action (env_mujoco_util.py:602-646): EE target, gripper command, marker poses; no physics."""
        a = self._action(a)
        self.sim._chk(self.L.jaco_take_action(self.h, self._p(a), self.sim._stream()))

    def terminal_inspection(self):
        """env_mujoco.py:144-150: current_steps += 1, then (done, additional_reward, wb, succ) of the task's termination rule."""
        bonus = torch.zeros(self.num_envs, device=self.device)
        self.sim._chk(self.L.jaco_terminal_inspection(self.h, self._p(self._done), self._p(bonus), self.sim._stream()))
        self.current_steps += 1
        ts = self.task_state()
        done, wb, succ = self._done.bool().clone(), ts[:, 30], ts[:, 29] > 0.5
        if self.num_envs == 1:
            return bool(done[0].item()), float(bonus[0].item()), float(wb[0].item()), int(succ[0].item())
        return done, bonus, wb, succ

    def _set_subgoal(self, subgoal):
        """rulebased_subgoal=False: `subgoal` is what the reference's step() gets from HPC.predict_subgoal -- a dict with the
        reaching primitive's 6 offsets under 'level1_reaching/level0' (env_mujoco_util.py:609) -- or a [num_envs, 6] tensor."""
        if subgoal is None or self.rulebased_subgoal:
            if self._subgoal is not None:
                self._subgoal = None
                self.sim._chk(self.L.jaco_set_subgoal(self.h, None))
            return
        if isinstance(subgoal, dict):
            subgoal = subgoal["level1_reaching/level0"]
        sg = torch.as_tensor(subgoal, dtype=torch.float32, device=self.device).reshape(self.num_envs, -1)[:, :6].contiguous()
        self._subgoal = sg
        self.sim._chk(self.L.jaco_set_subgoal(self.h, self._p(sg)))

    def step(self, action, weight=None, subgoal=None, id=None):
        """env_mujoco.py:116-139.  With num_envs > 1 the returned obs / reward / done tensors are the handle's own output buffers (done: a
        bool view of the kernel's byte flags): valid until the next step() / reset() call, `.clone()` what has to outlive it."""
        self._env_tier()
        a = self._action(action)
        self._set_subgoal(subgoal)
        self.sim._chk(self.L.jaco_step(self.h, self._p(a), self._p(self._obs), self._p(self._rew), self._p(self._done), self.sim._stream()))
        self.current_steps += 1
        obs, rew, done = self._out(self._obs, self._rew, self._done)
        if self.num_envs != 1:
            done = done.view(torch.bool)   # (the kernel writes 0 / 1: a reinterpreting view of the same buffer, no launch)
        return obs, rew, done, {0: 0}

    def make_observation(self):
        self.sim._chk(self.L.jaco_forward(self.h, self._p(self._obs), self.sim._stream()))
        return self._out(self._obs)[0]

    def task_state(self):
        n = self.L.jaco_task_row_floats()
        t = torch.empty(self.num_envs, n, device=self.device)
        self.sim._chk(self.L.jaco_get_task_state(self.h, self._p(t), self.sim._stream()))
        return t

    def set_task_state(self, t):
        t = t.to(self.device, torch.float32).contiguous()
        self.sim._chk(self.L.jaco_set_task_state(self.h, self._p(t), self.sim._stream()))

    def get_state_shape(self):
        return self.state_shape

    def get_num_observation(self):
        return self.observation_space.shape[0]

    def get_num_action(self):
        return self.action_space.shape[0]

    def get_action_bound(self):
        return self.pose_action_space_max

    def get_wb(self):
        wb = self.task_state()[:, 30]
        return float(wb[0].item()) if self.num_envs == 1 else wb

    def successes(self):
        return self.task_state()[:, 29] > 0.5

    def set_init_buffer(self, rows):
        """rows: array-like / tensor [n >= 2, >= 7 floats per row] (the reference indexes buffer[i][1:4] and [4:7]); None = sample the goal."""
        if rows is None:
            self.sim._chk(self.L.jaco_set_init_buffer(self.h, None, 0, 0, self.sim._stream()))
            self.goal_buffer = None
            return
        t = rows if torch.is_tensor(rows) else torch.as_tensor(np.asarray(rows, dtype=np.float32))
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        if t.ndim != 2:
            raise ValueError("init_buffer must be a 2-D array of rows")
        self.sim._chk(self.L.jaco_set_init_buffer(self.h, self._p(t), int(t.shape[0]), int(t.shape[1]), self.sim._stream()))
        self.goal_buffer = t

    def last_terminal(self):
        """(success [num_envs] bool, wb [num_envs] f32) of every env's most recent terminal step -- `succ` and `wb` of
        terminal_inspection (env_mujoco.py:125,144-150), latched by the step that ended the episode.  With auto_reset the task row
        (successes(), get_wb()) already belongs to the new episode when step() returns; this is what the reference's success-rate
        bookkeeping (env_mujoco.py:129-136) reads."""
        t = torch.empty(self.num_envs, 2, device=self.device)
        self.sim._chk(self.L.jaco_get_last_terminal(self.h, self._p(t), self.sim._stream()))
        return t[:, 0] > 0.5, t[:, 1]

    def terminal_observation(self):
        """[num_envs, 26]: the observation of every env's most recent terminal step under auto_reset (its obs row is the new episode's first)."""
        t = torch.empty(self.num_envs, 26, device=self.device)
        self.sim._chk(self.L.jaco_get_terminal_obs(self.h, self._p(t), self.sim._stream()))
        return t

    def seed(self, seed):
        pass  # the reference's seed() is a no-op too (env_mujoco.py:163-164); pass `seed=` to the constructor instead

    def set_capture_path(self, path):
        """JacoMujocoEnvUtil.set_capture_path (env_mujoco_util.py:683): where the reference's renderer writes frames.  Rendering is out
        of scope here (no viewer); the path is kept so callers that set it unconditionally keep working."""
        self.capture_path = path

    def close(self):
        self.sim.close()
        return None

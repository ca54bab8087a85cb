"""Learner-side adapter: the batched env behind the vectorised-env protocol that SB-style loops
drive (the reference trains through `model.learn(...)` on a single `JacoMujocoEnv`, main.py:172-178).

`JacoVecEnv.step(actions)` returns `(obs [B,26], rewards [B], dones [B], infos)`; envs that finished are reset inside the
call (masked `jaco_reset`; with `auto_reset=True` -- tasks picking / reaching / pickAndplace / pushing -- inside `jaco_step` itself, no
extra launches), their returned observation is the first one of the new episode and the last observation of
the finished episode is kept in `infos["terminal_observation"]` rows, as SB's VecEnv does.  Everything stays on the GPU
(`torch` tensors); `to_numpy=True` copies the four outputs to host arrays for learners that want numpy.

`dense_infos=True` (needs `auto_reset=True`): the step issues NO host synchronisation -- no `done.any()`, no boolean-mask gather (whose
result size the host would have to wait for).  `infos` then holds full-size device tensors, rows / entries valid where `done` is set:
`terminal_observation [B,26]`, `episode_return [B]`, `episode_length [B]`, `is_success [B]`, plus `quarantined` as a 0-dim device
tensor; a learner masks them with `done` on the device.  The default (sparse rows gathered with `[done]`) synchronises once per step.
"""
import torch

from .env import JacoBatchedEnv


class JacoVecEnv:
    def __init__(self, num_envs, to_numpy=False, dense_infos=False, **kwargs):
        assert int(num_envs) >= 2, "the vectorised adapter is for batches; JacoBatchedEnv(num_envs=1) is the reference's single-env surface"
        self.env = JacoBatchedEnv(num_envs=num_envs, **kwargs)
        self.num_envs = int(num_envs)
        self.observation_space, self.action_space = self.env.observation_space, self.env.action_space
        self.to_numpy = bool(to_numpy)
        self._actions = None
        self.episode_returns = torch.zeros(self.num_envs, device=self.env.device)
        self.episode_lengths = torch.zeros(self.num_envs, dtype=torch.int64, device=self.env.device)
        self.quarantined_total = 0
        self.dense_infos = bool(dense_infos)
        if self.dense_infos and not self.env.auto_reset:
            raise ValueError("dense_infos=True needs auto_reset=True on a task whose reset runs inside jaco_step (picking, reaching, pickAndplace, pushing)")
        self.quarantined_total_dev = torch.zeros((), dtype=torch.int64, device=self.env.device)

    def _out(self, *ts):
        return tuple(t.cpu().numpy() for t in ts) if self.to_numpy else ts

    def reset(self):
        self.episode_returns.zero_(); self.episode_lengths.zero_()
        obs = self.env.reset()
        return self._out(obs.clone())[0]

    def step_async(self, actions):
        self._actions = actions

    def step_wait(self):
        obs, rew, done, _ = self.env.step(self._actions)
        obs, rew, done = obs.clone(), rew.clone(), done.clone()
        self.episode_returns += rew; self.episode_lengths += 1
        infos = {"terminal_observation": None, "episode_return": None, "episode_length": None, "is_success": None, "quarantined": 0}
        if self.dense_infos:
            # no host round trip: everything is computed for every env and is meaningful where `done` is set
            q = (((self.env.sim.flags() & 8) != 0) & done).sum()
            self.quarantined_total_dev += q
            self.env.sim.clear_flags()   # (the quarantine bit is sticky; the envs it belonged to have been reset inside jaco_step)
            infos.update(terminal_observation=self.env.terminal_observation(), episode_return=self.episode_returns.clone(),
                         episode_length=self.episode_lengths.clone(), is_success=self.env.last_terminal()[0], quarantined=q)
            keep = ~done
            self.episode_returns *= keep; self.episode_lengths *= keep
        elif bool(done.any()) and self.env.auto_reset:
            # the env was reset inside jaco_step (option auto_reset): obs already holds the new episodes' first observations; what the
            # terminal step returned besides was latched by the kernel (jaco_get_terminal_obs / jaco_get_last_terminal)
            bad = (self.env.sim.flags() & 8) != 0
            if bool(bad.any()):
                infos["quarantined"] = int((bad & done).sum().item())
                self.quarantined_total += infos["quarantined"]
                self.env.sim.clear_flags()
            infos["terminal_observation"] = self.env.terminal_observation()[done]
            infos["episode_return"] = self.episode_returns[done].clone()
            infos["episode_length"] = self.episode_lengths[done].clone()
            infos["is_success"] = self.env.last_terminal()[0][done]
            self.episode_returns[done] = 0; self.episode_lengths[done] = 0
        elif bool(done.any()):
            # envs whose state went non-finite end their episode inside jaco_step (reward 0, JACO_FLAG_NAN) and are reset here
            # with the others; the count is reported and their flag cleared (failure row)
            bad = (self.env.sim.flags() & 8) != 0
            if bool(bad.any()):
                infos["quarantined"] = int((bad & done).sum().item())
                self.quarantined_total += infos["quarantined"]
                self.env.sim.clear_flags()
            infos["terminal_observation"] = obs[done].clone()
            infos["episode_return"] = self.episode_returns[done].clone()
            infos["episode_length"] = self.episode_lengths[done].clone()
            infos["is_success"] = self.env.task_state()[:, 29][done] > 0.5   # JT_SUCC of the terminal step
            new_obs = self.env.reset(done)
            obs[done] = new_obs[done]
            self.episode_returns[done] = 0; self.episode_lengths[done] = 0
        o, r, d = self._out(obs, rew, done)
        return o, r, d, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.env.close()

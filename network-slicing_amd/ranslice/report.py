"""VecReportWrapper: the reference's ReportWrapper interface (reference wrapper.py:24-138) over the batched
simulator -- float action simplex in, normalised observations out, per-replica history arrays.

  action  a in R^{S+1} (S slices + "unused")  ->  PRBs_i = floor(n_prbs * |a_i| / sum|a|)      (wrapper.py:77-82)
  obs                                          ->  clip(obs, -0.5, 1.5) - 0.5                   (wrapper.py:87-89)
  history: violation (int16), reward (float64), resources (int16) per step, saved as npz with the
  reference's keys (wrapper.py:120-123).
The mapping is elementwise host arithmetic on [N, S] arrays; the simulation itself stays on the GPU.
"""
import numpy as np


def simplex_to_prbs(action, n_prbs, n_slices):
    """wrapper.py:77-82 for a batch [N, S+1] (or [N, S]: already integer PRBs, passed through)"""
    action = np.asarray(action)
    if action.shape[-1] > n_slices:
        a = np.abs(action.astype(np.float64))
        t = a.sum(axis=-1, keepdims=True)
        t = np.where(t == 0, 1.0, t)
        return np.floor(n_prbs * a[..., :n_slices] / t).astype(np.int32)
    return action.astype(np.int32)


def normalise_obs(obs):
    """wrapper.py:87-89"""
    return np.clip(obs, -0.5, 1.5) - 0.5


class VecReportWrapper:
    def __init__(self, env, steps=2000, control_steps=500, env_id=1, path='./logs/', verbose=False):
        self.env = env
        self.n_envs, self.n_slices, self.n_prbs = env.n_envs, env.n_slices, env.n_prbs
        self.n_variables = env.n_variables
        self.steps, self.control_steps, self.env_id, self.path, self.verbose = steps, control_steps, env_id, path, verbose
        self.file_path = '{}history_{}.npz'.format(path, env_id)
        self.step_counter = 0
        self.reset_history()

    def reset_history(self):
        self.violation_history = np.zeros((self.n_envs, self.steps), dtype=np.int16)
        self.reward_history = np.zeros((self.n_envs, self.steps), dtype=np.float64)
        self.action_history = np.zeros((self.n_envs, self.steps), dtype=np.int16)

    def reset(self, seeds=None):
        self.step_counter = 0
        self.obs = self.env.reset(seeds=seeds)
        return self.obs

    def step(self, action):
        prbs = simplex_to_prbs(action, self.n_prbs, self.n_slices)
        obs, reward, done, info = self.env.step(prbs)
        self.obs = normalise_obs(obs)
        if self.step_counter < self.steps:
            self.violation_history[:, self.step_counter] = info['total_violations']
            self.reward_history[:, self.step_counter] = reward
            self.action_history[:, self.step_counter] = prbs.sum(axis=1)
        self.step_counter += 1
        if self.step_counter % self.control_steps == 0:
            self.save_results()
        return self.obs, reward, done, {0: 0}

    def save_results(self):
        import os
        os.makedirs(self.path, exist_ok=True)
        np.savez(self.file_path, violation=self.violation_history, reward=self.reward_history,
                 resources=self.action_history)

    def set_evaluation(self, eval_steps, new_path=None, change_name=False):
        """wrapper.py:125-134"""
        self.step_counter = self.steps
        self.steps += eval_steps
        pad = [(0, 0), (0, eval_steps)]
        self.violation_history = np.pad(self.violation_history, pad)
        self.reward_history = np.pad(self.reward_history, pad)
        self.action_history = np.pad(self.action_history, pad)
        if new_path:
            self.path = new_path
        if change_name:
            self.file_path = '{}evaluation_{}.npz'.format(self.path, self.env_id)

"""Vectorised stepping with auto-reset + episode statistics + GAE.  ORACLE — test infra only.

Follows:
* envs/env_wrappers/vectorized_env/dummy_vec_env.py:29-48 (step_wait: auto-reset on done;
  the returned obs is the post-reset observation, ``terminal_observation`` /
  ``terminal_info`` carry the pre-reset values; reset()).
* envs/env_wrappers/record_episode_statistics.py:139-166 (VecRecordEpisodeStatistics
  .step_wait: running return/length, tracked info keys read from ``terminal_info`` on done).
* controllers/ppo/ppo_utils.py:374-400 (compute_returns_and_advantages).
"""
from collections import deque

import numpy as np


class OracleVecEnv:
    """DummyVecEnv semantics over a batched oracle env; info is columnar (arrays, not dicts)."""

    def __init__(self, env):
        self.env = env
        self.num_envs = env.num_envs

    def reset(self):
        return self.env.reset()

    def step(self, actions):
        obs, rew, done, info = self.env.step(actions)
        out = {
            'done': done.copy(),
            'truncated': (info['TimeLimit.truncated'] & info['time_limit_reached']),
            'time_limit_reached': info['time_limit_reached'].copy(),     # where upstream's info HAS the TimeLimit.truncated key
            'constraint_violation': info['constraint_violation'].copy(),
            'mse': info['mse'].copy(),
            'terminal_observation': obs.copy(),     # meaningful where done
            'current_step': info['current_step'].copy(),
        }
        for k in ('constraint_values', 'out_of_bounds', 'goal_reached'):
            if k in info:
                out[k] = info[k].copy()
        idx = np.nonzero(done)[0]
        if len(idx):
            r_obs, r_info = self.env.reset(idx)
            obs = obs.copy()
            obs[idx] = r_obs
            out['reset_info'] = r_info
        return obs, rew, done, out


class OracleEpisodeStats:
    """VecRecordEpisodeStatistics semantics, columnar."""

    def __init__(self, num_envs, deque_size=None):
        self.num_envs = num_envs
        self.episode_return = np.zeros(num_envs)
        self.episode_length = np.zeros(num_envs)
        self.return_queue = deque(maxlen=deque_size)
        self.length_queue = deque(maxlen=deque_size)
        self.deque_size = deque_size
        self.episode_stats, self.accumulated_stats, self.queued_stats = {}, {}, {}

    def add_tracker(self, name, init_value, mode='accumulate'):
        self.episode_stats[name] = np.full(self.num_envs, float(init_value))
        if mode == 'accumulate':
            self.accumulated_stats[name] = init_value
        elif mode == 'queue':
            self.queued_stats[name] = deque(maxlen=self.deque_size)
        else:
            raise Exception('Tracker mode not implemented.')

    def update(self, reward, done, info):
        """``info[key]`` holds the step's (pre-reset) value for every env."""
        self.episode_return += reward
        self.episode_length += 1
        for key in self.episode_stats:
            if key in info:
                self.episode_stats[key] += info[key]
        finished = []
        for i in np.nonzero(done)[0]:              # env order, like the reference's loop
            ep = {'r': self.episode_return[i], 'l': self.episode_length[i]}
            self.return_queue.append(self.episode_return[i])
            self.length_queue.append(self.episode_length[i])
            self.episode_return[i] = 0
            self.episode_length[i] = 0
            for key in self.episode_stats:
                ep[key] = self.episode_stats[key][i]
                if key in self.accumulated_stats:
                    self.accumulated_stats[key] += self.episode_stats[key][i]
                if key in self.queued_stats:
                    self.queued_stats[key].append(self.episode_stats[key][i])
                self.episode_stats[key][i] = 0
            finished.append((i, ep))
        return finished


def compute_returns_and_advantages(rews, vals, masks, terminal_vals=0, last_val=0, gamma=0.99,
                                   use_gae=False, gae_lambda=0.95):
    """ppo_utils.py:374-400.  Shapes (T, N, 1); ``rews`` is NOT mutated here (the reference adds
    gamma * terminal_vals in place, :389 — callers compare against ``rews + gamma*terminal_vals``)."""
    rews = np.asarray(rews, dtype=np.float64) + gamma * np.asarray(terminal_vals, dtype=np.float64)
    vals = np.asarray(vals, dtype=np.float64)
    masks = np.asarray(masks, dtype=np.float64)
    T, N = rews.shape[:2]
    rets, advs = np.zeros((T, N, 1)), np.zeros((T, N, 1))
    ret, adv = np.asarray(last_val, dtype=np.float64), np.zeros((N, 1))
    vals = np.concatenate([vals, np.asarray(last_val, dtype=np.float64)[np.newaxis, ...]], 0)
    for i in reversed(range(T)):
        ret = rews[i] + gamma * masks[i] * ret
        if not use_gae:
            adv = ret - vals[i]
        else:
            td_error = rews[i] + gamma * masks[i] * vals[i + 1] - vals[i]
            adv = adv * gae_lambda * gamma * masks[i] + td_error
        rets[i] = ret
        advs[i] = adv
    return rets, advs

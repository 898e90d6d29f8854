"""Columnar `VecRecordEpisodeStatistics`.

Same interface as the reference wrapper
(/root/reference/safe_control_gym/envs/env_wrappers/record_episode_statistics.py:92-166): `add_tracker(name, init,
mode)`, `return_queue`, `length_queue`, `episode_stats`, `accumulated_stats`, `queued_stats`, `reset()`, `step_wait()`.
The per-env Python loop of the reference (`for i, (r, d) in enumerate(zip(reward, done))`, :141-165) does not exist here:
running returns / lengths / constraint violations / mse are accumulated inside the step kernel (scg_step_out.d_ep_* and
d_fin_*), and this wrapper only moves the finished episodes of a step (usually a handful) to the host deques.
"""
from collections import deque

import numpy as np
import torch

TRACKED_IN_KERNEL = {'constraint_violation': 'fin_violation', 'mse': 'fin_mse'}


class VecRecordEpisodeStatistics:
    def __init__(self, venv, deque_size=None, **kwargs):
        self.venv = venv
        self.num_envs = venv.num_envs
        self.observation_space, self.action_space = venv.observation_space, venv.action_space
        self.deque_size = deque_size
        self.return_queue = deque(maxlen=deque_size)
        self.length_queue = deque(maxlen=deque_size)
        self.episode_stats, self.accumulated_stats, self.queued_stats = {}, {}, {}

    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError(name)
        return getattr(self.venv, name)

    @property
    def episode_return(self):
        return self.venv.ep_return.cpu().numpy().astype(np.float64)

    @property
    def episode_length(self):
        return self.venv.ep_length.cpu().numpy().astype(np.float64)

    def add_tracker(self, name, init_value, mode='accumulate'):
        if name not in TRACKED_IN_KERNEL:
            raise NotImplementedError(f"tracker '{name}': the step kernel accumulates {sorted(TRACKED_IN_KERNEL)}")
        self.episode_stats[name] = init_value
        if mode == 'accumulate':
            self.accumulated_stats[name] = init_value
        elif mode == 'queue':
            self.queued_stats[name] = deque(maxlen=self.deque_size)
        else:
            raise Exception('Tracker mode not implemented.')

    def reset(self, **kwargs):
        return self.venv.reset(**kwargs)

    def step_async(self, actions):
        self.venv.step_async(actions)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def step_wait(self):
        obs, reward, done, info = self.venv.step_wait()
        out = self.venv.out
        d = out.done.bool()
        if bool(d.any()):
            idx = d.nonzero(as_tuple=False)[:, 0]
            rets = out.fin_return[idx].cpu().numpy().astype(np.float64)
            lens = out.fin_length[idx].cpu().numpy().astype(np.float64)
            self.return_queue.extend(rets.tolist())
            self.length_queue.extend(lens.tolist())
            for key, field in TRACKED_IN_KERNEL.items():
                if key in self.episode_stats:
                    vals = getattr(out, field)[idx].cpu().numpy().astype(np.float64)
                    if key in self.accumulated_stats:
                        self.accumulated_stats[key] += float(vals.sum())
                    if key in self.queued_stats:
                        self.queued_stats[key].extend(vals.tolist())
        return obs, reward, done, info

    def close(self):
        self.venv.close()


def make_vec_envs(env_id, task_config, batch_size=1, n_processes=1, seed=None, **kwargs):
    """Counterpart of vectorized_env/__init__.py:42-66: ONE HipVecEnv holds the whole batch (n_processes is
    accepted for call compatibility and ignored — there are no worker processes)."""
    from safe_control_gym_amd.vec_env import HipVecEnv
    return HipVecEnv(env_id, batch_size, seed=0 if seed is None else seed, **{**task_config, **kwargs})

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


def resolve_env_func(env_func):
    """(env_id, task_config) behind the `env_func` every reference controller receives —
    `partial(make, task, output_dir=..., **task_config)` (examples/rl/train_rl_model.py / rl_experiment.py:
    train_rl_controller.py:32-36), with `make` the reference's utils.registration.make or this package's.  Also accepted: a
    partial of an env class named CartPole / Quadrotor, and objects exposing `.env_id` / `.task_config`."""
    import functools
    if hasattr(env_func, 'env_id') and hasattr(env_func, 'task_config'):
        return env_func.env_id, dict(env_func.task_config)
    if isinstance(env_func, functools.partial):
        kw = dict(env_func.keywords or {})
        if env_func.args and isinstance(env_func.args[0], str):
            return env_func.args[0], kw
        name = getattr(env_func.func, '__name__', '').lower()
        if name in ('cartpole', 'quadrotor'):
            return name, kw
    raise TypeError('make_vec_envs needs env_func = functools.partial(make, <env id>, **task_config) (what the reference\'s '
                    'training scripts build), a partial of CartPole / Quadrotor, or an object with .env_id / .task_config')


def make_vec_envs(env_func, env_configs=None, batch_size=1, n_processes=1, seed=None, **kwargs):
    """Drop-in for envs/env_wrappers/vectorized_env/__init__.py:42-66 — same signature, same call sites
    (`make_vec_envs(env_func, None, rollout_batch_size, num_workers, seed)`, controllers/ppo/ppo.py:48, sac/sac.py:50):
    returns ONE HipVecEnv holding the whole batch instead of DummyVecEnv / SubprocVecEnv over `batch_size` Python envs.

    * env_func: see resolve_env_func; keys the simulator has no use for (output_dir, gui, verbose, ...) are accepted and
      ignored exactly like the env constructors' **kwargs upstream.
    * env_configs: upstream's per-env "non-shareable" kwargs (env k = env_func(**env_configs[k])).  None / all-equal entries are
      merged into the task config of ONE HipVecEnv; entries that differ are grouped by identical config into one HipVecEnv per
      group behind vec_env.GroupedVecEnv (each env keeps its own config across auto-resets, as upstream's env objects do).
    * n_processes: accepted for call compatibility; there are no worker processes (the batch is one kernel launch).
    * seed: upstream seeds env `rank` with seed + rank (:28-38); here it is the Philox key, env ids are the counter.
    Historic form `make_vec_envs(env_id: str, task_config: dict, batch_size, ...)` is still accepted."""
    from safe_control_gym_amd.vec_env import HipVecEnv
    if isinstance(env_func, str):
        env_id, cfg = env_func, dict(env_configs or {})
    else:
        env_id, cfg = resolve_env_func(env_func)
        if env_configs is not None:
            cfgs = [dict(c) for c in env_configs]
            if cfgs and len(cfgs) != batch_size:
                raise ValueError(f'env_configs has {len(cfgs)} entries for batch_size {batch_size}')
            if any(not _same_config(c, cfgs[0]) for c in cfgs):
                return _grouped(env_id, cfg, cfgs, kwargs, 0 if seed is None else seed)
            if cfgs:
                cfg.update(cfgs[0])
    cfg.update(kwargs)
    cfg.pop('seed', None)
    return HipVecEnv(env_id, batch_size, seed=0 if seed is None else seed, **cfg)


def _same_config(a, b):
    import numpy as np
    if a.keys() != b.keys():
        return False
    for k in a:
        x, y = a[k], b[k]
        if isinstance(x, np.ndarray) or isinstance(y, np.ndarray):
            if not np.array_equal(np.asarray(x), np.asarray(y)):
                return False
        elif x != y:
            return False
    return True


def _grouped(env_id, base_cfg, cfgs, kwargs, seed):
    """One HipVecEnv per distinct per-env config (first-occurrence order).  The groups get DISJOINT ranges of global env ids
    (env_id_offset = the number of envs in the groups before it; the kernel's Philox counter word 0 is offset + row), so no two
    envs of the batch share a random stream whatever the interleaving; when the groups are contiguous runs of the batch that is
    exactly "env k draws the stream of global env id k", i.e. the same numbers as a homogeneous batch."""
    from safe_control_gym_amd.vec_env import GroupedVecEnv, HipVecEnv
    reps, members = [], []
    for k, c in enumerate(cfgs):
        for g, r in enumerate(reps):
            if _same_config(c, r):
                members[g].append(k)
                break
        else:
            reps.append(c)
            members.append([k])
    groups, first_id = [], 0
    for r, idx in zip(reps, members):
        c = dict(base_cfg)
        c.update(r)
        c.update(kwargs)
        c.pop('seed', None)
        groups.append((HipVecEnv(env_id, len(idx), seed=seed, env_id_offset=first_id, **c), idx))
        first_id += len(idx)
    return GroupedVecEnv(groups)

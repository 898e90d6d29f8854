#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE'S OWN PYTHON.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference is pure Python but imports gymnasium / casadi / pybullet, which are absent
here; tests/golden/ref_stubs.py provides stand-ins (see its docstring — the only restated
piece is Bullet's rigid-body integrator, taken from oracle/bullet.py).  Everything else
executed here — BenchmarkEnv, CartPole, Quadrotor, constraints, disturbances, DummyVecEnv
auto-reset, VecRecordEpisodeStatistics, compute_returns_and_advantages, trajectory
generation — is the reference's real code, so these fixtures pin the oracle (and through
it the HIP kernels) to the reference for all of SURVEY.md §8a except the internals of
``p.stepSimulation``.

Outputs (all small):
    tests/golden/rollout_<case>.npz   per-step obs / rew / done / info columns + config JSON
    tests/golden/gae.npz              compute_returns_and_advantages known answers
    tests/golden/policies.npz         shipped PPO actor/critic weights (closed-loop controllers)
    tests/golden/xgoal_kat.npz        'obs' batches stored in the shipped checkpoints (X_GOAL KAT)
"""
import copy
import json
import os
import sys

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden import ref_stubs  # noqa: E402

ref_stubs.install()
REF = ref_stubs.REFERENCE_ROOT

import torch  # noqa: E402

from safe_control_gym.controllers.ppo.ppo_utils import compute_returns_and_advantages  # noqa: E402
from safe_control_gym.envs.env_wrappers.record_episode_statistics import VecRecordEpisodeStatistics  # noqa: E402
from safe_control_gym.envs.env_wrappers.vectorized_env import make_vec_envs  # noqa: E402
from safe_control_gym.envs.gym_control.cartpole import CartPole  # noqa: E402
from safe_control_gym.envs.gym_pybullet_drones.quadrotor import Quadrotor  # noqa: E402
from safe_control_gym.utils.utils import merge_dict  # noqa: E402

ENV_CLS = {'cartpole': CartPole, 'quadrotor': Quadrotor}
DEFAULT_YAML = {'cartpole': 'safe_control_gym/envs/gym_control/cartpole.yaml',
                'quadrotor': 'safe_control_gym/envs/gym_pybullet_drones/quadrotor.yaml'}


def load_task_config(task, override_rel=None, extra=None):
    """Default YAML next to the env class merged with an example override (utils/configuration.py:53-92)."""
    with open(os.path.join(REF, DEFAULT_YAML[task])) as f:
        cfg = yaml.safe_load(f)
    if override_rel is not None:
        with open(os.path.join(REF, override_rel)) as f:
            merge_dict(cfg, yaml.safe_load(f)['task_config'])
    if extra:
        merge_dict(cfg, copy.deepcopy(extra))
    return cfg


def _policy_mean(weights, tag, obs, activation):
    """Deterministic action of a shipped PPO actor (ppo_utils.py:233-238: act = dist.mode()), float64."""
    h = np.asarray(obs, dtype=np.float64)
    act = {'tanh': np.tanh, 'leaky_relu': lambda v: np.where(v > 0, v, 0.01 * v)}[activation]
    for i in range(3):
        W = weights[f'{tag}/actor.pi_net.fcs.{i}.weight'].astype(np.float64)
        b = weights[f'{tag}/actor.pi_net.fcs.{i}.bias'].astype(np.float64)
        h = h @ W.T + b
        if i < 2:
            h = act(h)
    return h


def rollout_case(name, task, cfg, n_envs, n_steps, seed, act_scale=1.0, act_seed=123, adversary=False,
                 policy=None):
    cfg_run = copy.deepcopy(cfg)
    cfg_run.pop('seed', None)
    cfg_run['output_dir'] = '/tmp'

    def env_func(**kw):
        return ENV_CLS[task](**{**cfg_run, **kw})

    venv = make_vec_envs(env_func, None, n_envs, 1, seed)
    venv = VecRecordEpisodeStatistics(venv, deque_size=1000)
    venv.add_tracker('constraint_violation', 0)
    venv.add_tracker('constraint_violation', 0, mode='queue')
    venv.add_tracker('mse', 0, mode='queue')
    envs = venv.venv.envs
    e0 = envs[0]
    obs0, info0 = venv.reset()
    nx, nu, no = e0.state_dim, e0.action_dim, obs0.shape[1]
    nc = e0.num_constraints
    arng = np.random.default_rng(act_seed)
    rec = {k: [] for k in ('actions', 'obs', 'rew', 'done', 'truncated', 'violation', 'mse', 'oob',
                           'c_values', 'terminal_obs', 'state', 'ep_return', 'ep_length', 'adv_actions')}
    state0 = np.stack([e.state for e in envs])
    reset_c = None
    if 'constraint_values' in info0['n'][0]:
        reset_c = np.stack([inf['constraint_values'] for inf in info0['n']])
    cur_obs = obs0
    for t in range(n_steps):
        if policy is not None:
            act = _policy_mean(policy[0], policy[1], cur_obs, policy[2]) + act_scale * arng.standard_normal((n_envs, nu))
        else:
            act = act_scale * arng.standard_normal((n_envs, nu))
            if t % 7 == 3:
                act *= 4.0                  # exercise the physical clipping / pwm saturation
        if adversary:
            adv_dim = e0.adversary_action_space.shape[0]
            adv = arng.uniform(-1.5, 1.5, size=(n_envs, adv_dim))
            for e, a in zip(envs, adv):
                e.set_adversary_control(a)
            rec['adv_actions'].append(adv)
        obs, rew, done, info = venv.step(act)
        cur_obs = obs
        rec['actions'].append(act)
        rec['obs'].append(obs)
        rec['rew'].append(np.asarray(rew, dtype=float))
        rec['done'].append(np.asarray(done, dtype=bool))
        tr, vi, ms, ob, cv, to, er, el = [], [], [], [], [], [], [], []
        for i, inf in enumerate(info['n']):
            step_inf = inf['terminal_info'] if done[i] else inf
            tr.append(bool(step_inf.get('TimeLimit.truncated', False)))
            vi.append(int(step_inf['constraint_violation']))
            ms.append(float(step_inf['mse']))
            ob.append(bool(step_inf.get('out_of_bounds', False)))
            cv.append(np.asarray(step_inf['constraint_values'], dtype=float) if nc else np.zeros(0))
            to.append(np.asarray(inf['terminal_observation'], dtype=float) if done[i] else np.full(no, np.nan))
            er.append(float(inf['episode']['r']) if done[i] else np.nan)
            el.append(float(inf['episode']['l']) if done[i] else np.nan)
        rec['truncated'].append(tr)
        rec['violation'].append(vi)
        rec['mse'].append(ms)
        rec['oob'].append(ob)
        rec['c_values'].append(np.stack(cv))
        rec['terminal_obs'].append(np.stack(to))
        rec['ep_return'].append(er)
        rec['ep_length'].append(el)
        rec['state'].append(np.stack([e.state for e in envs]))
    out = {k: np.asarray(v) for k, v in rec.items() if len(v)}
    out['obs0'] = obs0
    out['state0'] = state0
    if reset_c is not None:
        out['reset_c_values'] = reset_c
    out['x_goal'] = np.asarray(e0.X_GOAL, dtype=float)
    out['u_goal'] = np.asarray(e0.U_GOAL, dtype=float)
    out['state_space_low'] = e0.state_space.low
    out['state_space_high'] = e0.state_space.high
    out['observation_space_low'] = e0.observation_space.low
    out['action_space_low'] = e0.action_space.low
    out['action_space_high'] = e0.action_space.high
    out['physical_action_low'] = np.asarray(e0.physical_action_bounds[0], dtype=float)
    out['physical_action_high'] = np.asarray(e0.physical_action_bounds[1], dtype=float)
    out['return_queue'] = np.asarray(venv.return_queue, dtype=float)
    out['length_queue'] = np.asarray(venv.length_queue, dtype=float)
    out['accumulated_violation'] = np.asarray(venv.accumulated_stats['constraint_violation'], dtype=float)
    out['queued_mse'] = np.asarray(venv.queued_stats['mse'], dtype=float)
    meta = {'task': task, 'config': cfg, 'n_envs': n_envs, 'n_steps': n_steps, 'seed': seed,
            'adversary': adversary}
    out['meta_json'] = np.array(json.dumps(meta))
    venv.close()
    path = os.path.join(HERE, f'rollout_{name}.npz')
    np.savez_compressed(path, **out)
    print(f'{name:28s} steps={n_steps} envs={n_envs} dones={int(out["done"].sum())} '
          f'trunc={int(out["truncated"].sum())} viol={int(out["violation"].sum())} -> {os.path.basename(path)}')


def gae_cases():
    rng = np.random.default_rng(7)
    out = {}
    for k, (T, N, use_gae) in enumerate([(12, 5, True), (12, 5, False), (40, 3, True)]):
        rews = rng.standard_normal((T, N, 1)).astype(np.float32)
        vals = rng.standard_normal((T, N, 1)).astype(np.float32)
        masks = (rng.uniform(size=(T, N, 1)) > 0.2).astype(np.float32)
        term = (rng.standard_normal((T, N, 1)) * (masks == 0) * (rng.uniform(size=(T, N, 1)) > 0.5)).astype(np.float32)
        last = rng.standard_normal((N, 1)).astype(np.float32)
        r_in = rews.copy()
        rets, advs = compute_returns_and_advantages(r_in, vals, masks, term, last, gamma=0.99,
                                                    use_gae=use_gae, gae_lambda=0.95)
        out.update({f'rews{k}': rews, f'vals{k}': vals, f'masks{k}': masks, f'term{k}': term,
                    f'last{k}': last, f'rets{k}': rets, f'advs{k}': advs, f'use_gae{k}': np.array(use_gae)})
    np.savez_compressed(os.path.join(HERE, 'gae.npz'), **out)
    print('gae.npz written')


def shipped_models():
    pol, kat = {}, {}
    base = os.path.join(REF, 'examples/rl/models/ppo')
    for tag in ('cartpole_stab', 'cartpole_track', 'quadrotor_2D_track', 'quadrotor_2D_stab',
                'quadrotor_3D_track'):
        d = torch.load(os.path.join(base, f'ppo_model_{tag}.pt'), weights_only=False, map_location='cpu')
        for k, v in d['agent']['ac'].items():
            pol[f'{tag}/{k}'] = v.numpy()
        kat[f'{tag}/obs'] = np.asarray(d['obs'], dtype=float)
        kat[f'{tag}/total_steps'] = np.array(d['total_steps'])
    # one shipped SAC actor (layout / squashing check of sac.MLPActorCritic; critics omitted to keep the fixture small)
    d = torch.load(os.path.join(REF, 'examples/rl/models/sac/sac_model_cartpole_stab.pt'), weights_only=False, map_location='cpu')
    sac = {k: v.numpy() for k, v in d['agent']['ac'].items() if k.startswith('actor.')}
    sac['log_alpha'] = np.asarray(d['agent']['log_alpha'].detach().numpy() if hasattr(d['agent']['log_alpha'], 'detach') else d['agent']['log_alpha'])
    sac['obs'] = np.asarray(d['obs'], dtype=float)
    np.savez_compressed(os.path.join(HERE, 'sac_actor_cartpole_stab.npz'), **sac)
    np.savez_compressed(os.path.join(HERE, 'policies.npz'), **pol)
    np.savez_compressed(os.path.join(HERE, 'xgoal_kat.npz'), **kat)
    print('policies.npz, xgoal_kat.npz written')


def main():
    rl = 'examples/rl/config_overrides'
    # --- the three shipped RL task configs (BASELINE configs #2, #3, #5 envs) ---
    rollout_case('cartpole_stab', 'cartpole', load_task_config('cartpole', f'{rl}/cartpole/cartpole_stab.yaml'),
                 n_envs=3, n_steps=320, seed=42, act_scale=0.6)
    rollout_case('cartpole_track', 'cartpole', load_task_config('cartpole', f'{rl}/cartpole/cartpole_track.yaml'),
                 n_envs=3, n_steps=200, seed=5, act_scale=0.3)
    rollout_case('quadrotor_2D_track', 'quadrotor',
                 load_task_config('quadrotor', f'{rl}/quadrotor_2D/quadrotor_2D_track.yaml'),
                 n_envs=4, n_steps=520, seed=1337, act_scale=0.5)
    rollout_case('quadrotor_2D_stab', 'quadrotor',
                 load_task_config('quadrotor', f'{rl}/quadrotor_2D/quadrotor_2D_stab.yaml'),
                 n_envs=3, n_steps=300, seed=11, act_scale=0.5)
    rollout_case('quadrotor_3D_track', 'quadrotor',
                 load_task_config('quadrotor', f'{rl}/quadrotor_3D/quadrotor_3D_track.yaml'),
                 n_envs=3, n_steps=520, seed=1337, act_scale=0.4)
    # --- feature coverage: disturbances, randomised inertia, penalties, quadratic cost, adversary ---
    cp_dist = {
        'randomized_inertial_prop': True,
        'inertial_prop_randomization_info': {
            'pole_length': {'distrib': 'choice', 'args': [[0.1, 0.3, 0.5]]},
            'cart_mass': {'distrib': 'uniform', 'low': -0.2, 'high': 0.5},
            'pole_mass': {'distrib': 'normal', 'loc': 0.05, 'scale': 0.005}},
        'disturbances': {
            'observation': [{'disturbance_func': 'white_noise', 'std': [0.01, 0.02, 0.005, 0.03]}],
            'action': [{'disturbance_func': 'white_noise', 'std': 0.5},
                       {'disturbance_func': 'impulse', 'magnitude': 3, 'duration': 6, 'decay_rate': 0.7}],
            'dynamics': [{'disturbance_func': 'uniform', 'low': [-0.2, -0.1], 'high': [0.2, 0.1]},
                         {'disturbance_func': 'step', 'magnitude': 0.3, 'mask': [1, 0]},
                         {'disturbance_func': 'periodic', 'scale': 0.2, 'frequency': 2.0}]},
        'done_on_violation': True, 'use_constraint_penalty': True, 'constraint_penalty': 1.5,
        'obs_wrap_angle': True,
        'constraints': [{'constraint_form': 'default_constraint', 'constrained_variable': 'state',
                         'upper_bounds': [2, 4, 0.5, 4], 'lower_bounds': [-2, -4, -0.5, -4]},
                        {'constraint_form': 'abs_bound', 'constrained_variable': 'state',
                         'bound': 0.45, 'active_dims': 2, 'strict': True},
                        {'constraint_form': 'default_constraint', 'constrained_variable': 'input'}]}
    rollout_case('cartpole_disturbed', 'cartpole',
                 load_task_config('cartpole', f'{rl}/cartpole/cartpole_stab.yaml', cp_dist),
                 n_envs=3, n_steps=260, seed=3, act_scale=0.5)
    q2_quad = {
        'cost': 'quadratic', 'task': 'stabilization', 'normalized_rl_action_space': False,
        'task_info': {'stabilization_goal': [0.2, 1.1], 'stabilization_goal_tolerance': 0.3},
        'rew_state_weight': [2, 0.1, 2, 0.1, 0.5, 0.05], 'rew_act_weight': [0.3, 0.2],
        'obs_goal_horizon': 0, 'randomized_inertial_prop': True,
        'disturbances': {'dynamics': [{'disturbance_func': 'white_noise', 'std': [0.02, 0.01]}],
                         'action': [{'disturbance_func': 'uniform', 'low': -0.01, 'high': 0.01}],
                         'observation': [{'disturbance_func': 'white_noise', 'std': 0.01,
                                          'mask': [1, 0, 1, 0, 1, 0]}]},
        'constraints': [{'constraint_form': 'bounded_constraint', 'constrained_variable': 'state',
                         'active_dims': [0, 2], 'lower_bounds': [-1.0, 0.2], 'upper_bounds': [1.0, 1.8]},
                        {'constraint_form': 'linear_constraint', 'constrained_variable': 'input',
                         'A': [[1.0, 1.0], [-1.0, 0.5]], 'b': [0.5, 0.1]},
                        {'constraint_form': 'quadratic_constraint', 'constrained_variable': 'state',
                         'P': [[1.0, 0.1], [0.1, 2.0]], 'b': 3.0, 'active_dims': [1, 3]}],
        'done_on_violation': False, 'use_constraint_penalty': False}
    cfg = load_task_config('quadrotor', f'{rl}/quadrotor_2D/quadrotor_2D_track.yaml', q2_quad)
    rollout_case('quadrotor_2D_quadratic', 'quadrotor', cfg, n_envs=3, n_steps=300, seed=9, act_scale=0.03)
    q3_hard = {
        'randomized_inertial_prop': True,
        'disturbances': {'dynamics': [{'disturbance_func': 'white_noise', 'std': [0.01, 0.01, 0.02]}],
                         'action': [{'disturbance_func': 'white_noise', 'std': 0.005}]},
        'use_constraint_penalty': True, 'constraint_penalty': 0.2, 'done_on_violation': False}
    rollout_case('quadrotor_3D_disturbed', 'quadrotor',
                 load_task_config('quadrotor', f'{rl}/quadrotor_3D/quadrotor_3D_track.yaml', q3_hard),
                 n_envs=3, n_steps=300, seed=21, act_scale=0.4)
    q1 = {'quad_type': 1, 'init_state': {'init_x': 1.0, 'init_x_dot': 0.0},
          'rew_state_weight': [1, 0.1], 'rew_act_weight': 0.01, 'constraints': None,
          'task_info': {'trajectory_type': 'circle', 'num_cycles': 1, 'trajectory_plane': 'zx',
                        'trajectory_position_offset': [1.0, 0], 'trajectory_scale': 0.5}}
    rollout_case('quadrotor_1D_track', 'quadrotor',
                 load_task_config('quadrotor', f'{rl}/quadrotor_2D/quadrotor_2D_track.yaml', q1),
                 n_envs=2, n_steps=300, seed=4, act_scale=0.5)
    adv = {'adversary_disturbance': 'dynamics', 'adversary_disturbance_scale': 0.05,
           'adversary_disturbance_offset': 0.01, 'done_on_out_of_bound': False, 'episode_len_sec': 2,
           'task_info': {'trajectory_type': 'square', 'num_cycles': 1, 'trajectory_plane': 'xz',
                         'trajectory_position_offset': [0, 1], 'trajectory_scale': 0.5}}
    rollout_case('quadrotor_2D_adversary', 'quadrotor',
                 load_task_config('quadrotor', f'{rl}/quadrotor_2D/quadrotor_2D_track.yaml', adv),
                 n_envs=2, n_steps=220, seed=2, act_scale=0.5, adversary=True)
    # --- goal horizon > 1, adversary on the ACTION channel, linear (non-exponential) reward with penalty ---
    hz = {'obs_goal_horizon': 3, 'rew_exponential': False, 'adversary_disturbance': 'action',
          'adversary_disturbance_scale': 0.02, 'use_constraint_penalty': True, 'constraint_penalty': 0.1, 'episode_len_sec': 3,
          'task_info': {'trajectory_type': 'circle', 'num_cycles': 2, 'trajectory_plane': 'xz',
                        'trajectory_position_offset': [0, 1.2], 'trajectory_scale': 0.6}}
    rollout_case('quadrotor_2D_horizon', 'quadrotor',
                 load_task_config('quadrotor', f'{rl}/quadrotor_2D/quadrotor_2D_track.yaml', hz),
                 n_envs=3, n_steps=260, seed=5, act_scale=0.3, adversary=True)
    # --- 3-D stabilisation (goal_reached / stale out_of_bounds paths in 3-D) ---
    rollout_case('quadrotor_3D_stab', 'quadrotor',
                 load_task_config('quadrotor', f'{rl}/quadrotor_3D/quadrotor_3D_track.yaml',
                                  {'task': 'stabilization', 'episode_len_sec': 3,
                                   'task_info': {'stabilization_goal': [0.3, 0.4, 1.3], 'stabilization_goal_tolerance': 0.35}}),
                 n_envs=3, n_steps=240, seed=6, act_scale=0.3)
    # --- cartpole: adversary on the dynamics channel, trajectory tracking with a 2-row goal horizon ---
    rollout_case('cartpole_adversary', 'cartpole',
                 load_task_config('cartpole', f'{rl}/cartpole/cartpole_stab.yaml',
                                  {'task': 'traj_tracking', 'obs_goal_horizon': 2, 'adversary_disturbance': 'dynamics',
                                   'adversary_disturbance_scale': 0.5, 'episode_len_sec': 4,
                                   'task_info': {'trajectory_type': 'circle', 'num_cycles': 1, 'trajectory_plane': 'zx',
                                                 'trajectory_position_offset': [0, 0], 'trajectory_scale': 0.3}}),
                 n_envs=3, n_steps=200, seed=8, act_scale=0.5, adversary=True)
    gae_cases()
    shipped_models()
    # --- closed loop with the shipped policies: full-length episodes, time-limit truncation ---
    pol = dict(np.load(os.path.join(HERE, 'policies.npz')))
    rollout_case('quadrotor_2D_track_policy', 'quadrotor',
                 load_task_config('quadrotor', f'{rl}/quadrotor_2D/quadrotor_2D_track.yaml'),
                 n_envs=3, n_steps=600, seed=77, act_scale=0.02, policy=(pol, 'quadrotor_2D_track', 'tanh'))
    rollout_case('quadrotor_3D_track_policy', 'quadrotor',
                 load_task_config('quadrotor', f'{rl}/quadrotor_3D/quadrotor_3D_track.yaml'),
                 n_envs=2, n_steps=520, seed=78, act_scale=0.0, policy=(pol, 'quadrotor_3D_track', 'tanh'))
    rollout_case('cartpole_stab_policy', 'cartpole',
                 load_task_config('cartpole', f'{rl}/cartpole/cartpole_stab.yaml'),
                 n_envs=3, n_steps=400, seed=79, act_scale=0.0, policy=(pol, 'cartpole_stab', 'leaky_relu'))


if __name__ == '__main__':
    main()

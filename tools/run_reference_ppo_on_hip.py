#!/usr/bin/env python3
"""The reference's OWN PPO / SAC classes driven through HipVecEnv — the executed form of INTEGRATION.md's binding.

    python tools/run_reference_ppo_on_hip.py [--reference /root/reference] [--algo ppo|sac] [--envs 64] [--steps 3]

Needs a machine that has BOTH a HIP device and the reference's Python.  The gpurun box has no /root/reference, so
tools/stage_reference.py copies the package as untracked scratch to oracle/_ref/reference (git-ignored, ships with the
snapshot); without either the script prints SKIP and exits 0.
What it does — exactly the one-line change a maintainer makes in controllers/ppo/ppo.py:25 / sac/sac.py:
    from safe_control_gym.envs.env_wrappers.vectorized_env import make_vec_envs
 -> from safe_control_gym_amd.record_episode_statistics import make_vec_envs
(applied here by rebinding the name in the imported module), then `ctrl = PPO(env_func, training=True, ...)`,
`ctrl.reset()`, `ctrl.train_step()` x N with the reference's VecRecordEpisodeStatistics wrapper, PPOBuffer, PPOAgent and
logging untouched.  Missing third-party wheels (gymnasium, casadi, pybullet for the single eval env, tensorboard) are
replaced by tests/golden/ref_stubs.py where absent.
"""
import argparse
import functools
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', default=None, help='default: tests/golden/ref_stubs.reference_root()')
    ap.add_argument('--algo', default='ppo', choices=['ppo', 'sac'])
    ap.add_argument('--envs', type=int, default=64)
    ap.add_argument('--steps', type=int, default=3)
    args = ap.parse_args()
    import torch
    from tests.golden import ref_stubs
    args.reference = args.reference or ref_stubs.reference_root() or '/root/reference'
    if not os.path.isdir(os.path.join(args.reference, 'safe_control_gym')):
        print(f'SKIP: no reference checkout at {args.reference}')
        return 0
    if not torch.cuda.is_available():
        print('SKIP: no HIP device')
        return 0
    ref_stubs.REFERENCE_ROOT = args.reference
    ref_stubs.install()
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:                                           # noqa: BLE001
        tb = types.ModuleType('torch.utils.tensorboard')
        tb.SummaryWriter = type('SummaryWriter', (), {'__init__': lambda s, *a, **k: None, 'add_scalar': lambda s, *a, **k: None,
                                                      'close': lambda s: None, 'flush': lambda s: None})
        sys.modules['torch.utils.tensorboard'] = tb
    import yaml
    import safe_control_gym.envs  # noqa: F401  (the package __init__ registers 'cartpole' / 'quadrotor', envs/__init__.py:5-11)
    from safe_control_gym.utils.registration import make
    from safe_control_gym_amd.record_episode_statistics import make_vec_envs
    over = yaml.safe_load(open(os.path.join(args.reference, 'examples/rl/config_overrides/quadrotor_2D/quadrotor_2D_track.yaml')))
    task_config = over['task_config']
    env_func = functools.partial(make, 'quadrotor', output_dir='/tmp/scg_ref_run', **task_config)      # train_rl_controller.py:32-36
    if args.algo == 'ppo':
        import safe_control_gym.controllers.ppo.ppo as mod
        algo_yaml, cls = 'safe_control_gym/controllers/ppo/ppo.yaml', 'PPO'
    else:
        import safe_control_gym.controllers.sac.sac as mod
        algo_yaml, cls = 'safe_control_gym/controllers/sac/sac.yaml', 'SAC'
    mod.make_vec_envs = make_vec_envs                           # <- the binding
    cfg = yaml.safe_load(open(os.path.join(args.reference, algo_yaml)))
    cfg.update(rollout_batch_size=args.envs, num_workers=1, tensorboard=False)
    if args.algo == 'ppo':
        cfg.update(rollout_steps=16, mini_batch_size=256, opt_epochs=2)
    else:
        cfg.update(warm_up_steps=args.envs * 2, train_interval=args.envs, train_batch_size=64)
    ctrl = getattr(mod, cls)(env_func, training=True, output_dir='/tmp/scg_ref_run', use_gpu=False, seed=3, **cfg)
    from safe_control_gym_amd.vec_env import HipVecEnv
    inner = ctrl.env.venv if hasattr(ctrl.env, 'venv') else ctrl.env
    assert isinstance(inner, HipVecEnv), type(inner)
    ctrl.reset()
    for k in range(args.steps if args.algo == 'ppo' else args.steps * 8):
        res = ctrl.train_step()
        print(f'[{cls}.train_step {k}]', {a: (round(float(b), 5) if isinstance(b, (int, float)) else b) for a, b in res.items()})
    ctrl.close()
    print('OK: the reference', cls, 'ran on HipVecEnv through make_vec_envs')
    return 0


if __name__ == '__main__':
    sys.exit(main())

#!/usr/bin/env python3
"""The reference's own EXAMPLE SCRIPTS, run unmodified against this package's single-env facade.

  lqr:  examples/lqr/lqr_experiment.py (BASELINE config #1) — its ConfigFactory, YAML overrides (config_overrides/cartpole/
        {cartpole_stab,<algo>_cartpole_stab}.yaml), registry, `LQR` / `iLQR` controller, `BaseExperiment`, `RecordDataWrapper`,
        `MetricExtractor`;
  rl:   examples/rl/rl_experiment.py — the reference's `PPO` / `SAC` class in evaluation mode LOADS THE SHIPPED CHECKPOINT
        (examples/rl/models/<algo>/<algo>_model_<system>_<task>.pt, trained upstream on the real PyBullet envs) and `BaseExperiment`
        evaluates it for one episode with the overrides of rl_experiment.sh (training=False, randomized_init=False).
The ONE change is the one INTEGRATION.md describes: the registry's env ids point at `safe_control_gym_amd.benchmark_env:CartPole` /
`:Quadrotor` instead of the PyBullet envs.

    python tools/run_reference_example.py lqr [--algo lqr|ilqr] [--stub-handle]
    python tools/run_reference_example.py rl --algo ppo|sac --system cartpole|quadrotor_2D|quadrotor_3D --task stab|track [--stub-handle]

Needs the reference checkout (build container: /root/reference; GPU box: the scratch copy tools/stage_reference.py stages) and runs it
under tests/golden/ref_stubs.py (stand-ins for gymnasium / casadi / pybullet / munch / dict_deep / tensorboard, all absent in this image).
--stub-handle: no GPU — the facade's batch-of-1 handle is the oracle-backed stand-in of the CPU suite (tests/test_facade_cpu.py);
without it the handle is the real HipVecEnv.  Prints the example's own metrics ("FINAL METRICS ..." / "METRICS {...}").
"""
import argparse
import importlib.util
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Munch(dict):
    """munch.Munch for ConfigFactory.merge (utils/configuration.py:92): attribute access on nested dicts."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    __setattr__ = dict.__setitem__


def munchify(x):
    if isinstance(x, dict):
        return Munch({k: munchify(v) for k, v in x.items()})
    return [munchify(v) for v in x] if isinstance(x, list) else x


def deep_set(d, key, value):                        # dict_deep.deep_set for --kv_overrides
    ks = key.split('.')
    for k in ks[:-1]:
        d = d.setdefault(k, {})
    d[ks[-1]] = value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('example', choices=['lqr', 'rl'])
    ap.add_argument('--algo', default=None)
    ap.add_argument('--system', default='quadrotor_2D', choices=['cartpole', 'quadrotor_2D', 'quadrotor_3D'])
    ap.add_argument('--task', default='track', choices=['stab', 'track'])
    ap.add_argument('--stub-handle', action='store_true')
    a = ap.parse_args()
    a.algo = a.algo or ('lqr' if a.example == 'lqr' else 'ppo')
    import types

    from tests.golden import ref_stubs
    ref = ref_stubs.reference_root()
    if ref is None:
        sys.exit('no reference checkout on this machine')
    ref_stubs.install()
    sys.modules['munch'].munchify, sys.modules['munch'].Munch = munchify, Munch
    sys.modules['dict_deep'].deep_set = deep_set
    tb = types.ModuleType('torch.utils.tensorboard')           # ExperimentLogger's writer (utils/logging.py); nothing is logged here
    tb.SummaryWriter = type('SummaryWriter', (), {'__init__': lambda s, *a, **k: None, 'add_scalar': lambda s, *a, **k: None,
                                                  'close': lambda s: None, 'flush': lambda s: None})
    sys.modules.setdefault('torch.utils.tensorboard', tb)
    import matplotlib
    matplotlib.use('Agg')
    import safe_control_gym.envs  # noqa: F401  (registers the reference's env ids)
    from safe_control_gym.utils.registration import register, registry
    # ---- the one change of INTEGRATION.md ---------------------------------------------------------------------------------
    registry.specs['cartpole'].entry_point = 'safe_control_gym_amd.benchmark_env:CartPole'
    registry.specs['quadrotor'].entry_point = 'safe_control_gym_amd.benchmark_env:Quadrotor'
    # ----------------------------------------------------------------------------------------------------------------------
    # (controllers/__init__.py registers every controller at once, MPC's casadi / gpytorch imports included: register the one that runs)
    pkg, cls = {'lqr': ('lqr', 'lqr:LQR'), 'ilqr': ('lqr', 'ilqr:iLQR'), 'ppo': ('ppo', 'ppo:PPO'), 'sac': ('sac', 'sac:SAC')}[a.algo]
    register(idx=a.algo, entry_point=f'safe_control_gym.controllers.{pkg}.{cls}', config_entry_point=f'safe_control_gym.controllers.{pkg}:{a.algo}.yaml')
    if a.stub_handle:
        import safe_control_gym_amd.benchmark_env as B
        from tests.test_facade_cpu import _OracleBackedVec
        B.HipVecEnv = _OracleBackedVec
    with tempfile.TemporaryDirectory() as tmp:
        if a.example == 'lqr':
            ov = os.path.join(ref, 'examples', 'lqr', 'config_overrides', 'cartpole')
            sys.argv = ['lqr_experiment.py', '--algo', a.algo, '--task', 'cartpole', '--overrides', os.path.join(ov, 'cartpole_stab.yaml'),
                        os.path.join(ov, f'{a.algo}_cartpole_stab.yaml')]
            spec = importlib.util.spec_from_file_location('lqr_experiment', os.path.join(ref, 'examples', 'lqr', 'lqr_experiment.py'))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            os.chdir(tmp)
            mod.run(gui=False, plot=False, n_episodes=1, n_steps=None, save_data=False)
        else:
            name = 'cartpole' if a.system == 'cartpole' else 'quadrotor'
            ov = os.path.join(ref, 'examples', 'rl', 'config_overrides', a.system)
            sys.argv = ['rl_experiment.py', '--task', name, '--algo', a.algo, '--overrides', os.path.join(ov, f'{a.system}_{a.task}.yaml'),
                        os.path.join(ov, f'{a.algo}_{a.system}.yaml'), '--kv_overrides', 'algo_config.training=False',
                        'task_config.randomized_init=False']                                   # (rl_experiment.sh)
            spec = importlib.util.spec_from_file_location('rl_experiment', os.path.join(ref, 'examples', 'rl', 'rl_experiment.py'))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            os.makedirs(os.path.join(tmp, 'models'))
            os.symlink(os.path.join(ref, 'examples', 'rl', 'models', a.algo), os.path.join(tmp, 'models', a.algo))   # the shipped checkpoints
            os.chdir(tmp)
            _, _, metrics = mod.run(gui=False, plot=False, n_episodes=1, n_steps=None, curr_path=tmp)
            import json
            print('METRICS ' + json.dumps({k: float(v) for k, v in metrics.items()}))


if __name__ == '__main__':
    main()

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
    python tools/run_reference_example.py matrix [--stub-handle]
    python tools/run_reference_example.py train --algo ppo|sac|safe_explorer_ppo --system … --task … [--env-steps 1200] [--stub-handle]

  matrix: the reference's OWN TEST MATRIX — tests/test_examples/test_lqr.py (LQR / iLQR x stab / track x cartpole / quadrotor_2D / _3D: 12
        cases), test_rl.py (ppo / sac / safe_explorer_ppo with the shipped checkpoints: 18), test_pid.py (4) — with the arguments those tests
        pass (`n_steps=10`, `algo_config.max_iterations=2`, `algo_config.training=False`), and test_no_controller.py (verbose_api.py on both
        systems: 2), one line per case.  (test_mpc / test_cbf / test_mpsc need CasADi + IPOPT, test_hpo needs optuna / MySQL: not runnable
        here at all.)

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


def unmunchify(x):
    if isinstance(x, dict):
        return {k: unmunchify(v) for k, v in x.items()}
    return [unmunchify(v) for v in x] if isinstance(x, (list, tuple)) else x


def deep_set(d, key, value):                        # dict_deep.deep_set for --kv_overrides
    ks = key.split('.')
    for k in ks[:-1]:
        d = d.setdefault(k, {})
    d[ks[-1]] = value


def matrix(ref):
    """The parametrisations of the reference's tests/test_examples/{test_lqr,test_rl,test_pid}.py, called the way those tests call them."""
    import contextlib
    import io

    def load(rel, name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ref, rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    os.chdir(ref)                                               # the tests' relative override paths
    lqr, rl, pid = (load(f'examples/{d}/{d}_experiment.py', f'{d}_experiment') for d in ('lqr', 'rl', 'pid'))
    models = tempfile.mkdtemp()                                 # rl_experiment.py deletes <curr_path>/temp: give it a scratch curr_path
    os.makedirs(os.path.join(models, 'models'))
    for alg in ('ppo', 'sac', 'safe_explorer_ppo'):
        os.symlink(os.path.join(ref, 'examples', 'rl', 'models', alg), os.path.join(models, 'models', alg))
    cases = []
    for SYS in ('cartpole', 'quadrotor_2D', 'quadrotor_3D'):
        NAME = 'quadrotor' if 'quadrotor' in SYS else SYS
        for TASK in ('stab', 'track'):
            for ALGO in ('lqr', 'ilqr'):
                cases.append((f'test_lqr[{ALGO}-{TASK}-{SYS}]', ['--algo', ALGO, '--task', NAME, '--overrides',
                              f'./examples/lqr/config_overrides/{SYS}/{SYS}_{TASK}.yaml', f'./examples/lqr/config_overrides/{SYS}/{ALGO}_{SYS}_{TASK}.yaml',
                              '--kv_overrides', 'algo_config.max_iterations=2'],
                              lambda: lqr.run(gui=False, plot=False, n_episodes=None, n_steps=10, save_data=False)))
            for ALGO in ('ppo', 'sac', 'safe_explorer_ppo'):
                cases.append((f'test_rl[{ALGO}-{TASK}-{SYS}]', ['--algo', ALGO, '--task', NAME, '--overrides',
                              f'./examples/rl/config_overrides/{SYS}/{SYS}_{TASK}.yaml', f'./examples/rl/config_overrides/{SYS}/{ALGO}_{SYS}.yaml',
                              '--kv_overrides', 'algo_config.training=False'],
                              lambda: rl.run(gui=False, plot=False, n_episodes=None, n_steps=10, curr_path=models)))
    for SYS in ('quadrotor_2D', 'quadrotor_3D'):
        for TASK in ('stab', 'track'):
            cases.append((f'test_pid[{TASK}-{SYS}]', ['--algo', 'pid', '--task', 'quadrotor', '--overrides',
                          f'./examples/pid/config_overrides/{SYS}/{SYS}_{TASK}.yaml'],
                          lambda: pid.run(gui=False, n_episodes=None, n_steps=10, save_data=False)))
    # tests/test_examples/test_no_controller.py: examples/no_controller/verbose_api.py prints `pybullet.getDynamicsInfo(env.DRONE_ID, env.PYB_CLIENT)`
    # first — there is no Bullet body behind the facade (ids -1), so the Bullet stand-in of this harness answers that one call with a note
    vb = load('examples/no_controller/verbose_api.py', 'verbose_api')
    sys.modules['pybullet'].getDynamicsInfo = lambda **k: ('no Bullet body behind the HIP facade',)
    for t in ('cartpole', 'quadrotor'):
        cases.append((f'test_verbose_api_{t}', ['--task', t, '--overrides', './examples/no_controller/verbose_api.yaml'], vb.run))
    n_ok = 0
    for tag, argv, call in cases:
        sys.argv[1:] = argv
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                call()
            n_ok += 1
            print('PASSED', tag, flush=True)
        except BaseException as e:                              # noqa: BLE001  (SystemExit from argparse included)
            print('FAILED', tag, type(e).__name__, str(e)[:200], flush=True)
    print(f'MATRIX {n_ok} passed of {len(cases)}')
    return 0 if n_ok == len(cases) else 1


def train(ref, a):
    """safe_control_gym/experiments/train_rl_controller.py::train(), unmodified, with the arguments of examples/rl/train_rl_model.sh and a small
    budget: the reference's own PPO / SAC / Safe-Explorer training loop — its make_vec_envs (DummyVecEnv of `rollout_batch_size` facade envs),
    RecordEpisodeStatistics wrappers, buffers, agent updates, periodic evaluation, logger, checkpointing."""
    for n in ('tensorboard', 'tensorboard.backend', 'tensorboard.backend.event_processing', 'tensorboard.backend.event_processing.event_accumulator'):
        sys.modules.setdefault(n, __import__('types').ModuleType(n))      # utils/plotting.py reads TensorBoard logs; plotting is skipped below
    sys.modules['tensorboard.backend.event_processing.event_accumulator'].EventAccumulator = object
    os.chdir(ref)
    import safe_control_gym.experiments.train_rl_controller as TR
    TR.make_plots = lambda config: None
    name = 'cartpole' if a.system == 'cartpole' else 'quadrotor'
    out = a.output_dir or tempfile.mkdtemp()
    if a.env_steps == 0:                                                     # the reference's own budget, exactly train_rl_model.sh's arguments
        kv = ['task_config.init_state=None', 'task_config.randomized_init=True']
    else:
        kv = ['task_config.init_state=None', 'task_config.randomized_init=True', f'algo_config.max_env_steps={a.env_steps}',
              'algo_config.rollout_batch_size=2', 'algo_config.eval_batch_size=2', f'algo_config.eval_interval={a.env_steps // 2}',
              f'algo_config.log_interval={a.env_steps // 2}', 'algo_config.save_interval=0', 'algo_config.num_checkpoints=0']
        kv += (['algo_config.rollout_steps=100', 'algo_config.mini_batch_size=64'] if a.algo != 'sac' else
               ['algo_config.warm_up_steps=200', 'algo_config.train_interval=100', 'algo_config.train_batch_size=64'])
    overrides = [f'./examples/rl/config_overrides/{a.system}/{a.algo}_{a.system}.yaml', f'./examples/rl/config_overrides/{a.system}/{a.system}_{a.task}.yaml']
    if a.algo == 'safe_explorer_ppo':                                        # (train_rl_model.sh: the shipped pre-trained safety layer)
        kv += [f'algo_config.pretrained={ref}/examples/rl/models/{a.algo}/{a.algo}_pretrain_{a.system}_{a.task}.pt']
    sys.argv[1:] = ['--algo', a.algo, '--task', name, '--overrides'] + overrides + ['--output_dir', out, '--seed', '2', '--kv_overrides'] + kv
    TR.train()
    import torch
    ck = torch.load(os.path.join(out, 'model_latest.pt'), weights_only=False, map_location='cpu')
    print('TRAINED', a.algo, a.system, a.task, 'checkpoint keys', sorted(ck)[:6], 'files', sorted(os.listdir(out)))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('example', choices=['lqr', 'rl', 'matrix', 'train'])
    ap.add_argument('--env-steps', type=int, default=1200, help='train: small budget (0 = the YAML budget of the reference, e.g. 300 000 steps for PPO on cartpole)')
    ap.add_argument('--output-dir', default=None)
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
    sys.modules['munch'].munchify, sys.modules['munch'].Munch, sys.modules['munch'].unmunchify = munchify, Munch, unmunchify
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
    ctrls = {'lqr': ('lqr', 'lqr:LQR', 'lqr'), 'ilqr': ('lqr', 'ilqr:iLQR', 'ilqr'), 'ppo': ('ppo', 'ppo:PPO', 'ppo'), 'sac': ('sac', 'sac:SAC', 'sac'),
             'pid': ('pid', 'pid:PID', 'pid'), 'safe_explorer_ppo': ('safe_explorer', 'safe_ppo:SafeExplorerPPO', 'safe_ppo')}
    for idx in (ctrls if a.example in ('matrix', 'train') else [a.algo]):
        pkg, cls, yml = ctrls[idx]
        register(idx=idx, entry_point=f'safe_control_gym.controllers.{pkg}.{cls}', config_entry_point=f'safe_control_gym.controllers.{pkg}:{yml}.yaml')
    if a.stub_handle:
        import safe_control_gym_amd.benchmark_env as B
        from tests.test_facade_cpu import _OracleBackedVec
        B.HipVecEnv = _OracleBackedVec
    if a.example == 'matrix':
        return matrix(ref)
    if a.example == 'train':
        return train(ref, a)
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
    sys.exit(main())

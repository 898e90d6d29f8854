"""The single-env facade (safe_control_gym_amd/benchmark_env.py) in the CPU suite.

The facade sits on a batch-of-1 HipVecEnv, which needs a GPU; here that handle is replaced by stand-ins so that everything ABOVE it
runs without one:
  * a stub that only carries the EnvSpec — constructors, attribute surface, the prior-model flow of upstream
    (`_setup_symbolic`, `_randomize_values_by_info`);
  * `_OracleBackedVec`, which fills the tensor interface from a private oracle instance — the facade's host logic (flag decoding,
    info key sets, TimeLimit.truncated, action attributes) against the oracle on random configs, and the REFERENCE's own single-env
    callers driven through it unmodified: `LQR`, `iLQR`, `PID`, and the experiment harness `BaseExperiment` + `RecordDataWrapper` +
    `MetricExtractor` (BASELINE config #1, examples/lqr/lqr_experiment.py).  Those import the reference checkout (build container, or
    the scratch copy tools/stage_reference.py stages) under tests/golden/ref_stubs.py and are skipped where there is none.
The kernels' half of the same comparisons is tests/test_gpu_config_fuzz.py / tests/test_gpu_facade.py.
"""
import os

import numpy as np
import pytest


def test_facade_symbolic_model_follows_upstreams_prior_prop_flow():
    """Upstream builds `env.symbolic` at construction with the env's TRUE parameters — `_setup_symbolic()` is called without
    arguments (quadrotor.py:326, cartpole.py:236), the config's `prior_prop` is only stored (benchmark_env.py:155) — and controllers
    install a prior through `BaseController.get_prior -> env._setup_symbolic(prior_prop=...)` (base_controller.py:177-191).  The
    facade's half of that, without a GPU: the method on an instance whose batch-of-1 handle is replaced by its EnvSpec.
    (tests/golden/sweep_symbolic.py compares both models with the reference's own expressions on random parameters.)"""
    import types

    import numpy as np
    from safe_control_gym_amd.benchmark_env import CartPole, Quadrotor
    from safe_control_gym_amd.env_config import EnvSpec
    from safe_control_gym_amd.registration import load_task
    env_id, cfg = load_task('quadrotor_2D_track')
    q = Quadrotor.__new__(Quadrotor)
    q._venv = types.SimpleNamespace(spec=EnvSpec(env_id, dict(cfg, prior_prop={'M': 0.04, 'Iyy': 2e-5})))
    q._setup_symbolic()                                       # what the constructor does: the stored prior_prop is NOT applied
    assert q.symbolic.quad_mass == 0.027 and np.allclose(q.symbolic.U_EQ, 0.027 * 9.8 / 2)
    x, u = np.array([0.1, 0.2, 1.0, -0.1, 0.05, 0.3]), np.array([0.15, 0.12])
    f_true = q.symbolic.f(x, u)
    q._setup_symbolic(prior_prop={'M': 0.04, 'Iyy': 2e-5})    # what BaseController.get_prior does
    assert q.symbolic.quad_mass == 0.04 and q.symbolic.quad_Iyy == 2e-5 and np.allclose(q.symbolic.U_EQ, 0.04 * 9.8 / 2)
    f_prior = q.symbolic.f(x, u)
    assert np.allclose(f_prior[[0, 2, 4]], f_true[[0, 2, 4]]) and not np.allclose(f_prior[[1, 3, 5]], f_true[[1, 3, 5]])
    env_id, cfg = load_task('cartpole_stab')
    c = CartPole.__new__(CartPole)
    c._venv = types.SimpleNamespace(spec=EnvSpec(env_id, dict(cfg)))
    c._setup_symbolic(prior_prop={'pole_length': 0.7})
    assert c.symbolic.pole_length == 0.7 and c.symbolic.cart_mass == c._venv.spec.CART_MASS

    # BaseController.get_prior with `randomize_prior_prop` (base_controller.py:180-187): additive draws from the env's host generator
    q.np_random = np.random.default_rng(4)
    got = q._randomize_values_by_info({'M': 0.03, 'Iyy': 1.4e-5}, {'M': {'distrib': 'uniform', 'low': -0.001, 'high': 0.001},
                                                                   'Iyy': {'distrib': 'choice', 'args': [[1e-6, 2e-6]]}})
    ref = np.random.default_rng(4)
    assert got['M'] == 0.03 + ref.uniform(low=-0.001, high=0.001) and got['Iyy'] == 1.4e-5 + ref.choice([1e-6, 2e-6])


def test_facade_constructors_run_on_a_stub_handle(monkeypatch):
    """The whole `BenchmarkEnv.__init__` of the single-env facade (attribute surface the reference's controllers read, prior model,
    randomisation tables) with the batch-of-1 HIP handle replaced by a stub that only carries the EnvSpec: constructor regressions show
    up in the CPU suite, not first on the GPU box."""
    import safe_control_gym_amd.benchmark_env as B
    from safe_control_gym_amd.env_config import EnvSpec
    from safe_control_gym_amd.registration import load_task

    class StubVec:
        def __init__(self, name, n, seed=0, device=None, dtype=None, return_numpy=False, auto_reset=False, specialize='auto', **cfg):
            assert n == 1 and auto_reset is False
            self.spec = EnvSpec(name, cfg)
            self.dtype, self.device, self._adv = dtype, 'cpu', None

        def seed(self, s):
            pass

        def close(self):
            pass
    monkeypatch.setattr(B, 'HipVecEnv', StubVec)
    for task, cls, prior in (('cartpole_stab', B.CartPole, {'pole_length': 0.6}), ('quadrotor_2D_track', B.Quadrotor, {'M': 0.03}),
                             ('quadrotor_3D_track_disturbed', B.Quadrotor, {'M': 0.03, 'Izz': 3e-5})):
        env_id, cfg = load_task(task)
        e = cls(seed=3, **dict(cfg, prior_prop=prior))
        for k in ('X_GOAL', 'U_GOAL', 'TASK', 'COST', 'NAME', 'CTRL_FREQ', 'PYB_FREQ', 'CTRL_TIMESTEP', 'EPISODE_LEN_SEC', 'TASK_INFO',
                  'action_space', 'observation_space', 'state_space', 'physical_action_bounds', 'constraints', 'STATE_LABELS', 'STATE_UNITS',
                  'ACTION_LABELS', 'ACTION_UNITS', 'np_random', 'INIT_STATE_RAND_INFO', 'INERTIAL_PROP_RAND_INFO', 'PRIOR_PROP', 'symbolic',
                  'done_on_out_of_bound', 'pyb_step_counter', 'ctrl_step_counter', 'state_dim', 'action_dim', 'obs_dim'):
            assert hasattr(e, k), (task, k)
        assert e.PRIOR_PROP == prior and e.symbolic.nx == e.state_dim
        if env_id == 'quadrotor':
            assert e.symbolic.quad_mass == 0.027 and e.QUAD_TYPE in (2, 3)          # the stored prior is not applied at construction
            e._setup_symbolic(prior_prop=e.PRIOR_PROP)
            assert e.symbolic.quad_mass == 0.03
        with pytest.raises(RuntimeError):
            e.step([0.0] * e.action_dim)                                           # before reset (benchmark_env.py:230-235)
        e.close()


class _OracleBackedVec:
    """Stand-in for the batch-of-1 HipVecEnv under the facade (CPU suite only): the slice of the tensor interface BenchmarkEnv uses,
    filled from a private oracle instance — so tests/test_gpu_config_fuzz.py::facade_vs_oracle exercises the FACADE's host logic (flag
    decoding, info key sets, TimeLimit.truncated, action attributes, reset info) without a GPU.  The kernels' half of that comparison is
    the GPU test of the same name."""

    def __init__(self, name, n, seed=0, device=None, dtype=None, return_numpy=False, auto_reset=False, specialize='auto', **cfg):
        import types

        import torch
        from oracle.envs import make_oracle_env, make_rng
        from safe_control_gym_amd.env_config import EnvSpec
        assert n == 1 and auto_reset is False
        self.spec = EnvSpec(name, cfg)
        self.spec.num_constraints_or_zero = len(self.spec.con_rows)
        self.o = make_oracle_env(name, 1, make_rng('philox', 1, seed), **cfg)
        self.dtype, self.device, self._adv, self.num_envs = torch.float64, torch.device('cpu'), None, 1
        self.out = types.SimpleNamespace(state=None, c_values=None)
        self._torch = torch

    def seed(self, s):
        pass

    def close(self):
        pass

    def _t(self, a):
        return self._torch.as_tensor(np.asarray(a, dtype=np.float64))

    def _fill(self, obs, c_values):
        self.out.obs = self._t(obs)
        self.out.state = self._t(self.o.state.T)
        self.out.c_values = None if c_values is None else self._t(np.asarray(c_values).T)

    def reset_tensors(self):
        obs, info = self.o.reset()
        nrows = len(self.spec.con_rows)
        c = None
        if nrows:
            c = np.zeros((1, nrows))
            if 'constraint_values' in info:
                c[:, :info['constraint_values'].shape[1]] = info['constraint_values']
        self._fill(obs, c)
        return self.out.obs

    def _reset_info(self, host, i, with_constraints=False):
        from safe_control_gym_amd.vec_env import HipVecEnv
        return HipVecEnv._reset_info(self, host, i, with_constraints)

    def physical_parameters(self, i):
        o = self.o
        if self.spec.name == 'cartpole':
            return {'pole_effective_length': float(o.pole_length_env[i]), 'pole_mass': float(o.pole_mass_env[i]), 'cart_mass': float(o.cart_mass_env[i])}
        return {'quadrotor_mass': float(o.mass_env[i]), 'quadrotor_inertia': [float(v) for v in o.J_env[i]]}

    def set_adversary_control(self, a):
        from safe_control_gym_amd.vec_env import HipVecEnv
        self._as_device = lambda x, cols: self._t(x).reshape(1, cols)
        HipVecEnv.set_adversary_control(self, a)

    def step_tensors(self, a, adv=None):
        o = self.o
        if adv is not None:
            o.adv_action = adv.numpy().copy()                       # already clipped / scaled / offset by set_adversary_control
        obs, rew, done, info = o.step(a.numpy())
        flags = (info['TimeLimit.truncated'] & info['time_limit_reached']).astype(np.uint8) * 1 + (info['constraint_violation'] > 0).astype(np.uint8) * 2
        if 'out_of_bounds' in info:
            flags = flags + info['out_of_bounds'].astype(np.uint8) * 4
        if 'goal_reached' in info:
            flags = flags + info['goal_reached'].astype(np.uint8) * 8
        self._fill(obs, info.get('constraint_values'))
        self.out.reward, self.out.done, self.out.flags = self._t(rew), self._t(done.astype(np.uint8)), self._torch.as_tensor(flags)
        self.out.mse, self.out.noisy_action = self._t(info['mse']), self._t(np.asarray(o.current_noisy_physical_action).T)
        return self.out


@pytest.mark.parametrize('seed', range(3))
@pytest.mark.parametrize('system', ['cartpole', 'quadrotor_1D', 'quadrotor_2D', 'quadrotor_3D'])
def test_facade_host_logic_on_an_oracle_backed_handle(system, seed, monkeypatch):
    import safe_control_gym_amd.benchmark_env as B
    from tests.test_gpu_config_fuzz import facade_vs_oracle
    monkeypatch.setattr(B, 'HipVecEnv', _OracleBackedVec)

    def make(env_id, seed=None, **cfg):
        return {'cartpole': B.CartPole, 'quadrotor': B.Quadrotor}[env_id](seed=seed, **cfg)
    facade_vs_oracle(system, seed, make)


@pytest.mark.parametrize('system', ['cartpole', 'quadrotor_2D'])
def test_the_references_own_lqr_controller_drives_the_facade(system, monkeypatch, tmp_path):
    """BASELINE config #1 (examples/lqr/lqr_experiment.py: LQR stabilisation, one env) with the REFERENCE's own controller class
    (controllers/lqr/lqr.py + lqr_utils.py, imported from the reference checkout): `LQR(env_func, q_lqr, r_lqr)` builds its env through
    `env_func()`, takes the prior model through `BaseController.get_prior`, linearises it with `model.df_func(X_EQ, U_EQ)`, solves the
    discrete Riccati equation, and `select_action(obs, info)` closes the loop on `env.step` — every call lands on this package's facade
    (here on the oracle-backed handle of the CPU suite) and the task is solved."""
    import functools
    import sys

    from tests.golden import ref_stubs
    if ref_stubs.reference_root() is None:
        pytest.skip('needs the reference checkout (build container, or the scratch copy staged by tools/stage_reference.py)')
    ref_stubs.install()
    from safe_control_gym.controllers.lqr.lqr import LQR
    import safe_control_gym_amd.benchmark_env as B
    monkeypatch.setattr(B, 'HipVecEnv', _OracleBackedVec)
    if system == 'cartpole':
        cls, q, r = B.CartPole, [1, 1, 1, 1], [0.1]
        cfg = dict(ctrl_freq=15, pyb_freq=750, task='stabilization', task_info={'stabilization_goal': [1.0, 0.0], 'stabilization_goal_tolerance': 0.0},
                   episode_len_sec=6, cost='quadratic', rew_state_weight=q, rew_act_weight=r, done_on_out_of_bound=True, randomized_init=False,
                   normalized_rl_action_space=False, init_state={'init_x': -0.5, 'init_x_dot': 0.05, 'init_theta': 0.1, 'init_theta_dot': -0.05})
    else:
        cls, q, r = B.Quadrotor, [1, 1, 1, 1, 1, 1], [0.1, 0.1]
        cfg = dict(quad_type=2, ctrl_freq=50, pyb_freq=1000, task='stabilization', episode_len_sec=5, cost='quadratic', rew_state_weight=q,
                   rew_act_weight=r, task_info={'stabilization_goal': [0.5, 1.2], 'stabilization_goal_tolerance': 0.0}, done_on_out_of_bound=True,
                   randomized_init=False, normalized_rl_action_space=False, init_state={'init_x': 0.0, 'init_z': 1.0, 'init_theta': 0.05})
    env_func = functools.partial(cls, **cfg)
    ctrl = LQR(env_func, q_lqr=q, r_lqr=r, discrete_dynamics=True, output_dir=str(tmp_path), training=False, seed=42)
    assert ctrl.gain.shape == (len(r), len(q)) and ctrl.model is ctrl.env.symbolic
    env = env_func(seed=42)
    obs, info = env.reset()
    done, steps = False, 0
    while not done:
        obs, rew, done, info = env.step(ctrl.select_action(obs, info))
        steps += 1
    assert steps == env.CTRL_STEPS and info['TimeLimit.truncated'] is True          # the controller kept it in bounds to the time limit
    assert np.linalg.norm(obs - env.X_GOAL) < 0.05, obs
    ctrl.close(); env.close()
    for m in [k for k in sys.modules if k.startswith('safe_control_gym.') or k == 'safe_control_gym']:
        sys.modules.pop(m, None)                        # (keep the reference package out of the other tests' module table)


def test_the_references_own_ilqr_controller_learns_on_the_facade(monkeypatch, tmp_path):
    """controllers/lqr/ilqr.py of the reference, unmodified: `learn(env)` rolls the facade out (`env.reset` / `env.step` / `info`), runs
    its backward pass on `model.df_func` and `model.loss` — every output of which it reads through CasADi's `.toarray()` (ilqr.py:210-247:
    the analytic model's results are DM-like for that reason) — and its cost falls from iteration to iteration."""
    import functools
    import sys

    from tests.golden import ref_stubs
    if ref_stubs.reference_root() is None:
        pytest.skip('needs the reference checkout')
    ref_stubs.install()
    from safe_control_gym.controllers.lqr.ilqr import iLQR
    import safe_control_gym_amd.benchmark_env as B
    monkeypatch.setattr(B, 'HipVecEnv', _OracleBackedVec)
    cfg = dict(ctrl_freq=15, pyb_freq=750, task='stabilization', task_info={'stabilization_goal': [1.0, 0.0], 'stabilization_goal_tolerance': 0.0},
               episode_len_sec=4, cost='quadratic', rew_state_weight=[1, 1, 1, 1], rew_act_weight=[0.1], done_on_out_of_bound=True,
               randomized_init=False, normalized_rl_action_space=False,
               init_state={'init_x': -0.5, 'init_x_dot': 0.05, 'init_theta': 0.1, 'init_theta_dot': -0.05})
    env_func = functools.partial(B.CartPole, **cfg)
    ctrl = iLQR(env_func, q_lqr=[1, 1, 1, 1], r_lqr=[0.1], discrete_dynamics=True, max_iterations=4, output_dir=str(tmp_path), training=True, seed=42)
    ctrl.learn(env=env_func(seed=42))
    assert ctrl.input_ff_best is not None and ctrl.gains_fb_best is not None          # a feed-forward / feedback schedule was accepted
    env = env_func(seed=42)
    obs, info = env.reset()
    done, steps, total = False, 0, 0.0
    while not done:
        obs, rew, done, info = env.step(ctrl.select_action(obs, info))
        steps += 1
        total += rew
    assert steps == env.CTRL_STEPS and abs(obs[0] - 1.0) < 0.15 and abs(obs[2]) < 0.05, obs         # 4 s: the cart is almost at x = 1, pole up
    ctrl.close(); env.close()
    for m in [k for k in sys.modules if k.startswith('safe_control_gym.') or k == 'safe_control_gym']:
        sys.modules.pop(m, None)


def test_the_references_experiment_harness_runs_lqr_on_the_facade(monkeypatch, tmp_path, capsys):
    """examples/lqr/lqr_experiment.py's flow with the reference's OWN harness and controller: `BaseExperiment(env, ctrl, train_env)`,
    `launch_training()`, `run_evaluation(n_episodes=1)` — its `RecordDataWrapper` around the facade, `_execute_evaluations`' reset / step /
    `select_action` loop, `MetricExtractor` (experiments/base_experiment.py:90-165,310-492) — on envs built the way the example builds them
    (`env_func(randomized_init=False, init_state=<obs of a random env's reset>)`)."""
    import functools
    import sys

    from tests.golden import ref_stubs
    if ref_stubs.reference_root() is None:
        pytest.skip('needs the reference checkout')
    ref_stubs.install()
    from safe_control_gym.controllers.lqr.lqr import LQR
    from safe_control_gym.experiments.base_experiment import BaseExperiment
    import safe_control_gym_amd.benchmark_env as B
    monkeypatch.setattr(B, 'HipVecEnv', _OracleBackedVec)
    cfg = dict(ctrl_freq=15, pyb_freq=750, task='stabilization', task_info={'stabilization_goal': [1.0, 0.0], 'stabilization_goal_tolerance': 0.0},
               episode_len_sec=6, cost='quadratic', rew_state_weight=[1, 1, 1, 1], rew_act_weight=[0.1], done_on_out_of_bound=True,
               randomized_init=True, normalized_rl_action_space=False)
    env_func = functools.partial(B.CartPole, seed=42, **cfg)
    random_env = env_func()
    ctrl = LQR(env_func, q_lqr=[1, 1, 1, 1], r_lqr=[0.1], discrete_dynamics=True, output_dir=str(tmp_path), training=False)
    init_state, _ = random_env.reset()
    static_env = env_func(randomized_init=False, init_state=init_state)
    static_train_env = env_func(randomized_init=False, init_state=init_state)
    experiment = BaseExperiment(env=static_env, ctrl=ctrl, train_env=static_train_env)
    experiment.launch_training()
    trajs, metrics = experiment.run_evaluation(training=True, n_episodes=1)
    obs = np.asarray(trajs['obs'][0])
    assert obs.shape == (91, 4) and np.asarray(trajs['action'][0]).shape == (90, 1)            # 6 s at 15 Hz + the reset observation
    np.testing.assert_allclose(obs[0], init_state, atol=1e-12)                                 # the static env starts where the random one did
    assert np.linalg.norm(obs[-1] - static_env.X_GOAL) < 0.05
    for k in ('average_length', 'average_return', 'average_rmse', 'failure_rate', 'average_constraint_violation'):
        assert k in metrics, sorted(metrics)
    assert metrics['average_length'] == 90 and metrics['failure_rate'] == 0.0
    for e in (static_env, static_train_env, random_env):
        e.close()
    ctrl.close()
    for m in [k for k in sys.modules if k.startswith('safe_control_gym.') or k == 'safe_control_gym']:
        sys.modules.pop(m, None)


@pytest.mark.parametrize('task,rmse_max', [('quadrotor_2D_track', 0.2), ('quadrotor_3D_track', 0.4)])
def test_the_references_own_pid_controller_flies_the_facade(task, rmse_max, monkeypatch, tmp_path):
    """controllers/pid/pid.py of the reference (the DSL cascade PID, gains tuned upstream on the real PyBullet drone; it reads NAME,
    QUAD_TYPE, TASK, CTRL_TIMESTEP, X_GOAL and uses pybullet's quaternion helpers) on the shipped tracking configs through the facade:
    the whole figure-8 episode is flown inside the bounds, the time limit ends it, the tracking error stays small — 0.12 m / 0.25 m RMSE of
    the weighted state error measured here.  Physics under it in the CPU suite = the oracle's restatement of Bullet; nothing in the
    controller was written against it."""
    import functools
    import sys

    from tests.golden import ref_stubs
    if ref_stubs.reference_root() is None:
        pytest.skip('needs the reference checkout')
    ref_stubs.install()
    from safe_control_gym.controllers.pid.pid import PID
    import safe_control_gym_amd.benchmark_env as B
    from safe_control_gym_amd.registration import load_task
    monkeypatch.setattr(B, 'HipVecEnv', _OracleBackedVec)
    env_id, cfg = load_task(task)
    start = {'init_x': 0.0, 'init_z': 1.0} if '2D' in task else {'init_x': 0.0, 'init_y': 0.0, 'init_z': 1.0}
    cfg.update(cost='quadratic', normalized_rl_action_space=False, randomized_init=False, done_on_out_of_bound=True, constraints=None, init_state=start)
    env_func = functools.partial(B.Quadrotor, **cfg)
    ctrl = PID(env_func, output_dir=str(tmp_path), training=False, seed=1)
    env = env_func(seed=1)
    obs, info = env.reset()
    ctrl.reset()
    done, steps, mse = False, 0, []
    while not done:
        obs, rew, done, info = env.step(ctrl.select_action(obs, info))
        steps += 1
        mse.append(info['mse'])
    assert steps == env.CTRL_STEPS == 250 and info['TimeLimit.truncated'] is True and info['out_of_bounds'] is False
    assert np.sqrt(np.mean(mse)) < rmse_max
    ctrl.close(); env.close()
    for m in [k for k in sys.modules if k.startswith('safe_control_gym.') or k == 'safe_control_gym']:
        sys.modules.pop(m, None)


@pytest.mark.parametrize('algo', ['lqr', 'ilqr'])
def test_the_references_lqr_example_script_runs_unmodified(algo):
    """BASELINE config #1 as the reference ships it: examples/lqr/lqr_experiment.py with its own ConfigFactory, YAML overrides, registry,
    controller and experiment harness, run by tools/run_reference_example.py with the ONE change INTEGRATION.md describes (the
    registry's `cartpole` id -> this package's facade).  The example's own FINAL METRICS line: the 6 s episode is completed (90 steps at
    15 Hz), no failure, no constraint violation."""
    import os
    import re
    import subprocess
    import sys

    from tests.golden import ref_stubs
    if ref_stubs.reference_root() is None:
        pytest.skip('needs the reference checkout')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, 'tools', 'run_reference_example.py'), 'lqr', '--algo', algo, '--stub-handle'],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith('FINAL METRICS')]
    assert line, res.stdout[-2000:]
    m = dict(re.findall(r'(\w[\w.]*): ([-\d.e]+)', line[0]))
    assert float(m['average_length']) == 90.0 and float(m['failure_rate']) == 0.0 and float(m['average_constraint_violation']) == 0.0
    assert float(m['average_rmse']) < 0.6 and -30.0 < float(m['average_return']) < 0.0


# shipped checkpoint -> (episode length, lower bound on the return the reference's own evaluation reports for it here)
SHIPPED = [('ppo', 'quadrotor_2D', 'track', 250, 230.0), ('sac', 'quadrotor_2D', 'track', 250, 200.0), ('ppo', 'quadrotor_3D', 'track', 250, 200.0),
           ('ppo', 'cartpole', 'stab', 150, 115.0), ('sac', 'cartpole', 'stab', 150, 115.0)]


@pytest.mark.parametrize('algo,system,task,length,min_return', SHIPPED, ids=[f'{a}-{s}-{t}' for a, s, t, _, _ in SHIPPED])
def test_the_references_rl_example_evaluates_the_shipped_models(algo, system, task, length, min_return):
    """examples/rl/rl_experiment.py, unmodified (tools/run_reference_example.py rl): the reference's own `PPO` / `SAC` class in
    evaluation mode loads the checkpoint the reference SHIPS (trained upstream on the real PyBullet env) and its `BaseExperiment` evaluates
    one episode from the config's initial state — on this package's facade.  The policies hold the whole episode and collect the
    returns they were trained to: 236.0 / 250 for the Quadrotor2D tracking PPO model (BASELINE.md's reference reward), 214 for the SAC
    one and for Quadrotor3D PPO, 125 / 150 for CartPole.  A policy trained on Bullet's dynamics flying the restated dynamics to the
    reward it reached there is also evidence about those dynamics (physics under it in the CPU suite: the oracle)."""
    import json
    import os
    import subprocess
    import sys

    from tests.golden import ref_stubs
    if ref_stubs.reference_root() is None:
        pytest.skip('needs the reference checkout')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, 'tools', 'run_reference_example.py'), 'rl', '--algo', algo, '--system', system,
                          '--task', task, '--stub-handle'], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith('METRICS ')]
    assert line, res.stdout[-2000:]
    m = json.loads(line[0][len('METRICS '):])
    assert m['average_length'] == length and m['average_return'] >= min_return, m


def test_the_references_own_example_test_matrix_passes_on_the_facade():
    """The reference's tests/test_examples/{test_lqr,test_rl,test_pid}.py parametrisations (its end-to-end test strategy)
    — LQR / iLQR (12), PPO / SAC / Safe-Explorer PPO evaluating the shipped checkpoints (18), PID (4), each `n_steps=10` through the
    example's own run(), and test_no_controller.py's verbose_api.py on both systems (2) — with the registry's env ids bound to this package's facade (tools/run_reference_example.py matrix).  The other
    example tests need CasADi + IPOPT (mpc, cbf, mpsc) or optuna / MySQL (hpo)."""
    import os
    import subprocess
    import sys

    from tests.golden import ref_stubs
    if ref_stubs.reference_root() is None:
        pytest.skip('needs the reference checkout')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, 'tools', 'run_reference_example.py'), 'matrix', '--stub-handle'],
                         capture_output=True, text=True, timeout=600)
    lines = res.stdout.splitlines()
    assert not [ln for ln in lines if ln.startswith('FAILED')] and res.returncode == 0, (res.stdout[-3000:], res.stderr[-2000:])
    assert 'MATRIX 36 passed of 36' in lines[-1] and sum(ln.startswith('PASSED') for ln in lines) == 36
    # the checkout is read-only input: importing ~60 of its modules must not leave bytecode caches in it (ref_stubs sets sys.pycache_prefix)
    assert not [d for d, sub, _ in os.walk(ref_stubs.reference_root()) if '__pycache__' in sub]


@pytest.mark.parametrize('system', ['cartpole', 'quadrotor_2D', 'quadrotor_3D'])
def test_symbolic_constraint_forms_match_the_references_constraint_objects(system):
    """`info['symbolic_constraints']` / `env.constraints.constraints[i]`: upstream hands out its Constraint objects' `sym_func` lambdas
    (constraints.py:222, 274, 458-468).  On random task configs (tests/config_fuzz.py) plus a quadratic and a linear constraint, the facade's
    ConstraintInfo entries evaluate to the same numbers as the REFERENCE's own objects on random vectors, and carry the same `dim`,
    `num_constraints`, `strict`, `constrained_variable`, `A` / `b` / `P`."""
    import inspect

    from tests.golden import ref_stubs
    if ref_stubs.reference_root() is None:
        pytest.skip('needs the reference checkout')
    from tests.config_fuzz import fuzz_config
    from tests.golden import make_golden as MG
    from safe_control_gym_amd.env_config import EnvSpec
    rng = np.random.default_rng(3)
    n_checked = 0
    for seed in range(8):
        env_id, cfg = fuzz_config(system, seed)
        cfg = dict(cfg)
        cons = list(cfg.get('constraints') or [])
        nx = {'cartpole': 4, 'quadrotor_2D': 6, 'quadrotor_3D': 12}[system]
        if seed % 2 == 0:
            P = rng.uniform(0.1, 1.0, (2, 2)); P = (P @ P.T).tolist()
            cons += [{'constraint_form': 'quadratic_constraint', 'constrained_variable': 'state', 'P': P, 'b': 3.0, 'active_dims': [0, 2]},
                     {'constraint_form': 'linear_constraint', 'constrained_variable': 'state', 'A': rng.normal(0, 1, (3, nx)).tolist(), 'b': [1.0, 2.0, 3.0]}]
        if not cons:
            continue
        cfg['constraints'] = cons
        ref = MG.ENV_CLS[env_id](**dict(cfg, output_dir='/tmp'))
        spec = EnvSpec(env_id, dict(cfg))
        assert len(ref.constraints.constraints) == len(spec.con_meta)
        for rc, mine in zip(ref.constraints.constraints, spec.con_meta):
            assert (rc.dim, rc.strict, str(rc.constrained_variable.value), rc.decimals) == (mine.dim, mine.strict, mine.constrained_variable, mine.decimals)
            if type(rc).__name__ != 'SymmetricStateConstraint':         # (that class overrides num_constraints to n while its sym_func keeps 2n rows)
                assert rc.num_constraints == mine.num_constraints
            np.testing.assert_array_equal(rc.constraint_filter, mine.constraint_filter)
            for k in ('A', 'b', 'P'):
                assert hasattr(rc, k) == (k in mine)
                if k in mine:
                    np.testing.assert_array_equal(np.asarray(getattr(rc, k)), np.asarray(mine[k]))
            for _ in range(3):
                x = rng.normal(0, 1, mine.constraint_filter.shape[1])
                np.testing.assert_array_equal(np.asarray(rc.sym_func(x)), np.asarray(mine.get_symbolic_model()(x)))
            assert 'lambda x' in inspect.getsource(mine.sym_func)        # examples/no_controller/verbose_api.py:59-61 prints the source
            n_checked += 1
        ref.close()
    assert n_checked >= 4


@pytest.mark.parametrize('algo,system,task', [('ppo', 'cartpole', 'stab'), ('sac', 'quadrotor_2D', 'track'), ('safe_explorer_ppo', 'quadrotor_3D', 'stab')])
def test_the_references_training_script_trains_on_the_facade(algo, system, task):
    """safe_control_gym/experiments/train_rl_controller.py::train() — the script behind BASELINE config #3 — unmodified, with the arguments of
    examples/rl/train_rl_model.sh and a small budget (tools/run_reference_example.py train): the reference's own PPO / SAC / Safe-Explorer
    training loop, its make_vec_envs -> DummyVecEnv of facade envs, RecordEpisodeStatistics, buffers, updates, evaluation, logger and
    checkpointing (model_latest.pt / model_best.pt incl. `env_random_state` gathered from the facade envs) run to completion."""
    import os
    import subprocess
    import sys

    from tests.golden import ref_stubs
    if ref_stubs.reference_root() is None:
        pytest.skip('needs the reference checkout')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, 'tools', 'run_reference_example.py'), 'train', '--algo', algo, '--system', system,
                          '--task', task, '--env-steps', '800', '--stub-handle'], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and 'Training done.' in res.stdout, (res.stdout[-2000:], res.stderr[-2000:])
    line = [ln for ln in res.stdout.splitlines() if ln.startswith('TRAINED')]
    assert line and 'model_latest.pt' in line[0] and "'agent'" in line[0], res.stdout[-1500:]


@pytest.mark.parametrize('system', ['cartpole', 'quadrotor_2D'])
def test_constraint_list_api_follows_the_references_env(system, monkeypatch):
    """`env.constraints` as the reference's ConstraintList: sizes, per-kind lists, `get_values`, `get_violations`, `is_violated`,
    `is_almost_active` (with `tolerance`), per-constraint `get_value / is_violated / is_almost_active` — the facade (oracle-backed handle)
    and the REFERENCE's own env stepped with the same actions from the same deterministic start agree at every step."""
    from tests.golden import ref_stubs
    if ref_stubs.reference_root() is None:
        pytest.skip('needs the reference checkout')
    import safe_control_gym_amd.benchmark_env as B
    from tests.config_fuzz import fuzz_config
    from tests.golden import make_golden as MG
    monkeypatch.setattr(B, 'HipVecEnv', _OracleBackedVec)
    env_id, cfg = fuzz_config(system, 2)
    cfg = dict(cfg, randomized_init=False, randomized_inertial_prop=False, disturbances=None, adversary_disturbance=None, done_on_violation=False,
               done_on_out_of_bound=False, cost='rl_reward')
    nx = 4 if system == 'cartpole' else 6
    lo, hi = ([-0.4, -0.3], [0.4, 0.3]) if system == 'cartpole' else ([-0.5, 0.6], [0.5, 1.4])
    cfg['constraints'] = [
        {'constraint_form': 'bounded_constraint', 'constrained_variable': 'state', 'lower_bounds': lo, 'upper_bounds': hi, 'active_dims': [0, 2],
         'tolerance': [0.2, 0.2, 0.2, 0.2]},
        {'constraint_form': 'default_constraint', 'constrained_variable': 'input', 'strict': True},
        {'constraint_form': 'quadratic_constraint', 'constrained_variable': 'state', 'P': np.eye(nx).tolist(), 'b': 1.5, 'tolerance': [0.5]}]
    ref = MG.ENV_CLS[env_id](**dict(cfg, output_dir='/tmp', seed=5))
    mine = (B.CartPole if env_id == 'cartpole' else B.Quadrotor)(**dict(cfg, seed=5))
    rc, mc = ref.constraints, mine.constraints
    for k in ('num_constraints', 'num_state_constraints', 'num_input_constraints', 'num_input_state_constraints', 'constraint_lengths'):
        assert getattr(rc, k) == getattr(mc, k), k
    np.testing.assert_array_equal(rc.constraint_indices, mc.constraint_indices)
    assert (len(rc), len(rc.state_constraints), len(rc.input_constraints)) == (len(mc), len(mc.state_constraints), len(mc.input_constraints))
    assert len(mc.get_all_symbolic_models()) == 3 and len(mc.get_state_constraint_symbolic_models()) == 2 and len(mc.get_input_constraint_symbolic_models()) == 1
    ro, ri = ref.reset()
    mo, mi = mine.reset()
    np.testing.assert_allclose(mo, ro, atol=1e-12)
    np.testing.assert_allclose(mc.get_values(mine, only_state=True), rc.get_values(ref, only_state=True), atol=1e-8)
    rng = np.random.default_rng(0)
    seen = set()
    for t in range(60):
        a = rng.uniform(ref.action_space.low, ref.action_space.high) * (1.3 if t % 7 == 0 else 1.0)      # sometimes beyond the input bounds
        ro, rr, rd, ri = ref.step(a)
        mo, mr, md, mi = mine.step(a)
        np.testing.assert_allclose(mo, ro, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(mc.get_values(mine), rc.get_values(ref), atol=2e-8)
        assert mc.get_violations(mine) == rc.get_violations(ref) and mc.is_violated(mine) == rc.is_violated(ref)
        assert mc.is_violated(mine, c_value=mi['constraint_values']) == rc.is_violated(ref, c_value=ri['constraint_values'])
        assert mc.is_almost_active(mine) == rc.is_almost_active(ref)
        for m, r in zip(mc.constraints, rc.constraints):
            np.testing.assert_allclose(m.get_value(mine), r.get_value(ref), atol=2e-8)
            assert m.is_violated(mine) == r.is_violated(ref) and m.is_almost_active(mine) == r.is_almost_active(ref)
        seen.add((tuple(rc.get_violations(ref)), rc.is_almost_active(ref)))
        if rd:
            break
    assert len(seen) >= 2, seen                                           # the flags actually changed along the way
    ref.close(); mine.close()


def test_every_env_attribute_the_references_callers_read_exists(monkeypatch):
    """Static audit: every `env.<name>` / `self.env.<name>` that the reference's in-scope callers touch (PPO / SAC / Safe-Explorer / RARL /
    LQR / PID controllers, base_controller, experiments, env wrappers, metrics, utils) exists either on the single-env facade or — for the
    names that belong to the reference's wrapper / vec-env layer — on HipVecEnv / this package's episode-statistics wrapper."""
    import glob
    import re

    from tests.golden import ref_stubs
    root = ref_stubs.reference_root()
    if root is None:
        pytest.skip('needs the reference checkout')
    import safe_control_gym_amd.benchmark_env as B
    import safe_control_gym_amd.record_episode_statistics as R
    import safe_control_gym_amd.vec_env as V
    from tests.config_fuzz import fuzz_config
    monkeypatch.setattr(B, 'HipVecEnv', _OracleBackedVec)
    ref = os.path.join(root, 'safe_control_gym')
    files = [os.path.join(ref, 'controllers', 'base_controller.py')]
    for d in ('controllers/ppo', 'controllers/sac', 'controllers/safe_explorer', 'controllers/rarl', 'controllers/lqr', 'controllers/pid', 'experiments',
              'envs/env_wrappers', 'envs/env_wrappers/vectorized_env', 'math_and_models/metrics', 'utils'):
        files += glob.glob(os.path.join(ref, d, '*.py'))
    names = set()
    for f in files:
        names |= set(re.findall(r'\b(?:self\.)?(?:env|train_env|eval_env|test_env)\.([A-Za-z_][A-Za-z_0-9]*)', open(f).read()))
    assert len(names) >= 35
    envs = []
    for system, extra in (('cartpole', {}), ('quadrotor_2D', {'adversary_disturbance': 'dynamics'})):
        env_id, cfg = fuzz_config(system, 2)
        envs.append((B.CartPole if env_id == 'cartpole' else B.Quadrotor)(**dict(cfg, **extra)))
    wrapper_src = open(R.__file__).read() + open(V.__file__).read()
    missing = []
    for n in sorted(names):
        on_facade = hasattr(envs[1], n) and (hasattr(envs[0], n) or n in ('QUAD_TYPE', 'adversary_action_space', 'adversary_observation_space'))
        on_vec_layer = hasattr(V.HipVecEnv, n) or re.search(rf'def {n}\b|self\.{n}\b', wrapper_src)
        experiment_wrapper = n in ('data', 'clear_data', 'save_data')        # RecordDataWrapper's own members (experiments/base_experiment.py wraps the env in it)
        if not (on_facade or on_vec_layer or experiment_wrapper):
            missing.append(n)
    assert not missing, missing

"""Host logic under random task configs (tests/config_fuzz.py): EnvSpec — what the C-ABI's scg_config is filled from — against
the oracle, which is pinned to the reference's own Python (tests/test_oracle_golden.py).  No GPU: spaces, references, flattened
constraint rows and their values on random states, episode length; plus the errors both sides share with upstream."""
import numpy as np
import pytest

from oracle.envs import make_oracle_env, make_rng
from safe_control_gym_amd import _lib
from safe_control_gym_amd.env_config import EnvSpec
from tests.config_fuzz import SYSTEMS, fuzz_config


@pytest.mark.parametrize('seed', range(8))
@pytest.mark.parametrize('system', SYSTEMS)
def test_envspec_agrees_with_the_oracle_on_a_random_config(system, seed):
    env_id, cfg = fuzz_config(system, seed)
    n = 6
    o = make_oracle_env(env_id, n, make_rng('philox', n, seed), **cfg)
    spec = EnvSpec(env_id, dict(cfg))
    c, keep = spec.to_c_config(n, _lib.F64, 0)
    assert (spec.nx, spec.nu, spec.obs_dim) == (o.state_dim, o.action_dim, o.obs_dim)
    assert int(c.substeps) == o.PYB_STEPS_PER_CTRL and int(c.ctrl_steps) == int(np.ceil(o.CTRL_STEPS))     # (counter >= float CTRL_STEPS)
    np.testing.assert_allclose(np.atleast_2d(spec.X_GOAL), np.atleast_2d(o.X_GOAL), rtol=0, atol=1e-12)
    np.testing.assert_allclose(spec.U_GOAL, o.U_GOAL, rtol=1e-12)
    np.testing.assert_allclose(spec.action_space.low, o.action_space_low, rtol=1e-6)
    np.testing.assert_allclose(spec.action_space.high, o.action_space_high, rtol=1e-6)
    np.testing.assert_allclose(spec.physical_action_bounds[0], o.physical_action_bounds[0], rtol=1e-12)
    np.testing.assert_allclose(spec.physical_action_bounds[1], o.physical_action_bounds[1], rtol=1e-12)
    # constraint rows: same count, and the state rows evaluate to the oracle's values on random states
    n_rows = 0 if o.constraints is None else o.constraints.num_constraints
    assert spec.num_constraints == n_rows
    if n_rows and o.constraints.state_constraints:
        import torch
        rng = np.random.default_rng(seed)
        x = rng.uniform(-1, 1, (n, o.state_dim))
        want = o.constraints.get_values(x, None, only_state=True)
        got = spec.state_constraint_values(torch.as_tensor(x)).numpy()
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-8)


def test_short_stabilisation_goal_raises_like_upstream():
    """quadrotor.py:264-278 indexes the goal list: one entry for the 1-D / 2-D quadrotor (or two for 3-D) is an IndexError
    upstream and in the oracle — EnvSpec raised nothing and produced a NaN reference before round 3."""
    for system, goal in (('quadrotor_1D', [1.0]), ('quadrotor_2D', [0.5]), ('quadrotor_3D', [0.0, 1.0])):
        env_id, cfg = fuzz_config(system, 0)
        cfg.update(task='stabilization', task_info={'stabilization_goal': goal, 'stabilization_goal_tolerance': 0.0}, obs_goal_horizon=0,
                   disturbances=None)
        with pytest.raises(IndexError):
            make_oracle_env(env_id, 2, make_rng('philox', 2, 0), **cfg)
        with pytest.raises(IndexError):
            EnvSpec(env_id, dict(cfg))


def test_quad1d_with_a_dynamics_force_and_lateral_drift_is_refused():
    """Upstream's 1-D quadrotor carries an unobserved X velocity (init_x_dot, randomised by default); a force on the dynamics
    channel, applied at the position cached at the start of the control step (base_aviary.py:272), then pitches the drone.  The
    oracle shows it; the 1-D kernel integrates (z, z_dot) only, so EnvSpec refuses the combination instead of diverging."""
    env_id, cfg = fuzz_config('quadrotor_1D', 5)
    cfg.update(disturbances={'dynamics': [{'disturbance_func': 'step', 'magnitude': 0.07, 'step_offset': 0}]}, adversary_disturbance=None,
               randomized_init=False, init_state={'init_x': 0.3, 'init_x_dot': 0.2}, constraints=None, done_on_out_of_bound=False)
    n = 3
    o = make_oracle_env(env_id, n, make_rng('philox', n, 0), **cfg)
    o.reset()
    for _ in range(5):
        o.step(np.zeros((n, 1)))
    assert np.abs(o.rpy[:, 1]).min() > 1e-3            # the reference's drone has pitched
    with pytest.raises(NotImplementedError, match='init_x_dot'):
        EnvSpec(env_id, dict(cfg))
    with pytest.raises(NotImplementedError):            # the default randomisation table draws init_x_dot ~ U(-0.01, 0.01)
        EnvSpec(env_id, dict(cfg, randomized_init=True, init_state=None))
    EnvSpec(env_id, dict(cfg, init_state={'init_x': 0.3, 'init_x_dot': 0.0}))          # X at rest: nothing to refuse
    EnvSpec(env_id, dict(cfg, disturbances=None))                                      # no dynamics force: the drift is invisible


def test_fractional_episode_length_ends_at_the_ceiling():
    """benchmark_env.py:148,499: CTRL_STEPS = 3.5 s * 15 Hz = 52.5 stays a float upstream, `counter >= 52.5` first holds at
    step 53.  The kernels' integer limit was int(52.5) = 52 — one step early — until the config fuzz compared done flags."""
    from safe_control_gym_amd.registration import load_task
    env_id, cfg = load_task('cartpole_stab')
    cfg.update(ctrl_freq=15, pyb_freq=750, episode_len_sec=3.5, randomized_init=False, init_state=None, constraints=None,
               done_on_out_of_bound=False)
    o = make_oracle_env(env_id, 2, make_rng('philox', 2, 0), **cfg)
    o.reset()
    for t in range(53):
        _, _, done, info = o.step(np.zeros((2, 1)))
        assert bool(done[0]) == (t == 52), t
    spec = EnvSpec(env_id, dict(cfg))
    c, _ = spec.to_c_config(2, _lib.F64, 0)
    assert int(c.ctrl_steps) == 53 == spec.max_episode_steps and spec.CTRL_STEPS == 52.5

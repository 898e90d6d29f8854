"""Launch geometries of scg_step in the specialised libraries (include/scg_hip.h: scg_set_step_launch, scg_set_step_wsback) — the
wide launch (256-thread workgroups) and the write-back-workspace launch — must give bit for bit what the one-wave-per-64-envs launch gives: every output of scg_step, the episode statistics, the simulator state and
counters, across auto-resets, ragged tails and group counts that are not a multiple of the launch's packet, for every shipped
task, float32 and float64."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

CASES = [('quadrotor_2D_track', {}), ('cartpole_stab', {}), ('quadrotor_3D_track', {}), ('quadrotor_3D_track_disturbed', {}),
         ('quadrotor_2D_track', {'obs_goal_horizon': 3}), ('quadrotor_2D_track', {'episode_len_sec': 0.3}),
         ('quadrotor_2D_track', {'done_on_violation': True, 'use_constraint_penalty': True}),
         ('quadrotor_2D_track', {'randomized_inertial_prop': True, 'inertial_prop_randomization_info': {
             'M': {'distrib': 'uniform', 'low': -0.002, 'high': 0.002}, 'Iyy': {'distrib': 'uniform', 'low': -1e-6, 'high': 1e-6}}})]


NEVER = 2 ** 31 - 1
MODES = {'wide': (None, 0, (1, 0)), 'wsback': (None, NEVER, (0, NEVER))}


@pytest.mark.parametrize('mode', list(MODES))
@pytest.mark.parametrize('n', [64, 200, 1000, 1536])
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64], ids=['f32', 'f64'])
@pytest.mark.parametrize('task,over', CASES, ids=[c[0] + ('_' + '_'.join(c[1]) if c[1] else '') for c in CASES])
def test_launch_geometry_equals_single_wave_launch(task, over, dtype, n, mode):
    if n in (64, 1536) and (over or dtype == torch.float64):
        pytest.skip('geometry cases run on the plain float32 configs')
    if mode == 'wsback' and task.startswith('cartpole'):
        pytest.skip('the write-back launch serves the Quadrotor systems')
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    env_id, cfg = load_task(task)
    cfg = dict(cfg, **over)
    a, b = [HipVecEnv(env_id, n, seed=4, dtype=dtype, return_numpy=False, specialize=True, **cfg) for _ in range(2)]
    a.set_step_launch(*MODES[mode])
    b.set_step_launch(None, NEVER, (1, 0))          # one wave per 64 envs, one-wave workgroups, everything written through
    g = torch.Generator(device='cpu').manual_seed(11)
    oa, ob = a.reset_tensors(), b.reset_tensors()
    assert torch.equal(oa, ob)
    n_done = 0
    for t in range(60):
        act = (torch.rand(n, a.spec.nu, generator=g, dtype=torch.float64) * 2 - 1).to(a.device, dtype)
        x, y = a.step_tensors(act), b.step_tensors(act)
        for name in ('obs', 'reward', 'done', 'flags', 'mse', 'state', 'noisy_action', 'c_values'):
            u, v = getattr(x, name), getattr(y, name)
            if u is not None:
                assert torch.equal(u, v), (name, t)
        d = y.done.bool()
        n_done += int(d.sum())
        assert torch.equal(x.terminal_obs[d], y.terminal_obs[d]), ('terminal_obs', t)
        assert torch.equal(x.fin_stats[d], y.fin_stats[d]), ('fin_stats', t)
        assert torch.equal(a.ep_stats, b.ep_stats), ('ep_stats', t)
    assert n_done > 0
    np.testing.assert_array_equal(a.get_raw_state(), b.get_raw_state())
    for u, v in zip(a.get_counters(), b.get_counters()):
        np.testing.assert_array_equal(u, v)
    if a.spec.kw.get('randomized_inertial_prop'):
        np.testing.assert_array_equal(a.get_params(), b.get_params())
    a.close(); b.close()


def test_launch_thresholds_are_a_per_handle_setting():
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    env_id, cfg = load_task('quadrotor_2D_track')
    env = HipVecEnv(env_id, 128, seed=0, return_numpy=False, specialize=False, **cfg)     # generic library: accepted, no effect
    env.set_step_launch(4096, 0)                    # (split_max: accepted, ignored)
    env.set_step_launch(wide_min=1 << 20)           # None = unchanged
    env.reset_tensors()
    out = env.step_tensors(torch.zeros(128, 2, device=env.device))
    assert bool(torch.isfinite(out.obs).all())
    env.close()


def test_env_random_state_carries_the_rng_layout_version():
    """include/scg_hip.h: SCG_RNG_LAYOUT_VERSION — which bits of which Philox block feed which reset draw.  Checkpoints carry it; a state
    saved under another layout (or before the field existed: rounds 1-4, layout 1) loads with a warning, not silently."""
    import warnings
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    env_id, cfg = load_task('quadrotor_2D_track')
    env = HipVecEnv(env_id, 128, seed=3, return_numpy=False, **cfg)
    env.reset_tensors()
    st = env.get_env_random_state()
    assert st[0]['rng_layout_version'] == 2 == int(env._lib.scg_rng_layout_version())
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        env.set_env_random_state(st)                        # same layout: silent
    old = [dict(st[0])]
    old[0].pop('rng_layout_version')
    with pytest.warns(UserWarning, match='Philox word layout 1'):
        env.set_env_random_state(old)
    env.close()

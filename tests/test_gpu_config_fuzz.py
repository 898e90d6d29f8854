"""Differential test over RANDOM task configs (tests/config_fuzz.py): the float64 HIP kernels of the generic library (any
config, parameters staged in LDS) free-running against the oracle — 70 envs (a full wave and a partial one), 60 control steps of
random actions with auto-resets, random adversary actions where the config has the channel.  The 16 golden rollouts
(test_gpu_env_parity.py) are hand-picked combinations; this walks the YAML surface: substep counts 4-50 (incl. the
large-rotation path of the planar integrators at 200-250 Hz engines), both costs, goal horizons, every constraint form, every
disturbance kind per channel, uniform / normal / choice randomisation tables, projected 3-D references.  Integer / bool outputs
exactly, floating point to rtol 1e-7 / atol 2e-9 (free-running; test_gpu_env_parity.py holds the golden rollouts to 1e-9)."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from tests.config_fuzz import SYSTEMS, fuzz_config                      # noqa: E402
from tests.test_gpu_env_parity import _flags, _np, _params, _raw_state  # noqa: E402

N_ENVS, N_STEPS = 70, 60
SPEC_SEEDS = range(4)           # configs whose SPECIALISED libraries tools/prebuild_specs.py compiles ahead (both dtypes)


def _pair(system, seed, dtype, specialize):
    from oracle.envs import make_oracle_env, make_rng
    from oracle.vec import OracleVecEnv
    from safe_control_gym_amd.vec_env import HipVecEnv
    env_id, cfg = fuzz_config(system, seed)
    n = N_ENVS
    oracle = make_oracle_env(env_id, n, make_rng('philox', n, 100 + seed), **cfg)
    gpu = HipVecEnv(env_id, n, seed=100 + seed, dtype=dtype, return_numpy=False, specialize=specialize, **cfg)
    assert gpu.specialized == bool(specialize)
    return oracle, OracleVecEnv(oracle), gpu


def _draw_actions(rng, oracle, gpu):
    """Random actions reaching past the action space now and then (clipping); random adversary actions where configured."""
    n = oracle.num_envs
    act = rng.uniform(-1.2, 1.2, (n, oracle.action_dim))
    if not oracle.NORMALIZED_RL_ACTION_SPACE:
        lo, hi = oracle.physical_action_bounds
        act = lo + (act + 1.2) / 2.4 * (hi - lo) * 1.1 - 0.05 * (hi - lo)
    adv = None
    if oracle.adversary_disturbance is not None:
        a = rng.uniform(-1.3, 1.3, (n, oracle.adversary_dim))
        oracle.set_adversary_control(a)
        gpu.set_adversary_control(a)
        adv = gpu._adv
        gpu._adv = None
    return act, adv


def _free_running_f64(system, seed, specialize):
    oracle, ovec, gpu = _pair(system, seed, torch.float64, specialize)
    n = oracle.num_envs
    # (free-running, up to 50 engine substeps per step with randomised rates and disturbances: rounding differences of the two
    #  float64 programs grow to ~3e-9 within 30 steps of the 15 Hz / 750 Hz 3-D case; semantic differences show up at >= 1e-7)
    tol = dict(rtol=1e-7, atol=2e-9)
    obs_o, info_o = ovec.reset()
    obs_g = gpu.reset_tensors()
    np.testing.assert_allclose(_np(obs_g), obs_o, **tol)
    np.testing.assert_allclose(gpu.get_raw_state(), _raw_state(oracle), **tol)
    rng = np.random.default_rng(seed)
    n_done, sd_prev = 0, 0.0
    for t in range(N_STEPS):
        msg = f'{system} seed={seed} t={t}'
        act, adv = _draw_actions(rng, oracle, gpu)
        obs_o, rew_o, done_o, info = ovec.step(act)
        out = gpu.step_tensors(torch.as_tensor(act, dtype=torch.float64, device=gpu.device), adv)
        # every output of the step is compared before anything is asserted, so that a failure names ALL the fields that moved
        mask = 0x0F if 'out_of_bounds' in info else 0x03
        d = np.nonzero(done_o)[0]
        n_done += len(d)
        checks = [('state', _np(out.state).T, oracle.state), ('obs', _np(out.obs), obs_o), ('reward', _np(out.reward), rew_o),
                  ('mse', _np(out.mse), info['mse'])]
        if len(d):
            checks.append(('terminal_obs', _np(out.terminal_obs)[d], info['terminal_observation'][d]))
        bad = []
        for name, got, want in checks:
            err = np.abs(got - want) - (tol['atol'] + tol['rtol'] * np.abs(want))
            if not np.all(np.isfinite(got)) or err.max() > 0:
                bad.append(f'{name}: max |delta| {np.nanmax(np.abs(got - want)):.3e} (rel {np.nanmax(np.abs(got - want) / (np.abs(want) + 1e-300)):.2e})')
        if 'constraint_values' in info:
            cv = np.abs(_np(out.c_values).T - info['constraint_values']).max()
            # values are rounded to 8 decimals (constraints.py:109) and a row is at most a 12-term combination with |A_ij| <= 1:
            # the free-running state delta (<= 3e-9 late in the chaotic 3-D cases) may move a row by a few units of 1e-8
            # (on a step that ends the episode the values are the terminal ones while `state` is already the next episode's:
            #  the delta of the step before, grown by one step, stands in)
            sd_now = np.abs(_np(out.state).T - oracle.state).max()
            sd, sd_prev = max(sd_now, 4.0 * sd_prev), sd_now
            if not cv <= 2.1e-8 + 12.0 * sd:
                bad.append(f'constraint_values: max |delta| {cv:.3e}')
        if not np.array_equal(_np(out.done).astype(bool), done_o):
            bad.append(f'done: {int((_np(out.done).astype(bool) != done_o).sum())} envs differ')
        if not np.array_equal(_np(out.flags).astype(np.uint8) & mask, _flags(info) & mask):
            bad.append('flags differ')
        assert not bad, msg + ' -> ' + '; '.join(bad)
    step, ep = gpu.get_counters()
    np.testing.assert_array_equal(step, oracle.ctrl_step_counter)
    np.testing.assert_array_equal(ep.astype(np.int64), oracle.episode)
    if oracle.RANDOMIZED_INERTIAL_PROP:
        np.testing.assert_allclose(gpu.get_params(), _params(oracle), rtol=1e-12)
    gpu.close()
    return n_done


@pytest.mark.parametrize('seed', range(16))
@pytest.mark.parametrize('system', SYSTEMS)
def test_f64_generic_kernels_vs_oracle_on_a_random_config(system, seed):
    _free_running_f64(system, seed, specialize=False)


@pytest.mark.parametrize('seed', SPEC_SEEDS)
@pytest.mark.parametrize('system', SYSTEMS)
def test_f64_specialised_kernels_vs_oracle_on_a_random_config(system, seed):
    """The build the product runs: the same sources compiled with THIS config as constants (every `if constexpr` family of
    scg_env_core.h is decided by the config: disturbance lists unrolled, reset draws folded into the integrator, row paths)."""
    _free_running_f64(system, seed, specialize=True)


def _one_step_f32(system, seed, specialize):
    """Production dtype, protocol of test_gpu_env_parity.py::test_f32_kernels_one_step_error_vs_oracle: state, counters and
    parameters re-synchronised to the oracle before every step, one-step errors at float32 round-off level."""
    oracle, ovec, gpu = _pair(system, seed, torch.float32, specialize)
    ovec.reset()
    gpu.reset_tensors()
    rng = np.random.default_rng(seed)
    bad_flags, total = 0, 0
    for t in range(40):
        gpu.set_raw_state(_raw_state(oracle))
        gpu.set_counters(oracle.ctrl_step_counter, oracle.episode)
        if oracle.RANDOMIZED_INERTIAL_PROP:
            gpu.set_params(_params(oracle))
        act, adv = _draw_actions(rng, oracle, gpu)
        obs_o, rew_o, done_o, info = ovec.step(act)
        out = gpu.step_tensors(torch.as_tensor(act, dtype=torch.float32, device=gpu.device), adv)
        msg = f'{system} seed={seed} t={t}'
        same = _np(out.done).astype(bool) == done_o
        bad_flags += int((~same).sum())
        total += same.size
        scale = np.maximum(1.0, np.abs(oracle.state).max())
        keep = same & ~done_o          # post-reset rows differ when the done decision differs
        np.testing.assert_allclose(_np(out.state).T[keep], oracle.state[keep], rtol=1e-4, atol=2e-5 * scale, err_msg=msg)
        np.testing.assert_allclose(_np(out.obs)[keep], obs_o[keep], rtol=1e-4, atol=2e-5 * scale, err_msg=msg)
        rs = np.maximum(1.0, np.abs(rew_o).max())
        np.testing.assert_allclose(_np(out.reward), rew_o, rtol=3e-4, atol=3e-5 * rs, err_msg=msg)
        np.testing.assert_allclose(_np(out.mse), info['mse'], rtol=3e-4, atol=3e-5 * scale * scale, err_msg=msg)
        if 'constraint_values' in info:
            np.testing.assert_allclose(_np(out.c_values).T, info['constraint_values'], rtol=1e-4, atol=5e-5 * scale, err_msg=msg)
    assert bad_flags <= max(2, total // 100), f'{bad_flags}/{total} done flags differ'
    gpu.close()


@pytest.mark.parametrize('seed', range(16))
@pytest.mark.parametrize('system', SYSTEMS)
def test_f32_generic_kernels_one_step_error_on_a_random_config(system, seed):
    _one_step_f32(system, seed, specialize=False)


@pytest.mark.parametrize('seed', SPEC_SEEDS)
@pytest.mark.parametrize('system', SYSTEMS)
def test_f32_specialised_kernels_one_step_error_on_a_random_config(system, seed):
    _one_step_f32(system, seed, specialize=True)


def _sequence_equals_steps(system, seed, dtype, specialize, K=12):
    """scg_step_sequence on a random config: K control steps in one launch == K launches of scg_step, bit for bit (every
    per-step output, episode statistics, state and counters afterwards)."""
    from safe_control_gym_amd.vec_env import HipVecEnv
    env_id, cfg = fuzz_config(system, seed)
    n = N_ENVS
    a, b = (HipVecEnv(env_id, n, seed=7 + seed, dtype=dtype, return_numpy=False, specialize=specialize, **cfg) for _ in range(2))
    g = torch.Generator(device='cpu').manual_seed(seed)
    lo = torch.as_tensor(a.spec.action_space.low, dtype=torch.float64)
    hi = torch.as_tensor(a.spec.action_space.high, dtype=torch.float64)
    acts = (lo + (hi - lo) * (torch.rand(K, n, a.spec.nu, generator=g, dtype=torch.float64) * 1.2 - 0.1)).to(a.device, dtype).contiguous()
    adv = None
    if a.spec.adversary_disturbance is not None:
        adv = ((torch.rand(K, n, a.spec.adversary_dim, generator=g, dtype=torch.float64) * 2 - 1) * 0.03).to(a.device, dtype).contiguous()
    a.reset_tensors(); b.reset_tensors()
    seq = a.step_sequence(acts, adv_actions=adv, terminal_obs=True, mse=True, c_values=True, fin_stats=True, state=True, noisy_action=True)
    for t in range(K):
        out = b.step_tensors(acts[t], None if adv is None else adv[t])
        for name, got, ref in (('obs', seq['obs'][t], out.obs), ('reward', seq['reward'][t], out.reward), ('done', seq['done'][t], out.done),
                               ('flags', seq['flags'][t], out.flags), ('mse', seq['mse'][t], out.mse), ('state', seq['state'][t], out.state),
                               ('noisy_action', seq['noisy_action'][t], out.noisy_action)):
            assert torch.equal(got, ref), (system, seed, name, t)
        if 'c_values' in seq:
            assert torch.equal(seq['c_values'][t], out.c_values), (system, seed, 'c_values', t)
        d = out.done.bool()
        assert torch.equal(seq['terminal_obs'][t][d], out.terminal_obs[d]), (system, seed, 'terminal_obs', t)
        assert torch.equal(seq['fin_stats'][t][d], out.fin_stats[d]), (system, seed, 'fin_stats', t)
    assert torch.equal(a.ep_stats, b.ep_stats)
    np.testing.assert_array_equal(a.get_raw_state(), b.get_raw_state())
    for x, y in zip(a.get_counters(), b.get_counters()):
        np.testing.assert_array_equal(x, y)
    a.close(); b.close()


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64], ids=['f32', 'f64'])
@pytest.mark.parametrize('seed', range(8))
@pytest.mark.parametrize('system', SYSTEMS)
def test_sequence_equals_repeated_steps_on_a_random_config_generic(system, seed, dtype):
    _sequence_equals_steps(system, seed, dtype, specialize=False)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64], ids=['f32', 'f64'])
@pytest.mark.parametrize('seed', SPEC_SEEDS)
@pytest.mark.parametrize('system', SYSTEMS)
def test_sequence_equals_repeated_steps_on_a_random_config_specialised(system, seed, dtype):
    _sequence_equals_steps(system, seed, dtype, specialize=True)


@pytest.mark.parametrize('seed', range(8))
@pytest.mark.parametrize('system', SYSTEMS)
def test_single_env_facade_vs_oracle_on_a_random_config(system, seed):
    """BenchmarkEnv.reset() / step() of the reference (single env, the CALLER resets: benchmark_env.py:320-359, 400-502) through
    the facade on a batch-of-1 handle (`auto_reset` off — another branch of the step kernel) against the oracle's un-vectorised
    step: observations, rewards, done, and the info dict — the same KEYS (TimeLimit.truncated only once the time is up,
    out_of_bounds / goal_reached only where upstream defines them) and values — and the action attributes controllers read."""
    from safe_control_gym_amd.registration import make
    facade_vs_oracle(system, seed, make)


def facade_vs_oracle(system, seed, make):
    """Body of the facade test; `make(env_id, seed=..., **cfg)` builds the facade (tests/test_facade_cpu.py runs it on a stub handle
    backed by a second oracle instance: the facade's host logic in the CPU suite)."""
    from oracle.envs import make_oracle_env, make_rng
    env_id, cfg = fuzz_config(system, seed)
    env = make(env_id, seed=31 + seed, **cfg)
    o = make_oracle_env(env_id, 1, make_rng('philox', 1, 31 + seed), **cfg)
    tol = dict(rtol=1e-7, atol=2e-9)
    rng = np.random.default_rng(seed)
    obs_o, info_o = o.reset()
    obs, info = env.reset()
    np.testing.assert_allclose(obs, obs_o[0], **tol)
    assert info['current_step'] == 0 and ('constraint_values' in info) == ('constraint_values' in info_o)
    if 'constraint_values' in info_o:
        np.testing.assert_allclose(info['constraint_values'], info_o['constraint_values'][0], rtol=0, atol=3e-8)
    episodes = 0
    for t in range(90):
        msg = f'{system} seed={seed} t={t}'
        act = rng.uniform(-1.1, 1.1, o.action_dim)
        if not o.NORMALIZED_RL_ACTION_SPACE:
            lo, hi = o.physical_action_bounds
            act = lo + (act + 1.1) / 2.2 * (hi - lo)
        if o.adversary_disturbance is not None:
            a = rng.uniform(-1.2, 1.2, o.adversary_dim)
            o.set_adversary_control(a[None])
            env.set_adversary_control(a)
        obs_o, rew_o, done_o, info_o = o.step(act[None])
        obs, rew, done, info = env.step(act)
        assert done == bool(done_o[0]), msg
        np.testing.assert_allclose(obs, obs_o[0], err_msg=msg, **tol)
        np.testing.assert_allclose(rew, rew_o[0], err_msg=msg, **tol)
        np.testing.assert_allclose(env.state, o.state[0], err_msg=msg, **tol)
        want = {'current_step', 'constraint_violation', 'mse'}
        want |= {k for k in ('constraint_values', 'out_of_bounds', 'goal_reached') if k in info_o}
        if info_o['time_limit_reached'][0]:
            want.add('TimeLimit.truncated')
        assert set(info) == want, (msg, sorted(info), sorted(want))
        assert info['current_step'] == int(info_o['current_step'][0]) == env.ctrl_step_counter, msg
        assert info['constraint_violation'] == int(info_o['constraint_violation'][0]), msg
        np.testing.assert_allclose(info['mse'], info_o['mse'][0], err_msg=msg, **tol)
        for k in ('out_of_bounds', 'goal_reached', 'TimeLimit.truncated'):
            if k in want:
                assert info[k] == bool(info_o[k][0]), (msg, k)
        if 'constraint_values' in want:
            np.testing.assert_allclose(info['constraint_values'], info_o['constraint_values'][0], rtol=0, atol=5e-8, err_msg=msg)
        np.testing.assert_allclose(env.current_physical_action, o.current_physical_action[0], err_msg=msg, **tol)
        np.testing.assert_allclose(env.current_noisy_physical_action, o.current_noisy_physical_action[0], err_msg=msg, **tol)
        np.testing.assert_allclose(env.current_clipped_action, o.current_clipped_action[0], err_msg=msg, **tol)
        if done:
            episodes += 1
            obs_o, _ = o.reset()
            obs, _ = env.reset()
            np.testing.assert_allclose(obs, obs_o[0], err_msg=msg + ' (reset)', **tol)
    env.close()


@pytest.mark.parametrize('seed', range(4))
@pytest.mark.parametrize('system', SYSTEMS)
def test_reference_vec_api_info_dicts_on_a_random_config(system, seed):
    """The reference's own calling convention — NumPy in / out, `info['n'][i]` one dict per env (dummy_vec_env.py:24-41) — on a
    random config against the oracle: the KEY SETS of running envs (benchmark_env.py:447-502: current_step always, out_of_bounds /
    goal_reached / constraint_values only where upstream defines them) and of auto-reset envs (the new episode's reset info +
    terminal_observation + terminal_info with TimeLimit.truncated only once the time is up), and every value."""
    from oracle.envs import make_oracle_env, make_rng
    from oracle.vec import OracleVecEnv
    from safe_control_gym_amd.vec_env import HipVecEnv
    env_id, cfg = fuzz_config(system, seed)
    n = 9
    oracle = make_oracle_env(env_id, n, make_rng('philox', n, 55 + seed), **cfg)
    ovec = OracleVecEnv(oracle)
    env = HipVecEnv(env_id, n, seed=55 + seed, dtype=torch.float64, **cfg)             # return_numpy=True: the reference API
    tol = dict(rtol=1e-7, atol=2e-9)
    obs_o, info_o = ovec.reset()
    obs, info = env.reset()
    np.testing.assert_allclose(obs, obs_o, **tol)
    for i in range(n):
        reset_keys = {'current_step', 'x_reference', 'u_reference', 'physical_parameters'}
        assert set(info['n'][i]) == reset_keys | ({'constraint_values'} if 'constraint_values' in info_o else set())
    rng = np.random.default_rng(seed)
    ret_run = np.zeros(n)
    seen_done = 0
    for t in range(70):
        act, adv = _draw_actions(rng, oracle, env)
        env._adv = adv
        obs_o, rew_o, done_o, io = ovec.step(act)
        obs, rew, done, info = env.step(act)
        msg = f'{system} seed={seed} t={t}'
        np.testing.assert_array_equal(done, done_o, err_msg=msg)
        np.testing.assert_allclose(obs, obs_o, err_msg=msg, **tol)
        np.testing.assert_allclose(rew, rew_o, err_msg=msg, **tol)
        ret_run += rew_o
        step_keys = {'current_step', 'constraint_violation', 'mse'} | {k for k in ('constraint_values', 'out_of_bounds', 'goal_reached') if k in io}
        for i in range(n):
            inf = info['n'][i]
            st = inf
            if done_o[i]:
                seen_done += 1
                want = {'current_step', 'x_reference', 'u_reference', 'physical_parameters', 'terminal_observation', 'terminal_info', 'episode'}
                if oracle.constraints is not None and oracle.constraints.state_constraints:
                    want.add('constraint_values')
                assert set(inf) == want, (msg, i, sorted(inf))
                assert inf['current_step'] == 0
                np.testing.assert_allclose(inf['terminal_observation'], io['terminal_observation'][i], err_msg=msg, **tol)
                if 'constraint_values' in want:            # the NEW episode's state constraints (after_reset)
                    fresh = oracle.constraints.get_values(oracle.state[i:i + 1], None, only_state=True)[0]
                    np.testing.assert_allclose(inf['constraint_values'], fresh, rtol=0, atol=3e-8, err_msg=msg)
                np.testing.assert_allclose(inf['episode']['r'], ret_run[i], rtol=1e-6, atol=1e-8, err_msg=msg)
                assert inf['episode']['l'] == float(io['current_step'][i])
                ret_run[i] = 0.0
                st = inf['terminal_info']
            want = set(step_keys) | ({'TimeLimit.truncated'} if io['time_limit_reached'][i] else set())
            # ('ground_contact' is this package's documented extension key, raised only while the body is at the reference world's
            #  unmodelled ground plane — DESIGN 3, test_gpu_env_parity.py::test_ground_plane_flag_marks_the_unmodelled_contact)
            assert set(st) - {'ground_contact'} == want, (msg, i, sorted(st), sorted(want))
            assert st['current_step'] == int(io['current_step'][i]), (msg, i)
            assert st['constraint_violation'] == int(io['constraint_violation'][i]), (msg, i)
            np.testing.assert_allclose(st['mse'], io['mse'][i], err_msg=msg, **tol)
            for k in ('out_of_bounds', 'goal_reached'):
                if k in want:
                    assert st[k] == bool(io[k][i]), (msg, i, k)
            if 'TimeLimit.truncated' in want:
                assert st['TimeLimit.truncated'] == bool(io['truncated'][i]), (msg, i)
            if 'constraint_values' in want:
                np.testing.assert_allclose(st['constraint_values'], io['constraint_values'][i], rtol=0, atol=5e-8, err_msg=msg)
    env.close()


@pytest.mark.parametrize('seed', range(6))
@pytest.mark.parametrize('system', SYSTEMS)
def test_partial_reset_and_random_state_round_trip_on_a_random_config(system, seed):
    """scg_reset with a mask (the envs a caller resets itself) against the oracle's reset(idx) — other envs untouched, the reset
    ones on the next episode's draws — and get / set_env_random_state (dummy_vec_env.py:68-74): replaying the same actions from a
    restored snapshot reproduces every output bit for bit (disturbance offsets, randomised parameters, episode counters)."""
    oracle, ovec, gpu = _pair(system, seed, torch.float64, False)
    n = oracle.num_envs
    tol = dict(rtol=1e-7, atol=2e-9)
    ovec.reset()
    gpu.reset_tensors()
    rng = np.random.default_rng(1000 + seed)
    for t in range(6):
        act, adv = _draw_actions(rng, oracle, gpu)
        ovec.step(act)
        gpu.step_tensors(torch.as_tensor(act, dtype=torch.float64, device=gpu.device), adv)
    mask = rng.random(n) < 0.35
    mask[0] = True
    idx = np.nonzero(mask)[0]
    obs_o, _ = oracle.reset(idx)
    obs_g = _np(gpu.reset_tensors(mask.astype(np.uint8)))
    np.testing.assert_allclose(obs_g[idx], obs_o, **tol)
    raw_g, raw_o = gpu.get_raw_state(), _raw_state(oracle)
    if system == 'quadrotor_2D':        # the kernel's pitch accumulates, the oracle's quaternion knows it modulo 4 pi (tumbling drones)
        raw_g[:, 4] = raw_o[:, 4] + np.remainder(raw_g[:, 4] - raw_o[:, 4] + 2 * np.pi, 4 * np.pi) - 2 * np.pi
    np.testing.assert_allclose(raw_g, raw_o, **tol)
    step, ep = gpu.get_counters()
    np.testing.assert_array_equal(step, oracle.ctrl_step_counter)
    np.testing.assert_array_equal(ep.astype(np.int64), oracle.episode)
    snap = gpu.get_env_random_state()
    acts = [_draw_actions(rng, oracle, gpu) for _ in range(10)]
    runs = []
    for rep in range(2):
        outs = []
        for act, adv in acts:
            out = gpu.step_tensors(torch.as_tensor(act, dtype=torch.float64, device=gpu.device), adv)
            d = out.done.bool()                          # (fin_stats / terminal_obs rows are written where done only)
            outs.append([getattr(out, k).clone() for k in ('obs', 'reward', 'done', 'flags', 'state', 'mse')] +
                        [out.fin_stats[d].clone(), out.terminal_obs[d].clone()])
        runs.append((outs, gpu.get_raw_state(), gpu.get_counters(), gpu.ep_stats.clone()))
        if rep == 0:
            gpu.set_env_random_state(snap)
    for a, b in zip(runs[0][0], runs[1][0]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    np.testing.assert_array_equal(runs[0][1], runs[1][1])
    for x, y in zip(runs[0][2], runs[1][2]):
        np.testing.assert_array_equal(x, y)
    assert torch.equal(runs[0][3], runs[1][3])
    gpu.close()


@pytest.mark.parametrize('seed', range(6))
@pytest.mark.parametrize('system', SYSTEMS)
def test_rollout_random_on_a_random_config(system, seed):
    """scg_rollout_random (BASELINE config #2's loop: K control steps per launch, actions ~ U(-1, 1) drawn in the kernel, state in
    registers between steps) == K oracle steps fed the same Philox actions: reward sums, done / violation counts, last observation,
    final simulator state and counters."""
    from oracle.envs import make_oracle_env, make_rng
    from oracle.rng import CH_RANDOM_ACTION, make_tag
    from oracle.vec import OracleVecEnv
    from safe_control_gym_amd.vec_env import HipVecEnv
    env_id, cfg = fuzz_config(system, seed)
    cfg['adversary_disturbance'] = None                 # (the loop has no adversary input)
    n, K = N_ENVS, 45
    oracle = make_oracle_env(env_id, n, make_rng('philox', n, 300 + seed), **cfg)
    ovec = OracleVecEnv(oracle)
    gpu = HipVecEnv(env_id, n, seed=300 + seed, dtype=torch.float64, return_numpy=False, specialize=False, **cfg)
    ovec.reset()
    gpu.reset_tensors()
    idx = np.arange(n)
    rsum, dcount, vcount = np.zeros(n), np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    tag = make_tag(CH_RANDOM_ACTION, 0, 0)
    for _ in range(K):
        w = oracle.rng.words(idx, oracle.episode, oracle.ctrl_step_counter, tag)
        act = -1.0 + 2.0 * (((w[:, :oracle.action_dim] >> np.uint32(8)).astype(np.float64) + 0.5) / 16777216.0)
        obs, rew, done, info = ovec.step(act)
        rsum += rew
        dcount += done
        vcount += info['constraint_violation']
    r, d, v, last = gpu.rollout_random(K)
    np.testing.assert_array_equal(_np(d), dcount)
    np.testing.assert_array_equal(_np(v), vcount)
    np.testing.assert_allclose(_np(r), rsum, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(_np(last), obs, rtol=1e-7, atol=2e-9)
    raw_g, raw_o = gpu.get_raw_state(), _raw_state(oracle)
    if system == 'quadrotor_2D':
        raw_g[:, 4] = raw_o[:, 4] + np.remainder(raw_g[:, 4] - raw_o[:, 4] + 2 * np.pi, 4 * np.pi) - 2 * np.pi
    np.testing.assert_allclose(raw_g, raw_o, rtol=1e-7, atol=2e-9)
    step, ep = gpu.get_counters()
    np.testing.assert_array_equal(step, oracle.ctrl_step_counter)
    np.testing.assert_array_equal(ep.astype(np.int64), oracle.episode)
    gpu.close()

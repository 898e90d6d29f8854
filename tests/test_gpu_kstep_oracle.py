"""The K-steps-per-launch kernels DIRECTLY against the CPU oracle (not against scg_step): the loops they replace are
`for t in range(K): vec_env.step(actions[t])` (dummy_vec_env.py:24-41) and PPO's collector (controllers/ppo/ppo.py:259-303).

* scg_step_sequence (K = 8, every output of scg_step), float64, 250 control steps of quadrotor_2D_track from Philox-randomised
  initial states through auto-resets: every per-step output == OracleVecEnv on the same seed and actions at 1e-9 (integers /
  booleans exactly), on the generic and the specialised library;
* scg_rollout_policy (the shipped Quadrotor2D policy in the loop, float32 — the kernel has no float64 build), 250 control steps:
  the oracle is driven with the ACTIONS THE KERNEL RECORDED from the STATES THE KERNEL RECORDED (re-synchronised every step, like
  the float32 one-step tests of scg_step): every control step, terminal observation and in-launch auto-reset of the simulator
  inside the rollout kernel against the float64 oracle at the one-step float32 tolerance (2e-5), and the recorded action must
  be the actor's mean on the recorded observation.  (The closed-loop 1000-step bar of north_star is tests/test_gpu_parity_scale.py's.)"""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64)


@pytest.mark.parametrize('specialize', [False, True], ids=['generic', 'specialised'])
def test_step_sequence_f64_free_running_vs_oracle(specialize):
    from oracle.envs import make_oracle_env, make_rng
    from oracle.vec import OracleEpisodeStats, OracleVecEnv
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    from tests.test_gpu_env_parity import _flags, _raw_state
    env_id, cfg = load_task('quadrotor_2D_track')
    n, K, launches, seed = 200, 8, 32, 21                 # 256 control steps >= 250; three full waves + a ragged one
    oracle = make_oracle_env(env_id, n, make_rng('philox', n, seed), **cfg)
    ovec = OracleVecEnv(oracle)
    gpu = HipVecEnv(env_id, n, seed=seed, dtype=torch.float64, return_numpy=False, specialize=specialize, **cfg)
    assert gpu.specialized == bool(specialize)
    tol = dict(rtol=1e-9, atol=1e-10)
    obs_o, _ = ovec.reset()
    np.testing.assert_allclose(_np(gpu.reset_tensors()), obs_o, **tol)
    rs = np.random.RandomState(5)
    stats = OracleEpisodeStats(n)
    n_done = 0
    for launch in range(launches):
        acts = rs.uniform(-1, 1, (K, n, 2))
        if launch % 4 == 3:
            acts *= 0.1                                    # gentle phases: some episodes run to the time limit
        seq = gpu.step_sequence(torch.as_tensor(acts, dtype=torch.float64, device=gpu.device), terminal_obs=True, mse=True, c_values=True,
                                fin_stats=True, state=True, noisy_action=True)
        for t in range(K):
            obs_o, rew_o, done_o, info = ovec.step(acts[t])
            msg = f'launch {launch} t={t}'
            np.testing.assert_array_equal(_np(seq['done'][t]).astype(bool), done_o, err_msg=msg)
            np.testing.assert_array_equal(_np(seq['flags'][t]).astype(np.uint8) & 0x0F, _flags(info) & 0x0F, err_msg=msg)
            np.testing.assert_allclose(_np(seq['reward'][t]), rew_o, err_msg=msg, **tol)
            np.testing.assert_allclose(_np(seq['obs'][t]), obs_o, err_msg=msg, **tol)
            np.testing.assert_allclose(_np(seq['mse'][t]), info['mse'], err_msg=msg, **tol)
            np.testing.assert_allclose(_np(seq['state'][t]).T, oracle.state, err_msg=msg, **tol)
            np.testing.assert_allclose(_np(seq['c_values'][t]).T, info['constraint_values'], rtol=0, atol=2e-8, err_msg=msg)
            d = np.nonzero(done_o)[0]
            n_done += len(d)
            finished = stats.update(rew_o, done_o, info)         # VecRecordEpisodeStatistics (record_episode_statistics.py:139-166)
            if len(d):
                np.testing.assert_allclose(_np(seq['terminal_obs'][t])[d], info['terminal_observation'][d], err_msg=msg, **tol)
                fin = _np(seq['fin_stats'][t])
                for i, ep in finished:
                    np.testing.assert_allclose(fin[i, 0], ep['r'], err_msg=msg, rtol=1e-9, atol=1e-9)
                    assert fin[i, 1] == ep['l'], msg
    assert n_done > n                                      # every env finished at least one episode on average
    np.testing.assert_allclose(gpu.get_raw_state(), _raw_state(oracle), **tol)
    step, ep = gpu.get_counters()
    np.testing.assert_array_equal(step, oracle.ctrl_step_counter)
    np.testing.assert_array_equal(ep.astype(np.int64), oracle.episode)
    gpu.close()


def test_rollout_policy_simulator_vs_oracle_on_the_recorded_actions():
    from oracle.envs import make_oracle_env, make_rng
    from oracle.vec import OracleVecEnv
    from safe_control_gym_amd.ppo import PPO, PPOConfig
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    env_id, cfg = load_task('quadrotor_2D_track')
    n, K, seed = 256, 250, 13
    pol = np.load(os.path.join(GOLDEN, 'policies.npz'))
    hidden = int(pol['quadrotor_2D_track/actor.pi_net.fcs.0.weight'].shape[0])
    env = HipVecEnv(env_id, n, seed=seed, return_numpy=False, policy=(hidden, 'tanh'), **cfg)
    pcfg = PPOConfig(hidden_dim=hidden, activation='tanh', use_gae=True, rollout_batch_size=n, rollout_steps=8, mini_batch_size=n * 4, opt_epochs=1)
    ppo = PPO(env, pcfg, seed=0)
    _load_shipped_policy(ppo, pol, 'quadrotor_2D_track')
    assert ppo._fused_rollout
    nobs, nu = env.spec.obs_dim, env.spec.nu
    f = dict(device=env.device, dtype=torch.float32)
    obs, actb, logp, rew = torch.zeros(K + 1, n, nobs, **f), torch.zeros(K, n, nu, **f), torch.zeros(K, n, **f), torch.zeros(K, n, **f)
    done, flags = torch.zeros(K, n, dtype=torch.uint8, device=env.device), torch.zeros(K, n, dtype=torch.uint8, device=env.device)
    term = torch.zeros(K, n, nobs, **f)
    env.seed(seed); env.reset_tensors()
    env.rollout_policy(ppo._policy_struct(True), K, obs, actb, logp, rew, done, flags, terminal_obs=term)
    torch.cuda.synchronize()
    # the recorded action is the actor's mean on the recorded observation
    with torch.no_grad():
        a_ref = ppo.agent.ac.act(obs[:K].reshape(K * n, nobs)).reshape(K, n, nu)
    torch.testing.assert_close(actb, a_ref, rtol=1e-4, atol=2e-5)
    oracle = make_oracle_env(env_id, n, make_rng('philox', n, seed), **cfg)
    ovec = OracleVecEnv(oracle)
    ovec.reset()                                            # (PPO's constructor had reset `env` once already: same episode indices)
    obs_o, _ = ovec.reset()
    O, A, D, T = _np(obs), _np(actb), _np(done).astype(bool), _np(term)
    np.testing.assert_allclose(O[0], obs_o, rtol=2e-6, atol=2e-6)          # the fresh episodes' Philox draws
    idx = np.arange(n)
    worst, scale, worst_r, n_done = np.zeros(nobs), np.ones(nobs), 0.0, 0
    alive = np.ones(n, dtype=bool)
    for t in range(K):
        # re-synchronise: the oracle continues from the state the KERNEL recorded for step t (observation = state | goal row), so
        # that what is compared is one control step of the simulator inside the rollout kernel, not an open-loop replay of an
        # unstable plant (whose float32 rounding differences grow exponentially without the feedback)
        x = O[t][:, :6]
        zero = np.zeros(n)
        quat = np.stack([zero, np.sin(0.5 * x[:, 4]), zero, np.cos(0.5 * x[:, 4])], axis=1)
        oracle.set_body_state(idx, np.stack([x[:, 0], zero, x[:, 2]], axis=1), quat, np.stack([x[:, 1], zero, x[:, 3]], axis=1),
                              np.stack([zero, x[:, 5], zero], axis=1))
        obs_o, rew_o, done_o, info = ovec.step(A[t])
        alive &= (D[t] == done_o)           # an episode boundary crossed on one side only (a bound within float32 rounding): that env's
        keep = alive & ~done_o              # counters differ from here on — dropped from the comparison for the rest of the rollout
        worst = np.maximum(worst, np.abs(O[t + 1][keep] - obs_o[keep]).max(axis=0))
        scale = np.maximum(scale, np.abs(obs_o[keep]).max(axis=0))
        worst_r = max(worst_r, float(np.abs(_np(rew[t])[alive] - rew_o[alive]).max()))
        d = alive & done_o
        n_done += int(d.sum())
        if d.any():
            np.testing.assert_allclose(T[t][d], info['terminal_observation'][d], rtol=2e-5, atol=2e-5)
            np.testing.assert_allclose(O[t + 1][d], obs_o[d], rtol=2e-6, atol=2e-6)      # the auto-reset inside the launch: same Philox draws
    assert alive.mean() >= 0.98, alive.mean()
    assert (worst[:6] <= 2e-5 * scale[:6]).all() and worst_r <= 3e-5, (worst, scale, worst_r)      # (the one-step tolerances of test_gpu_env_parity.py)
    assert n_done > 0                                        # episodes ended (time limit: 250 steps) and were reset inside the launch
    env.close()


def _load_shipped_policy(ppo, pol, task):
    """tests/golden/policies.npz: the actor / critic tensors of the reference's shipped checkpoints under their state-dict names
    (the parameters are views of the agent's flat buffer: load_state_dict copies in place)."""
    ppo.agent.ac.load_state_dict({k[len(task) + 1:]: torch.as_tensor(pol[k]) for k in pol.files if k.startswith(task + '/')})

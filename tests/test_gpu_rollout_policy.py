"""scg_rollout_policy — K control steps per launch with the actor in the loop — against the step-by-step path
(PyTorch actor + scg_step), and the PPO collector / evaluation built on it."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


STAB6 = dict(task='stabilization', obs_goal_horizon=0, episode_len_sec=1.0)         # Quadrotor2D with the reference's default 6-float rows


def _setup(task, n, hidden, act, seed=5, override=None):
    from safe_control_gym_amd.ppo import PPO, PPOConfig
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    env_id, cfg = load_task(task)
    if override:
        cfg = {k: v for k, v in dict(cfg, **override).items() if k != 'task_info'}
    env = HipVecEnv(env_id, n, seed=seed, return_numpy=False, policy=(hidden, act), **cfg)
    ref = HipVecEnv(env_id, n, seed=seed, return_numpy=False, **cfg)
    assert env.policy_shape == (hidden, act)
    pcfg = PPOConfig(hidden_dim=hidden, activation=act, use_gae=True, rollout_batch_size=n, rollout_steps=8, mini_batch_size=n * 8 // 2,
                     opt_epochs=1)
    torch.manual_seed(0)
    ppo = PPO(env, pcfg, seed=0)
    assert ppo._fused_rollout and ppo.agent.use_fused
    return env, ref, ppo


@pytest.mark.parametrize('task,hidden,act,override', [('quadrotor_2D_track', 128, 'tanh', None), ('cartpole_stab', 64, 'leaky_relu', None),
                                                      ('quadrotor_3D_track', 128, 'relu', None),
                                                      ('quadrotor_2D_track', 96, 'tanh', None),       # three feature tiles per hidden layer
                                                      ('quadrotor_2D_track', 64, 'tanh', STAB6)])     # rows of 24 bytes: no LDS transpose
# launch geometry: envs per wave (lane = env / lane pair = env) x waves per workgroup behind one weight image (8 = two waves per SIMD: what shards
# above 32 768 envs get by default — forced here on a small batch)
@pytest.mark.parametrize('epw,wpw', [('64', '4'), ('32', '4'), ('32', '8'), ('64', '8')])
def test_fused_rollout_equals_step_by_step(task, hidden, act, override, epw, wpw, monkeypatch):
    monkeypatch.setenv('SCG_ROLLOUT_EPW', epw)
    monkeypatch.setenv('SCG_ROLLOUT_WPW', wpw)
    n, K = 320, 12                                 # 5 / 10 waves: a partial workgroup as well
    env, ref, ppo = _setup(task, n, hidden, act, override=override)
    assert env.spec.obs_dim == (6 if override else env.spec.obs_dim)
    nobs, nu = env.spec.obs_dim, env.spec.nu
    f = dict(device=env.device, dtype=torch.float32)
    obs, actb, logp, rew = torch.zeros(K + 1, n, nobs, **f), torch.zeros(K, n, nu, **f), torch.zeros(K, n, **f), torch.zeros(K, n, **f)
    done, flags = torch.zeros(K, n, dtype=torch.uint8, device=env.device), torch.zeros(K, n, dtype=torch.uint8, device=env.device)
    term, acc = torch.zeros(K, n, nobs, **f), torch.zeros(n, 8, **f)
    # (a) deterministic policy: must reproduce actor-mean actions fed to scg_step one step at a time
    ref.reset_tensors()                             # (PPO's constructor already reset `env` once: keep the episode indices equal)
    env.reset_tensors(); o = ref.reset_tensors().clone()
    env.rollout_policy(ppo._policy_struct(True), K, obs, actb, logp, rew, done, flags, terminal_obs=term, episode_acc=acc)
    torch.cuda.synchronize()
    ac = ppo.agent.ac
    n_done = 0
    for t in range(K):
        torch.testing.assert_close(obs[t], o, rtol=2e-4, atol=2e-4)
        with torch.no_grad():
            a = ac.act(obs[t])                      # same inputs as the kernel saw (no drift between the two paths)
        torch.testing.assert_close(actb[t], a, rtol=1e-4, atol=2e-5)
        out = ref.step_tensors(actb[t])
        torch.testing.assert_close(rew[t], out.reward, rtol=2e-4, atol=2e-5)
        assert torch.equal(done[t], out.done) and torch.equal(flags[t], out.flags)
        d = out.done.bool()
        if d.any():
            torch.testing.assert_close(term[t][d], out.terminal_obs[d], rtol=2e-4, atol=2e-4)
        n_done += int(d.sum())
        o = out.obs.clone()
    torch.testing.assert_close(obs[K], o, rtol=2e-4, atol=2e-4)
    assert int(acc[:, 0].sum()) == n_done
    lp = -(ac.actor.logstd + 0.9189385332).sum().item()
    assert torch.allclose(logp, torch.full_like(logp, lp), atol=1e-5)
    # (b) stochastic: log-prob consistent with the sampled action, unit normal noise, reproducible
    env.reset_tensors()
    env.rollout_policy(ppo._policy_struct(False), K, obs, actb, logp, rew, done, flags, terminal_obs=term, episode_acc=acc)
    from safe_control_gym_amd.ppo import normal_log_prob
    with torch.no_grad():
        mean, logstd = ac.actor(obs[:K].reshape(K * n, nobs))
        ref_lp = normal_log_prob(mean, logstd, actb.reshape(K * n, nu)).reshape(K, n)
        z = ((actb.reshape(K * n, nu) - mean) * torch.exp(-logstd))
    torch.testing.assert_close(logp, ref_lp, rtol=1e-3, atol=2e-3)
    assert abs(z.mean().item()) < 0.05 and abs(z.std().item() - 1.0) < 0.05
    a1 = actb.clone()
    env.seed(5); env.reset_tensors()                                 # same key, fresh episodes -> different draws (episode index)
    env.rollout_policy(ppo._policy_struct(False), K, obs, actb, logp, rew, done, flags, terminal_obs=term, episode_acc=acc)
    assert not torch.equal(a1, actb)
    env.close(); ref.close()



def test_rollout_does_not_depend_on_the_launch_geometry(monkeypatch):
    """Stochastic rollouts with 64 and with 32 envs per wave, 4 and 8 waves per workgroup are the same rollouts: the noise is a per-env
    Philox stream, the actor's arithmetic per env is the same MFMA sequence."""
    outs = []
    for epw, wpw in (('64', '4'), ('32', '4'), ('32', '8'), ('64', '8')):
        monkeypatch.setenv('SCG_ROLLOUT_EPW', epw)
        monkeypatch.setenv('SCG_ROLLOUT_WPW', wpw)
        env, ref, ppo = _setup('quadrotor_2D_track', 1000, 128, 'tanh')
        n, K, nobs, nu = 1000, 40, 12, 2
        f = dict(device=env.device, dtype=torch.float32)
        obs, actb, logp, rew = torch.zeros(K + 1, n, nobs, **f), torch.zeros(K, n, nu, **f), torch.zeros(K, n, **f), torch.zeros(K, n, **f)
        done, flags = torch.zeros(K, n, dtype=torch.uint8, device=env.device), torch.zeros(K, n, dtype=torch.uint8, device=env.device)
        env.seed(5); env.reset_tensors()
        env.rollout_policy(ppo._policy_struct(False), K, obs, actb, logp, rew, done, flags)
        torch.cuda.synchronize()
        outs.append((obs.clone(), actb.clone(), logp.clone(), rew.clone(), done.clone(), env.get_raw_state() if hasattr(env, 'get_raw_state') else None))
        env.close(); ref.close()
    for other in outs[1:]:
        for a, b in zip(outs[0][:5], other[:5]):
            assert torch.equal(a, b)
    assert int(outs[0][4].sum()) > 0                # episodes ended and were reset inside the launch


def test_ppo_iteration_and_evaluation_on_the_fused_rollout():
    from safe_control_gym_amd.ppo import evaluate
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    env, ref, ppo = _setup('quadrotor_2D_track', 1024, 128, 'tanh')
    res = ppo.train_step()
    assert np.isfinite([res['policy_loss'], res['value_loss'], res['approx_kl']]).all() and res['minibatches'] == 2
    st = ppo.episode_stats()
    assert st['episodes'] > 0 and 0 < st['ep_length'] <= 250
    # evaluation: the fused one-launch path == the graphed PyTorch-policy path
    env_id, cfg = load_task('quadrotor_2D_track')
    ev_cfg = dict(cfg, randomized_init=False)
    e1 = HipVecEnv(env_id, 128, seed=9, return_numpy=False, policy=(128, 'tanh'), **ev_cfg)
    e2 = HipVecEnv(env_id, 128, seed=9, return_numpy=False, **ev_cfg)
    a = evaluate(ppo.agent.ac, e1, policy=ppo._policy_struct(True))
    b = evaluate(ppo.agent.ac, e2)
    assert a['episodes'] == b['episodes'] == 128
    for k in ('ep_return', 'ep_length', 'ep_constraint_violation', 'ep_mse'):
        assert abs(a[k] - b[k]) <= 1e-3 * max(1.0, abs(b[k])), (k, a[k], b[k])
    # the evaluation as a sequence of short launches (what AsyncEvaluator enqueues beside the training stream) == the one launch, bit for bit
    from safe_control_gym_amd.ppo import _evaluate_fused_device
    r1, a1 = _evaluate_fused_device(e1, ppo._policy_struct(True), 1)
    r1, a1 = r1.clone(), a1.clone()
    for chunk in (25, 64, 250, 1000):
        r2, a2 = _evaluate_fused_device(e1, ppo._policy_struct(True), 1, chunk=chunk)
        assert torch.equal(r1, r2) and torch.equal(a1, a2), chunk
    for e in (env, ref, e1, e2):
        e.close()

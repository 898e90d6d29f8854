"""GPU smoke/semantics tests of the PPO and SAC collectors on the HIP env."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def _env(task, n, **over):
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    env_id, cfg = load_task(task)
    cfg.update(over)
    return HipVecEnv(env_id, n, seed=5, return_numpy=False, **cfg)


def test_ppo_train_step_and_rollout_bookkeeping():
    from safe_control_gym_amd.ppo import PPO, PPOConfig, evaluate
    env = _env('quadrotor_2D_track', 512)
    cfg = PPOConfig(hidden_dim=32, use_gae=True, opt_epochs=2, mini_batch_size=2048, rollout_steps=16, actor_lr=1e-3, critic_lr=1e-3)
    ppo = PPO(env, cfg, seed=1)
    res = ppo.train_step()
    assert res['minibatches'] == 2 * (16 * 512 // 2048) and res['step'] == 16 * 512
    # the kernel wrote straight into the rollout tensors: obs[t+1] is the env's observation after step t
    # (a rollout slot binds obs to ppo.obs[t + 1]; the env's own default output buffer is NOT written by those steps, so the
    #  check is against the simulator: re-deriving the observation of the current state must give obs[T])
    assert torch.isfinite(ppo.obs[16]).all() and ppo.obs[16].abs().sum() > 0
    goal_cols = ppo.obs[16][:, 6:]                  # obs rows carry an X_GOAL row of the tracked trajectory in their second half
    x_goal = torch.as_tensor(env.spec.X_GOAL, dtype=torch.float32, device=env.device)
    d = (goal_cols[:, None, :] - x_goal[None, :, :]).abs().sum(-1).min(dim=1).values
    assert float(d.max()) < 1e-5
    assert torch.isfinite(ppo.rew).all() and float(ppo.rew.max()) <= 1.0 and float(ppo.rew.min()) >= 0.0
    assert ppo.done.sum() > 0          # some episodes ended (out-of-bounds resets)
    st = ppo.episode_stats()
    assert st['episodes'] == float(ppo.done.sum()) and 0 < st['ep_length'] <= 250
    ev_env = _env('quadrotor_2D_track', 64, randomized_init=False)
    ev = evaluate(ppo.agent.ac, ev_env)
    assert ev['episodes'] == 64 and 0 < ev['ep_length'] <= 250
    env.close(); ev_env.close()


def test_ppo_truncation_bootstrap_uses_terminal_observation():
    """ppo.py:274-284: a time-limit truncation adds gamma * V(terminal_obs) to the reward."""
    from safe_control_gym_amd.ppo import PPO, PPOConfig
    env = _env('cartpole_stab', 64, episode_len_sec=1, done_on_out_of_bound=False, randomized_init=False,
               init_state={'init_x': 0.0, 'init_x_dot': 0.0, 'init_theta': 0.0, 'init_theta_dot': 0.0})
    cfg = PPOConfig(hidden_dim=16, use_gae=True, opt_epochs=1, mini_batch_size=64 * 20, rollout_steps=20)
    ppo = PPO(env, cfg, seed=0)
    ppo.collect()
    trunc = ((ppo.flags & 1).bool() & ppo.done.bool())
    assert trunc[14].all() and trunc.sum() == 64          # 15 control steps per 1 s episode at 15 Hz
    assert (ppo.term_obs[14].abs().sum(-1) > 0).all()
    env.close()


def test_sac_train_step_with_time_limit_fixup():
    from safe_control_gym_amd.sac import SAC, SACConfig
    env = _env('cartpole_stab', 256, episode_len_sec=1, done_on_out_of_bound=False)
    cfg = SACConfig(hidden_dim=32, warm_up_steps=256 * 4, train_interval=256, train_batch_size=128,
                    max_buffer_size=256 * 64, extra={'updates_per_step': 4})
    sac = SAC(env, cfg, seed=0)
    upd = 0
    for t in range(20):
        res = sac.train_step()
        upd += res.get('updates', 0)
    assert upd > 0 and sac.buffer.size == 256 * 20
    # transitions stored at the truncation step keep mask 1 and the terminal observation as next_obs
    m = sac.buffer.mask[:sac.buffer.size, 0]
    assert float(m.min()) == 1.0          # no true terminations in this config (no out-of-bound done)
    assert torch.isfinite(sac.buffer.next_obs).all()
    env.close()


def test_vec_record_episode_statistics_matches_reference_fixture():
    """Replay of the reference's own VecRecordEpisodeStatistics output (tests/golden): queues and accumulators."""
    import json, os
    from safe_control_gym_amd.record_episode_statistics import VecRecordEpisodeStatistics, make_vec_envs
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'rollout_quadrotor_2D_track_policy.npz'))
    meta = json.loads(str(g['meta_json']))
    cfg = dict(meta['config']); cfg.pop('seed', None)
    venv = VecRecordEpisodeStatistics(make_vec_envs('quadrotor', cfg, meta['n_envs'], 1, seed=1, dtype=torch.float64), deque_size=1000)
    venv.add_tracker('constraint_violation', 0)
    venv.add_tracker('constraint_violation', 0, mode='queue')
    venv.add_tracker('mse', 0, mode='queue')
    venv.reset()
    env = venv.venv
    env.set_raw_state(g['state0'])
    for t in range(meta['n_steps']):
        obs, rew, done, info = venv.step(g['actions'][t])
        np.testing.assert_array_equal(done, g['done'][t])
        d = np.nonzero(done)[0]
        if len(d):
            raw = env.get_raw_state()
            raw[d] = g['state'][t][d]
            env.set_raw_state(raw)
        if t % 16 == 15:
            env.set_raw_state(g['state'][t])
    np.testing.assert_allclose(np.asarray(venv.return_queue), g['return_queue'], rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(np.asarray(venv.length_queue), g['length_queue'])
    np.testing.assert_allclose(venv.accumulated_stats['constraint_violation'], float(g['accumulated_violation']))
    np.testing.assert_allclose(np.asarray(venv.queued_stats['mse']), g['queued_mse'], rtol=1e-3, atol=1e-5)
    venv.close()


def test_ppo_graphed_update_matches_eager_update():
    """The HIP-graph replay of the minibatch step (flat parameters, gated Adam) must reproduce the reference-shaped
    eager update (two torch.optim.Adam, actor step skipped when approx_kl > 1.5 target_kl)."""
    from safe_control_gym_amd.ppo import PPOAgent, PPOConfig
    dev = torch.device('cuda')
    M, obs_dim, act_dim = 8192, 12, 2
    g = torch.Generator(device=dev); g.manual_seed(3)
    data = {'obs': torch.randn(M, obs_dim, device=dev, generator=g), 'act': torch.randn(M, act_dim, device=dev, generator=g),
            'logp': -2.0 + 0.3 * torch.randn(M, device=dev, generator=g), 'adv': torch.randn(M, device=dev, generator=g),
            'ret': torch.randn(M, device=dev, generator=g), 'v': torch.randn(M, device=dev, generator=g)}
    from safe_control_gym_amd.ppo import normal_log_prob
    torch.manual_seed(11)
    ref = PPOAgent(obs_dim, act_dim, PPOConfig(hidden_dim=32, extra={'cuda_graphs': False}), dev)
    with torch.no_grad():                       # behaviour policy = the initial policy: approx_kl starts at 0 and grows
        mean, logstd = ref.ac.actor(data['obs'])
        data['act'] = mean + torch.exp(logstd) * data['act']
        data['logp'] = normal_log_prob(mean, logstd, data['act'])
    results = []
    for graphs in (False, True):
        torch.manual_seed(11)
        cfg = PPOConfig(hidden_dim=32, opt_epochs=3, mini_batch_size=1024, actor_lr=3e-3, critic_lr=2e-3, target_kl=0.004,
                        extra={'cuda_graphs': graphs})
        agent = PPOAgent(obs_dim, act_dim, cfg, dev)
        assert agent.use_graphs == graphs
        gen = torch.Generator(device=dev); gen.manual_seed(5)
        res = agent.update({k: v.clone() for k, v in data.items()}, generator=gen)
        results.append((res, {k: v.detach().clone() for k, v in agent.ac.state_dict().items()}))
    (r0, w0), (r1, w1) = results
    assert r0['minibatches'] == r1['minibatches'] == 24
    assert r0['actor_steps'] == r1['actor_steps'] and 0 < r0['actor_steps'] < 24     # the KL gate fired on both paths
    for k in ('policy_loss', 'value_loss', 'entropy_loss', 'approx_kl'):
        assert abs(r0[k] - r1[k]) <= 1e-4 * max(1.0, abs(r0[k])), (k, r0[k], r1[k])
    for k in w0:
        torch.testing.assert_close(w1[k], w0[k], rtol=2e-4, atol=2e-6)


def test_ppo_graphed_iteration_runs_and_learns_signal():
    from safe_control_gym_amd.ppo import PPO, PPOConfig
    env = _env('quadrotor_2D_track', 1024)
    cfg = PPOConfig(hidden_dim=32, use_gae=True, opt_epochs=2, mini_batch_size=4096, rollout_steps=16, actor_lr=1e-3, critic_lr=1e-3)
    ppo = PPO(env, cfg, seed=2)
    assert ppo.agent.use_graphs
    r = [ppo.train_step() for _ in range(3)]       # first call captures, later calls replay
    assert r[-1]['step'] == 3 * 16 * 1024 and all(np.isfinite(x['value_loss']) for x in r)
    assert torch.isfinite(ppo.obs).all() and ppo.done.sum() > 0
    env.close()


def test_ppo_checkpoint_resume_is_exact(tmp_path):
    """save -> (new process-equivalent: fresh env + trainer) -> load -> continue == uninterrupted run (eager mode: the
    PyTorch RNG streams are restored bit for bit; graph replay keeps its own Philox offsets)."""
    from safe_control_gym_amd.ppo import PPO, PPOConfig

    def make():
        env = _env('quadrotor_2D_track', 256)
        cfg = PPOConfig(hidden_dim=16, use_gae=True, opt_epochs=2, mini_batch_size=1024, rollout_steps=8,
                        extra={'cuda_graphs': False})
        return env, PPO(env, cfg, seed=4)

    env_a, a = make()
    a.train_step(); a.train_step()
    path = str(tmp_path / 'ckpt' / 'model.pt')
    a.save(path)
    a.train_step()
    ref = {k: v.clone() for k, v in a.agent.ac.state_dict().items()}
    env_b, b = make()
    b.load(path)
    assert b.total_steps == 2 * 8 * 256
    b.train_step()
    for k, v in b.agent.ac.state_dict().items():
        torch.testing.assert_close(v, ref[k], rtol=0, atol=0)
    torch.testing.assert_close(b.obs, a.obs, rtol=0, atol=0)
    env_a.close(); env_b.close()


def test_ppo_checkpoint_resume_in_fused_mode(tmp_path):
    """The default GPU mode (fused rollout + fused update: Adam moments in the flat buffers, torch optimisers never step,
    in-kernel Philox action noise, keyed minibatch permutations): save -> fresh env + trainer -> load -> continue follows the
    uninterrupted run bit for bit (the gradient kernel of this network shape uses no atomics), which it cannot do if the
    moments, step counts, env counters or permutation state were dropped."""
    from safe_control_gym_amd.ppo import PPO, PPOConfig
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    env_id, tc = load_task('quadrotor_2D_track')

    def make():
        env = HipVecEnv(env_id, 2048, seed=6, return_numpy=False, policy=(128, 'tanh'), **tc)
        cfg = PPOConfig(hidden_dim=128, activation='tanh', use_gae=True, opt_epochs=2, mini_batch_size=16384, rollout_steps=16,
                        actor_lr=1e-3, critic_lr=1e-3, target_kl=0.05)
        return env, PPO(env, cfg, seed=6)

    env_a, a = make()
    assert a._fused_rollout and a.agent.use_fused
    for _ in range(3):
        a.train_step()
    path = str(tmp_path / 'fused' / 'model.pt')
    a.save(path)
    a.train_step()
    ref = torch.cat([p.detach().reshape(-1) for p in a.agent.ac.parameters()]).clone()
    env_b, b = make()
    b.load(path)
    assert float(b.agent._flat['steps'][1]) == float(3 * 2 * (16 * 2048 // 16384)) and b.agent._perm_count == a.agent._perm_count - 2
    b.train_step()
    got = torch.cat([p.detach().reshape(-1) for p in b.agent.ac.parameters()])
    assert torch.equal(got, ref), float((got - ref).abs().max())      # bit for bit: nothing on this path sums in arrival order
    assert torch.equal(b.obs, a.obs)
    # control: the same continuation WITHOUT the Adam moments ends somewhere else
    env_c, c = make()
    c.load(path)
    for k in ('m', 'v', 'steps'):
        c.agent._flat[k].zero_()
    c.train_step()
    bad = torch.cat([p.detach().reshape(-1) for p in c.agent.ac.parameters()])
    assert float((bad - ref).abs().max()) > 1e-3
    for e in (env_a, env_b, env_c):
        e.close()


def test_ppo_iteration_graph_equals_per_launch_enqueue_and_never_waits():
    """One HIP-graph replay per iteration (PPO._run_iteration: fused collection, GAE, normalisation, the update's keyed permutations and
    optimiser steps — device-keyed, nothing about an iteration is baked into the graph) gives bit for bit what the same launches
    enqueued one by one give, iteration after iteration; train_step(lazy=True) returns device statistics + events and reads nothing back."""
    from safe_control_gym_amd.ppo import PPO, PPOAgent, PPOConfig
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    env_id, tc = load_task('quadrotor_2D_track')

    def make(graph):
        env = HipVecEnv(env_id, 2048, seed=8, return_numpy=False, policy=(128, 'tanh'), **tc)
        cfg = PPOConfig(hidden_dim=128, activation='tanh', use_gae=True, opt_epochs=3, mini_batch_size=4096, rollout_steps=16,
                        actor_lr=1e-3, critic_lr=1e-3, target_kl=0.03, extra={'minibatches_per_epoch': 3, 'iteration_graph': graph})
        return env, PPO(env, cfg, seed=8)

    env_a, a = make(True)
    env_b, b = make(False)
    assert a._iteration_graph_ok() and not b._iteration_graph_ok()
    flat = lambda p: torch.cat([q.detach().reshape(-1) for q in p.agent.ac.parameters()])      # noqa: E731
    for it in range(6):
        ra, rb = a.train_step(lazy=True), b.train_step(lazy=True)
        assert 'stats_dev' in ra and 'policy_loss' not in ra and len(ra['events']) == 3
        sa, sb = PPOAgent.stats_of(ra['stats_dev'], ra['minibatches']), PPOAgent.stats_of(rb['stats_dev'], rb['minibatches'])
        assert sa == sb, (it, sa, sb)
        assert torch.equal(flat(a), flat(b)), it
        assert torch.equal(a.obs, b.obs) and torch.equal(a.agent._flat['steps'], b.agent._flat['steps'])
        assert a.agent._perm_count == b.agent._perm_count == 3 * (it + 1)
        assert a.total_steps == b.total_steps == (it + 1) * 16 * 2048
    assert a._iter_graph['graph'] is not None                # iterations 2.. were replays
    assert int(a.agent._perm_dev[1]) == a.agent._perm_count     # the device mirror of the epochs drawn
    # the default (non-lazy) call returns the reference's float statistics from the same replay
    ra, rb = a.train_step(), b.train_step()
    assert ra['policy_loss'] == rb['policy_loss'] and ra['approx_kl'] == rb['approx_kl'] and ra['device_time'] > 0
    # a changed hyper-parameter is not silently ignored by the captured graph: it re-captures
    a.cfg.actor_lr = b.cfg.actor_lr = 5e-4
    for _ in range(3):
        a.train_step(lazy=True); b.train_step(lazy=True)
    assert torch.equal(flat(a), flat(b))
    env_a.close(); env_b.close()


def test_ppo_with_running_normalisers_graphed_and_eager():
    """norm_obs / norm_reward (ppo.yaml keys): the running statistics live on the device and update inside the captured
    rollout graph; same bookkeeping as the eager path."""
    from safe_control_gym_amd.ppo import PPO, PPOConfig, evaluate
    stats = []
    for graphs in (True, False):
        env = _env('quadrotor_2D_track', 512)
        cfg = PPOConfig(hidden_dim=16, use_gae=True, opt_epochs=1, mini_batch_size=2048, rollout_steps=8, norm_obs=True,
                        norm_reward=True, extra={'cuda_graphs': graphs})
        ppo = PPO(env, cfg, seed=3)
        for _ in range(2):
            res = ppo.train_step()
        assert np.isfinite(res['value_loss'])
        assert abs(float(ppo.obs_normalizer.rms.count) - (1e-4 + 512 * (1 + 2 * 8))) < 1e-6      # reset + 16 steps
        assert float(ppo.obs.abs().max()) <= cfg.clip_obs + 1e-6
        stats.append(ppo.obs_normalizer.rms.mean.cpu().numpy())
        ev = evaluate(ppo.agent.ac, _env('quadrotor_2D_track', 32, randomized_init=False), obs_normalizer=ppo.obs_normalizer)
        assert ev['episodes'] == 32
        assert abs(float(ppo.obs_normalizer.rms.count) - (1e-4 + 512 * 17)) < 1e-6                 # evaluation is read-only
        env.close()
    # positions dominate the first moments; both paths saw statistically identical data (different action noise streams)
    assert np.allclose(stats[0][[0, 2]], stats[1][[0, 2]], atol=0.2)


def test_rarl_alternates_protagonist_and_adversary_on_the_adversary_channel():
    """rarl.py:267-281,349-428: both policies act every step, the adversary through set_adversary_control into the
    kernel's dynamics channel; protagonist learns from +reward, adversary from -reward."""
    from safe_control_gym_amd.ppo import PPOConfig
    from safe_control_gym_amd.rarl import RARL
    env = _env('quadrotor_2D_track', 512, adversary_disturbance='dynamics', adversary_disturbance_scale=0.05)
    cfg = PPOConfig(hidden_dim=16, use_gae=True, opt_epochs=1, mini_batch_size=2048, rollout_steps=8, actor_lr=1e-3, critic_lr=1e-3)
    r = RARL(env, cfg, seed=1, agent_iterations=2, adversary_iterations=1)
    w_ag = {k: v.clone() for k, v in r.agent.ac.state_dict().items()}
    w_ad = {k: v.clone() for k, v in r.adversary.ac.state_dict().items()}
    res = r.train_step()
    assert res['step'] == 3 * 8 * 512 and 'value_loss' in res and 'value_loss_adv' in res
    assert r.act_adv.shape == (8, 512, 2) and float(r.act_adv.abs().sum()) > 0
    assert any(not torch.equal(v, w_ag[k]) for k, v in r.agent.ac.state_dict().items())
    assert any(not torch.equal(v, w_ad[k]) for k, v in r.adversary.ac.state_dict().items())
    # the adversary really reaches the dynamics: the same seed without it gives a different rollout
    env2 = _env('quadrotor_2D_track', 512, adversary_disturbance='dynamics', adversary_disturbance_scale=0.0)
    r2 = RARL(env2, PPOConfig(hidden_dim=16, use_gae=True, opt_epochs=1, mini_batch_size=2048, rollout_steps=8), seed=1)
    r2.collect()
    env3 = _env('quadrotor_2D_track', 512, adversary_disturbance='dynamics', adversary_disturbance_scale=0.05)
    r3 = RARL(env3, PPOConfig(hidden_dim=16, use_gae=True, opt_epochs=1, mini_batch_size=2048, rollout_steps=8), seed=1)
    r3.collect()
    assert not torch.allclose(r2.obs[1], r3.obs[1])
    for e in (env, env2, env3):
        e.close()
    with pytest.raises(ValueError):
        RARL(_env('quadrotor_2D_track', 64), cfg)


def test_rarl_graph_and_eager_collectors_agree_on_deterministic_quantities():
    """The captured two-policy collection (one HIP graph: T x (protagonist, adversary, env kernel) + both GAE passes) against
    the eager one: values are deterministic functions of the stored observations, and the adversary's returns are built
    from -reward."""
    from safe_control_gym_amd.ppo import PPOConfig
    from safe_control_gym_amd.rarl import RARL
    for graphs in (True, False):
        env = _env('quadrotor_2D_track', 256, adversary_disturbance='action', adversary_disturbance_scale=0.1)
        cfg = PPOConfig(hidden_dim=32, use_gae=True, opt_epochs=1, mini_batch_size=1024, rollout_steps=8, extra={'cuda_graphs': graphs})
        r = RARL(env, cfg, seed=3)
        assert (r._two_graph is None) and r._graph_rollout == graphs
        (ret, adv, mom), (ret_a, adv_a, mom_a) = r.collect()
        assert (r._two_graph is not None) == graphs
        with torch.no_grad():
            torch.testing.assert_close(r.v[3], r.agent.ac.critic(r.obs[3]).squeeze(-1), rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(r.v_adv[3], r.adversary.ac.critic(r.obs[3]).squeeze(-1), rtol=1e-5, atol=1e-6)
            mean, logstd = r.adversary.ac.actor(r.obs[5])
            lp = (-0.5 * ((r.act_adv[5] - mean) / logstd.exp()) ** 2 - logstd - 0.9189385332046727).sum(-1)
            torch.testing.assert_close(r.logp_adv[5], lp, rtol=1e-4, atol=1e-4)
        # the adversary's returns / advantages: ppo_utils.py:374-402 on -reward with the adversary's critic
        with torch.no_grad():
            crit = r.adversary.ac.critic
            mask = 1.0 - r.done.float()
            trunc = ((r.flags & 1).bool() & r.done.bool())
            tv = torch.where(trunc, crit(r.term_obs).squeeze(-1), torch.zeros_like(r.rew))
            rews = -r.rew + 0.99 * tv
            vals = torch.cat([r.v_adv, crit(r.obs[8]).squeeze(-1)[None]])
            run_ret, run_adv = vals[8], torch.zeros(256, device=r.device)
            for i in reversed(range(8)):
                run_ret = rews[i] + 0.99 * mask[i] * run_ret
                run_adv = run_adv * 0.95 * 0.99 * mask[i] + rews[i] + 0.99 * mask[i] * vals[i + 1] - vals[i]
                torch.testing.assert_close(ret_a[i], run_ret, rtol=1e-5, atol=1e-5)
                torch.testing.assert_close(adv_a[i], run_adv, rtol=1e-5, atol=1e-5)
        assert float(mom[2]) == 8 * 256 and torch.isfinite(adv).all() and torch.isfinite(adv_a).all()
        res = r.train_step()
        assert res['step'] == 3 * 8 * 256 and np.isfinite(res['policy_loss_adv'])
        env.close()


def test_rap_population_groups_and_updates():
    """rap.py:349-470: sorted adversary index per env (contiguous groups), every env faces ITS adversary, every collection
    starts from a reset, the protagonist learns from the whole batch and each sampled adversary from its slice."""
    from safe_control_gym_amd.ppo import PPOConfig
    from safe_control_gym_amd.rarl import RAP
    env = _env('quadrotor_2D_track', 384, adversary_disturbance='dynamics', adversary_disturbance_scale=0.05)
    cfg = PPOConfig(hidden_dim=32, use_gae=True, opt_epochs=2, mini_batch_size=512, rollout_steps=8, actor_lr=1e-3, critic_lr=1e-3)
    # what rap.py:70-71 / rarl.py:70-71 read from the vectorised env
    assert env.get_attr('adversary_observation_space')[0].shape == (12,) and env.get_attr('adversary_action_space')[0].shape == (2,)
    r = RAP(env, cfg, seed=4, num_adversaries=3)
    w0 = [{k: v.clone() for k, v in a.ac.state_dict().items()} for a in r.adversaries]
    wa = {k: v.clone() for k, v in r.agent.ac.state_dict().items()}
    res = r.train_step()
    idx = r.adv_index.cpu().numpy()
    assert (np.diff(idx) >= 0).all() and set(idx) <= {0, 1, 2}
    assert [g[0] for g in r.groups] == res['adv_indices'] == sorted(set(idx))
    assert r.groups[0][1] == 0 and r.groups[-1][2] == 384 and all(a[2] == b[1] for a, b in zip(r.groups, r.groups[1:]))
    for k, s, e in r.groups:
        assert (idx[s:e] == k).all()
        assert f'value_loss_adv{k}' in res
    assert res['step'] == 8 * 384
    assert any(not torch.equal(v, wa[k]) for k, v in r.agent.ac.state_dict().items())
    for k in range(3):
        changed = any(not torch.equal(v, w0[k][name]) for name, v in r.adversaries[k].ac.state_dict().items())
        assert changed == (k in res['adv_indices'])
    # a fresh collection: the stored adversary values are those of each env's own adversary
    r.collect()
    with torch.no_grad():
        for k, s, e in r.groups:
            torch.testing.assert_close(r.v_adv[2, s:e], r.adversaries[k].ac.critic(r.obs[2, s:e]).squeeze(-1), rtol=1e-5, atol=1e-6)
    step, _ = env.get_counters()
    assert (step <= 8).all()                        # the collection started from a reset
    env.close()


def test_safe_explorer_ppo_pretrain_and_step():
    """safe_ppo.py: constraint-model pre-training on random transitions (c_next = the step's pre-reset constraint values),
    then a PPO iteration whose policy mean is filtered by the safety layer with the current constraint values as input."""
    from safe_control_gym_amd.ppo import PPOConfig
    from safe_control_gym_amd.safe_explorer import SafeExplorerPPO
    env = _env('quadrotor_2D_track', 256)
    cfg = PPOConfig(hidden_dim=16, use_gae=True, opt_epochs=1, mini_batch_size=1024, rollout_steps=8, actor_lr=1e-3, critic_lr=1e-3)
    se = SafeExplorerPPO(env, cfg, seed=2, constraint_hidden_dim=16, constraint_batch_size=512)
    assert se.C == 12 and se.c.shape == (256, 12)
    # constraint values of the reset state == what the reset kernel reported
    torch.testing.assert_close(se.c, env.out.c_values[:12].t().to(torch.float32), rtol=0, atol=1e-6)
    hist = se.pretrain(256 * 40, epochs=4)
    assert se.constraint_buffer.size == 256 * 40 and np.mean(hist[-1]) < np.mean(hist[0])
    res = se.train_step()
    assert res['step'] == 8 * 256 and np.isfinite(res['policy_loss']) and se.c_buf.shape == (8, 256, 12)
    # where an episode ended the next policy input holds the NEW episode's constraint values, not the terminal ones
    d = se.done[-1].bool()
    if d.any():
        fresh = env.spec.state_constraint_values(env.out.state.t()).to(torch.float32)
        torch.testing.assert_close(se.c[d], fresh[d], rtol=0, atol=1e-6)
    env.close()
    with pytest.raises(ValueError):
        SafeExplorerPPO(_env('quadrotor_2D_track', 64, constraints=None), cfg)


def test_async_evaluator_matches_blocking_evaluation_and_overlaps_training():
    """ppo.AsyncEvaluator: the snapshot evaluated on the side stream is the weights at launch() time (training may already
    have moved on), the numbers equal a blocking evaluate() of those weights, and at most one evaluation is in flight."""
    from safe_control_gym_amd.ppo import PPO, AsyncEvaluator, PPOConfig, evaluate
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    env_id, cfg = load_task('quadrotor_2D_track')
    pol = (128, 'tanh')
    env = HipVecEnv(env_id, 2048, seed=3, return_numpy=False, policy=pol, **cfg)
    ev_env = HipVecEnv(env_id, 256, seed=5, return_numpy=False, policy=pol, **dict(cfg, randomized_init=False))
    ev_env2 = HipVecEnv(env_id, 256, seed=5, return_numpy=False, policy=pol, **dict(cfg, randomized_init=False))
    ppo = PPO(env, PPOConfig(hidden_dim=128, activation='tanh', use_gae=True, opt_epochs=2, mini_batch_size=16384, rollout_steps=16,
                             actor_lr=2e-3, critic_lr=2e-3, target_kl=0.03), seed=3)
    assert ppo._fused_rollout
    aev = AsyncEvaluator(ppo, ev_env)
    assert aev.poll() is None
    ppo.train_step()
    ref0 = evaluate(ppo.agent.ac, ev_env2, policy=ppo._policy_struct(True))          # blocking, weights after iteration 1
    assert aev.launch(tag=1)
    assert not aev.launch(tag=99)                                                      # one in flight
    ppo.train_step(); ppo.train_step()                                                 # training moves the weights on
    got = aev.poll(wait=True)
    assert got['tag'] == 1
    for k in ('episodes', 'ep_return', 'ep_length', 'ep_mse', 'ep_constraint_violation'):
        assert got[k] == pytest.approx(ref0[k], rel=1e-6), k
    ref2 = evaluate(ppo.agent.ac, ev_env2, policy=ppo._policy_struct(True))
    assert ref2['ep_return'] != ref0['ep_return']                                      # (they did move)
    assert aev.launch(tag=3) and aev.poll(wait=True)['ep_return'] == pytest.approx(ref2['ep_return'], rel=1e-6)
    for e in (env, ev_env, ev_env2):
        e.close()


@pytest.mark.parametrize('norm', [False, True, 'fused'], ids=['plain', 'normalised', 'fused'])
def test_sac_graph_collector_equals_the_eager_collector(norm):
    """SAC.train_step's collector as ONE HIP-graph replay per vector step (policy / uniform action, env step kernel, time-limit fix-up,
    normalisers, device-side ring push) against the same body run eagerly: warm-up and policy phases (two eager steps, then capture +
    replays, per phase), short episodes so that truncations and auto-resets occur, a ring that wraps.  Same seeds -> same Philox
    draws on both sides (torch's graph-safe generator): ring contents, write position, current observation and the normalisers'
    statistics must agree."""
    from safe_control_gym_amd.sac import SAC, SACConfig
    N = 256
    fused, norm = norm == 'fused', norm is True

    def run(graphs):
        env = _env('quadrotor_2D_track', N, episode_len_sec=0.12)               # 6-step episodes
        # plain / normalised: the PyTorch collector (fused_collect off), everything eager vs graphs; fused: the library collector
        # (scg_sac_sample / scg_sac_push, in-kernel Philox noise) replayed as a graph vs enqueued call by call
        extra = {'graph_collect': graphs} if fused else {'cuda_graphs': graphs, 'fused_collect': False}
        if norm:
            extra.update(norm_obs=True, norm_reward=True, clip_obs=5.0)
        cfg = SACConfig(hidden_dim=32, activation='relu', warm_up_steps=5 * N, train_interval=10 ** 9, train_batch_size=64,
                        max_buffer_size=9 * N, extra=extra)
        torch.manual_seed(123)
        sac = SAC(env, cfg, seed=7)
        assert sac._graph_collect == graphs and sac._fused_collect == fused
        for _ in range(12):                                                     # 5 warm-up + 7 policy steps; 12 N pushes into 9 N slots
            assert 'updates' not in sac.train_step()
        torch.cuda.synchronize()
        b = sac.buffer
        st = {k: getattr(b, k).clone() for k in ('obs', 'act', 'rew', 'next_obs', 'mask')}
        st.update(obs_now=sac.obs.clone(), pos=b.pos, size=b.size, pos_t=int(b.pos_t), size_i32=int(b.size_i32), total=sac.total_steps,
                  graphs=sorted(k[0] for k, v in sac._collect_graphs.items() if v['g'] is not None))
        if norm:
            st.update(mean=sac.obs_normalizer.rms.mean.clone(), var=sac.obs_normalizer.rms.var.clone(), ret=sac.reward_normalizer.ret.clone(),
                      rvar=sac.reward_normalizer.rms.var.clone())
        env.close()
        return st
    g, e = run(True), run(False)
    assert g['graphs'] == [False, True] and e['graphs'] == []                   # both phases were captured / none was
    assert (g['pos'], g['size'], g['pos_t'], g['size_i32'], g['total']) == (e['pos'], e['size'], e['pos_t'], e['size_i32'], e['total']) \
        == (3 * N, 9 * N, 3 * N, 9 * N, 12 * N)
    assert (g['mask'] == 0).sum() > 0 and (g['mask'] == 1).sum() > 0
    for k in ('obs', 'act', 'rew', 'next_obs', 'mask', 'obs_now') + (('mean', 'var', 'ret', 'rvar') if norm else ()):
        torch.testing.assert_close(g[k], e[k], rtol=1e-5, atol=1e-5, msg=lambda m, k=k: f'{k}: {m}')


def test_sac_evaluation_with_the_fused_deterministic_actor_equals_the_torch_policy():
    """ppo.evaluate driven by SACAgent.deterministic_policy() (scg_sac_act: one launch per evaluation step on the flat parameters)
    against the same evaluation with the torch modules: same episodes, returns to float32 noise; and the packed accumulators of
    evaluate() still expose per-env totals (controllers.run reads them)."""
    from safe_control_gym_amd.ppo import evaluate
    from safe_control_gym_amd.sac import SACAgent, SACConfig
    dev = torch.device('cuda', 0)
    e1 = _env('quadrotor_3D_track', 64, randomized_init=False)
    e2 = _env('quadrotor_3D_track', 64, randomized_init=False)
    spec = e1.spec
    low = torch.as_tensor(spec.action_space.low, dtype=torch.float32, device=dev)
    high = torch.as_tensor(spec.action_space.high, dtype=torch.float32, device=dev)
    torch.manual_seed(2)
    ag = SACAgent(spec.obs_dim, spec.nu, low, high, SACConfig(hidden_dim=128, activation='relu'), dev)
    assert ag.use_fused
    obs = torch.randn(300, spec.obs_dim, device=dev)
    torch.testing.assert_close(ag.act_deterministic(obs), ag.ac.act(obs, deterministic=True), rtol=1e-4, atol=1e-5)

    class Torch:
        ac = ag.ac

        @staticmethod
        def act(o):
            return ag.ac.act(o, deterministic=True)
    a = evaluate(ag.deterministic_policy(), e1)
    b = evaluate(Torch(), e2)
    assert a['episodes'] == b['episodes'] == 64
    for k in ('ep_return', 'ep_length', 'ep_mse', 'ep_constraint_violation'):
        assert a[k] == pytest.approx(b[k], rel=5e-3, abs=5e-3), (k, a[k], b[k])         # (a termination one step apart in one env of 64)
    acc = e1._eval_cache['acc']
    assert acc['ret'].shape == (64,) and float(acc['count'].sum()) == 64 and float(acc['length'].min()) >= 1
    assert ag.deterministic_policy() is ag.deterministic_policy()
    e1.close(); e2.close()

"""SURVEY §8f-3 collectors pinned to the reference: rarl.RARL / rarl.RAP replay the transitions the REFERENCE's own
`RARL.collect_rollouts` / `RAP.collect_rollouts` (controllers/rarl/rarl.py:349-428, rap.py:349-470) recorded on the
reference's envs (tests/golden/make_adversarial.py -> adversarial.npz) with the same sampled actions, and must reproduce the
reference's rollout buffers: which side's action / value / log-prob is stored, the adversary's negated reward, whose critic
bootstraps a time-limit-truncated episode, returns, normalised advantages — on the eager path, on the HIP-graph path, and,
for RARL, through the update to the reference's final weights.  The env is tests/replay_env.py (the product's collectors,
GAE kernel and learners are the real ones)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'adversarial.npz'))
OVER = dict(episode_len_sec=0.2, adversary_disturbance='dynamics', adversary_disturbance_scale=0.05, adversary_disturbance_offset=0.01,
            randomized_init=False, done_on_out_of_bound=True)


def sd(prefix):
    return {k[len(prefix) + 1:]: torch.as_tensor(G[k]) for k in G.files if k.startswith(prefix + '/')}


def _replay(prefix, obs0, dev):
    from tests.replay_env import ReplayVecEnv, spec_for
    tr = {k: G[f'{prefix}/transitions/{k}'] for k in ('next_obs', 'rew', 'done', 'trunc', 'term_obs')}
    return ReplayVecEnv(spec_for(OVER), dev, obs0, tr['next_obs'], tr['rew'], tr['done'], tr['trunc'], tr['term_obs'])


def _check_side(buf_prefix, ppo, ret, adv_n, act, v, logp, rew_sign, atol=2e-5, gamma=0.99):
    B = {k: torch.as_tensor(G[f'{buf_prefix}/{k}'], dtype=torch.float32) for k in ('obs', 'act', 'rew', 'mask', 'v', 'logp', 'terminal_v', 'ret', 'adv')}
    cpu = lambda t: t.detach().float().cpu()                                     # noqa: E731
    torch.testing.assert_close(cpu(ppo.obs[:ppo.T]), B['obs'], rtol=0, atol=1e-6)
    torch.testing.assert_close(cpu(act), B['act'], rtol=0, atol=1e-6)
    # (compute_returns_and_advantages adds gamma * terminal_v to the buffer's rewards IN PLACE, ppo_utils.py:389: undo it)
    torch.testing.assert_close(rew_sign * cpu(ppo.rew), (B['rew'] - gamma * B['terminal_v'])[..., 0], rtol=0, atol=1e-6)
    torch.testing.assert_close(1.0 - cpu(ppo.done), B['mask'][..., 0], rtol=0, atol=0)
    torch.testing.assert_close(cpu(v), B['v'][..., 0], rtol=1e-5, atol=atol)
    torch.testing.assert_close(cpu(logp), B['logp'][..., 0], rtol=1e-5, atol=atol)
    torch.testing.assert_close(cpu(ret), B['ret'][..., 0], rtol=1e-5, atol=atol)
    torch.testing.assert_close(cpu(adv_n), B['adv'][..., 0], rtol=1e-4, atol=1e-4)
    assert B['terminal_v'].abs().max() > 0                                           # the fixture does hold truncated rows


@pytest.mark.parametrize('graphs', [False, True])
def test_rarl_collector_reproduces_the_reference_buffers(graphs):
    from safe_control_gym_amd import rarl
    from safe_control_gym_amd.ppo import PPOConfig
    from tests.replay_env import forced_step
    dev = torch.device('cuda', 0)
    gam, lam = G['rarl/gamma_lambda']
    cfg = PPOConfig(hidden_dim=16, activation='tanh', use_gae=True, gamma=float(gam), gae_lambda=float(lam), rollout_batch_size=4, rollout_steps=12,
                    opt_epochs=2, mini_batch_size=16, actor_lr=3e-3, critic_lr=1e-3, target_kl=0.01, entropy_coef=0.01,
                    extra={'cuda_graphs': graphs, 'fused_update': False})
    obs0 = G['rarl/obs0']
    for tag, adversary in (('agent', False), ('adversary', True)):
        env = _replay(f'rarl/{tag}', obs0, dev)
        ctl = rarl.RARL(env, PPOConfig(**{**cfg.__dict__}), seed=0)
        assert ctl._graph_rollout == graphs
        # weights at the time of the reference's collection: the protagonist had been updated before the adversary's turn
        ctl.agent.ac.load_state_dict(sd('rarl/agent/final' if adversary else 'rarl/agent_init'))
        ctl.adversary.ac.load_state_dict(sd('rarl/adversary_init'))
        f = dict(device=dev, dtype=torch.float32)
        a_pro = torch.as_tensor(G[f'rarl/{tag}/transitions/act'], **f)
        a_adv = torch.as_tensor(G[f'rarl/{tag}/transitions/adv_raw'], **f)
        ctl.agent.ac.step = forced_step(ctl.agent.ac, ctl.obs, a_pro)
        ctl.adversary.ac.step = forced_step(ctl.adversary.ac, ctl.obs, a_adv)
        (ret, adv, mom), (ret_a, adv_a, mom_a) = ctl._collect_both()
        torch.cuda.synchronize()
        # what reached the env: the protagonist's action as is, the adversary's after clip / scale / offset
        torch.testing.assert_close(env.seen_act.cpu(), a_pro.cpu(), rtol=0, atol=0)
        np.testing.assert_allclose(env.seen_adv.cpu().numpy(), G[f'rarl/{tag}/transitions/adv_applied'], rtol=0, atol=1e-6)
        if adversary:
            adv_n = rarl._normalised(adv_a, mom_a)
            _check_side(f'rarl/{tag}/buffer', ctl, ret_a, adv_n, ctl.act_adv, ctl.v_adv, ctl.logp_adv, -1.0)
            side_ret, learner = ret_a, ctl.adversary
        else:
            adv_n = rarl._normalised(adv, mom)
            _check_side(f'rarl/{tag}/buffer', ctl, ret, adv_n, ctl.act, ctl.v, ctl.logp, 1.0)
            side_ret, learner = ret, ctl.agent
        if not graphs:          # ... and through PPOAgent.update with the reference's minibatch permutations to its final weights
            res = learner.update(ctl._data(adversary, side_ret, adv_n), perms=G[f'rarl/{tag}/perms'])
            np.testing.assert_allclose([res['policy_loss'], res['value_loss'], res['entropy_loss'], res['approx_kl']],
                                       G[f'rarl/{tag}/results'], rtol=2e-4, atol=2e-5)
            final = sd(f'rarl/{tag}/final')
            for k, v in learner.ac.state_dict().items():
                torch.testing.assert_close(v.cpu(), final[k], rtol=1e-3, atol=2e-5, msg=lambda m, k=k: f'{tag} {k}: {m}')
        obs0 = G[f'rarl/{tag}/transitions/next_obs'][-1]        # the next collection starts where this one ended (rarl.py:409)


@pytest.mark.parametrize('graphs', [False, True])
def test_rap_collector_reproduces_the_reference_buffers(graphs):
    from safe_control_gym_amd import rarl
    from safe_control_gym_amd.ppo import PPOConfig
    from tests.replay_env import forced_step
    dev = torch.device('cuda', 0)
    cfg = PPOConfig(hidden_dim=16, activation='tanh', use_gae=True, gamma=0.99, gae_lambda=0.95, rollout_batch_size=6, rollout_steps=12,
                    opt_epochs=1, mini_batch_size=8, extra={'cuda_graphs': graphs, 'fused_update': False})
    env = _replay('rap', G['rap/obs0'], dev)
    ctl = rarl.RAP(env, cfg, seed=0, num_adversaries=3)
    ctl.agent.ac.load_state_dict(sd('rap/agent_init'))
    for k, a in enumerate(ctl.adversaries):
        a.ac.load_state_dict(sd(f'rap/adversary{k}_init'))
    f = dict(device=dev, dtype=torch.float32)
    a_pro, a_adv = torch.as_tensor(G['rap/transitions/act'], **f), torch.as_tensor(G['rap/transitions/adv_raw'], **f)
    ctl.agent.ac.step = forced_step(ctl.agent.ac, ctl.obs, a_pro)
    for a in ctl.adversaries:
        a.ac.step = forced_step(a.ac, ctl.obs, a_adv)
    idx = G['rap/adv_indices']

    class FixedDraw:                                    # the reference's draw (np.random.randint under seed 5, rap.py:356)
        def randint(self, n, size):
            assert n == 3 and size == 6
            return idx[::-1].copy()                     # (unsorted on purpose: RAP.collect sorts)
    ctl._rng = FixedDraw()
    (ret, adv, mom), (ret_a, adv_a, mom_a) = ctl.collect()
    torch.cuda.synchronize()
    assert [g[0] for g in ctl.groups] == G['rap/split_ids'].tolist()
    _check_side('rap/agent/buffer', ctl, ret, rarl._normalised(adv, mom), ctl.act, ctl.v, ctl.logp, 1.0)
    adv_an = rarl._normalised(adv_a, mom_a)             # over the WHOLE batch (rap.py:448), then split by group
    for k, s, e in ctl.groups:
        B = {n: torch.as_tensor(G[f'rap/adversary{k}/buffer/{n}'], dtype=torch.float32) for n in ('act', 'rew', 'v', 'logp', 'terminal_v', 'ret', 'adv')}
        cpu = lambda t: t[:, s:e].detach().float().cpu()                         # noqa: E731
        torch.testing.assert_close(cpu(ctl.act_adv), B['act'], rtol=0, atol=1e-6)
        torch.testing.assert_close(-cpu(ctl.rew), (B['rew'] - 0.99 * B['terminal_v'])[..., 0], rtol=0, atol=1e-6)
        torch.testing.assert_close(cpu(ctl.v_adv), B['v'][..., 0], rtol=1e-5, atol=2e-5)             # adversary k's critic on its envs
        torch.testing.assert_close(cpu(ctl.logp_adv), B['logp'][..., 0], rtol=1e-5, atol=2e-5)
        torch.testing.assert_close(cpu(ret_a), B['ret'][..., 0], rtol=1e-5, atol=2e-5)               # incl. ITS critic's truncation bootstrap
        torch.testing.assert_close(cpu(adv_an), B['adv'][..., 0], rtol=1e-4, atol=1e-4)
        d = ctl._data(True, ret_a, adv_an, s, e)
        assert d['obs'].shape == (12 * (e - s), 12) and d['act'].shape[0] == 12 * (e - s)

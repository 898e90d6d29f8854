"""CPU tests of the SAC learner pieces against direct statements of sac_utils.py."""
import math

import pytest
import torch

from safe_control_gym_amd.sac import DeviceReplay, MLPActorCritic, SACAgent, SACConfig
from tests.devices import DEVICES


def test_tanh_gaussian_logprob_matches_change_of_variables():
    torch.manual_seed(0)
    ac = MLPActorCritic(6, 2, [-1.0, -1.0], [1.0, 1.0], [16, 16], 'relu')
    obs = torch.randn(64, 6)
    torch.manual_seed(5)
    act, logp = ac.actor(obs)
    # same draw, explicit formula: log N(u; mu, std) - sum log(1 - tanh(u)^2)
    h = ac.actor.net(obs)
    mu, log_std = ac.actor.mu_layer(h), ac.actor.log_std_layer(h).clamp(-20, 2)
    torch.manual_seed(5)
    u = mu + log_std.exp() * torch.randn_like(mu)
    ref = torch.distributions.Normal(mu, log_std.exp()).log_prob(u).sum(-1, keepdim=True) \
        - torch.log(1 - torch.tanh(u) ** 2 + 1e-12).sum(-1, keepdim=True)
    torch.testing.assert_close(act, torch.tanh(u))
    torch.testing.assert_close(logp, ref, rtol=1e-4, atol=1e-4)
    det, none = ac.actor(obs, deterministic=True, with_logprob=False)
    assert none is None
    torch.testing.assert_close(det, torch.tanh(mu))


def test_action_rescaling_to_space_bounds():
    ac = MLPActorCritic(4, 1, [-10.0], [10.0], [8, 8], 'relu')
    a = ac.act(torch.randn(128, 4))
    assert a.shape == (128, 1) and float(a.min()) >= -10.0 and float(a.max()) <= 10.0


def test_losses_and_update_follow_reference_formulas():
    torch.manual_seed(1)
    cfg = SACConfig(hidden_dim=16, use_entropy_tuning=True, tau=0.1)
    agent = SACAgent(6, 2, [-1.0, -1.0], [1.0, 1.0], cfg, torch.device('cpu'))
    b = {'obs': torch.randn(32, 6), 'act': torch.rand(32, 2) * 2 - 1, 'rew': torch.randn(32, 1),
         'next_obs': torch.randn(32, 6), 'mask': (torch.rand(32, 1) > 0.2).float()}
    torch.manual_seed(3)
    loss = agent.q_loss(b)
    torch.manual_seed(3)
    with torch.no_grad():
        na, nlp = agent.ac.actor(b['next_obs'])
        nq = torch.min(agent.ac_targ.q1(b['next_obs'], na), agent.ac_targ.q2(b['next_obs'], na))
        targ = b['rew'] + 0.99 * b['mask'] * (nq - agent.alpha * nlp)
    ref = (agent.ac.q1(b['obs'], b['act']) - targ).pow(2).mean() + (agent.ac.q2(b['obs'], b['act']) - targ).pow(2).mean()
    torch.testing.assert_close(loss, ref)
    targ0 = [p.clone() for p in agent.ac_targ.parameters()]
    la0 = float(agent.log_alpha)
    res = agent.update(b)
    assert set(res) == {'policy_loss', 'critic_loss', 'entropy_loss'}
    assert float(agent.log_alpha) != la0
    for p, pt, p0 in zip(agent.ac.parameters(), agent.ac_targ.parameters(), targ0):
        torch.testing.assert_close(pt, 0.9 * p0 + 0.1 * p.detach())
    assert abs(float(agent.alpha) - math.exp(float(agent.log_alpha))) < 1e-7


def test_device_replay_ring_semantics():
    buf = DeviceReplay(10, 3, 1, torch.device('cpu'))
    for k in range(4):
        n = 4
        buf.push(torch.full((n, 3), float(k)), torch.zeros(n, 1), torch.full((n,), float(k)), torch.zeros(n, 3), torch.ones(n))
    assert buf.size == 10 and buf.pos == 6
    # the ring now holds steps {1 (2 rows), 2, 3} -> rewards in {1, 2, 3} only... plus wrapped rows of step 3 and 2
    assert set(buf.rew[:, 0].tolist()) <= {1.0, 2.0, 3.0}
    s = buf.sample(256)
    assert s['obs'].shape == (256, 3) and s['mask'].shape == (256, 1)


def test_shipped_reference_sac_actor_loads_and_acts():
    """The reference's SAC checkpoint layout (actor.net.fcs.*, mu_layer, log_std_layer) loads into sac.MLPActorCritic;
    the deterministic action equals a NumPy forward pass with tanh squashing onto the action bounds."""
    import os
    import numpy as np
    import torch
    from safe_control_gym_amd.sac import MLPActorCritic
    f = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'sac_actor_cartpole_stab.npz'))
    sd = {k: torch.as_tensor(f[k]) for k in f.files if k.startswith('actor.')}
    low, high = torch.tensor([-1.0]), torch.tensor([1.0])            # normalised RL action space of cartpole_stab
    ac = MLPActorCritic(4, 1, low, high, [256, 256], 'relu')
    missing, unexpected = ac.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.startswith(('q1.', 'q2.', 'actor.low', 'actor.high')) for m in missing)
    obs = np.asarray(f['obs'], dtype=np.float32)
    h = obs
    for i in (0, 1):
        h = h @ f[f'actor.net.fcs.{i}.weight'].T + f[f'actor.net.fcs.{i}.bias']
        if i == 0:
            h = np.maximum(h, 0.0)               # upstream MLP: activation between layers, none after the last
    mu = h @ f['actor.mu_layer.weight'].T + f['actor.mu_layer.bias']
    ref = -1.0 + 0.5 * (np.tanh(mu) + 1.0) * 2.0
    got = ac.act(torch.as_tensor(obs), deterministic=True).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)
    assert np.all(np.abs(got) <= 1.0)



class _SlicedNoise:
    """torch.randn_like stand-in: call k returns rows [lo, hi) of the k-th seeded [256, d] draw, so that two ranks holding
    half a batch each see exactly the noise one process holding the whole batch sees."""

    def __init__(self, lo, hi):
        self.lo, self.hi, self.k = lo, hi, 0

    def __call__(self, like):
        g = torch.Generator().manual_seed(1000 + self.k)
        self.k += 1
        return torch.randn(256, like.shape[1], generator=g)[self.lo:self.hi]


def _sac_batch():
    g = torch.Generator().manual_seed(3)
    return {'obs': torch.randn(256, 6, generator=g), 'act': torch.rand(256, 2, generator=g) * 2 - 1, 'rew': torch.randn(256, 1, generator=g),
            'next_obs': torch.randn(256, 6, generator=g), 'mask': (torch.rand(256, 1, generator=g) > 0.2).float()}


def _sac_run(lo, hi, seed):
    from unittest import mock
    torch.manual_seed(seed)
    agent = SACAgent(6, 2, [-1.0, -1.0], [1.0, 1.0], SACConfig(hidden_dim=16, use_entropy_tuning=True, actor_lr=1e-2, critic_lr=1e-2, entropy_lr=1e-2),
                     torch.device('cpu'))
    local = {k: v[lo:hi] for k, v in _sac_batch().items()}
    with mock.patch.object(torch, 'randn_like', _SlicedNoise(lo, hi)):
        for _ in range(3):
            agent.update(local)
    return torch.cat([p.detach().reshape(-1) for p in agent.ac.parameters()] + [agent.log_alpha.detach().reshape(-1)]
                     + [p.detach().reshape(-1) for p in agent.ac_targ.parameters()])


def _sac_dp_worker(rank, world, port, out_path):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    flat = _sac_run(rank * 128, (rank + 1) * 128, seed=50 + rank)      # different local init: overwritten by rank 0's broadcast
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        torch.save(gathered, out_path)
    dist.barrier()                                  # nobody tears its sockets down while a peer is still busy
    dist.destroy_process_group()


def test_data_parallel_sac_update_matches_single_process(tmp_path):
    """2 gloo ranks x 128 samples, flat all-reduce of actor / critic / temperature gradients == 1 process x 256 samples
    (three consecutive updates, entropy tuning on, Polyak targets included)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = str(tmp_path / 'sac_dp.pt')
    mp.spawn(_sac_dp_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    torch.testing.assert_close(got[0], got[1], rtol=0, atol=0)
    ref = _sac_run(0, 256, seed=50)
    torch.testing.assert_close(got[0], ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('device', DEVICES)
@pytest.mark.parametrize('norm', [False, True], ids=['plain', 'normalised'])
def test_sac_collector_reproduces_the_reference_buffer(device, norm):
    """sac.SAC.train_step against the REFERENCE's own `SAC.train_step` (controllers/sac/sac.py:269-335; tests/golden/
    make_sac_collector.py): the recorded transitions of 4 envs x 40 vector steps are replayed (tests/replay_env.py) with the actions the
    reference fed, and the replay ring must hold what the reference's SACBuffer holds — obs, act, rew and the TRUE next_obs / mask of the
    time-limit fix-up (12 truncated episodes store their terminal observation with mask 1, 8 terminated ones the post-reset observation
    with mask 0), in ring order after the wrap (160 pushes into 120 slots).
    normalised: the reference ran with `norm_obs: True, norm_reward: True` (sac.py:75-81) — the ring holds NORMALISED observations /
    rewards (statistics updated by next_obs, then the reward's running returns with upstream's index-array reset, then the truncated
    envs' terminal observations), and the normalisers' final mean / var / count / running returns equal the reference's.
    cuda: the same on device tensors (the replay ring, the normalisers' float64 statistics and the masked update live in HBM)."""
    import os

    import numpy as np
    from safe_control_gym_amd.sac import SAC
    from tests.replay_env import ReplayVecEnv, spec_for
    G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'sac_collector_norm.npz' if norm else 'sac_collector.npz'))
    tr = {k: G[f'transitions/{k}'] for k in ('act', 'next_obs', 'rew', 'done', 'trunc', 'term_obs')}
    env = ReplayVecEnv(spec_for(dict(episode_len_sec=0.2, randomized_init=True, done_on_out_of_bound=True)), device, G['obs0'],
                       tr['next_obs'], tr['rew'], tr['done'], tr['trunc'], tr['term_obs'])
    extra = {'cuda_graphs': False}
    if norm:
        extra.update(norm_obs=True, norm_reward=True, clip_obs=4.0, clip_reward=2.0)
    cfg = SACConfig(hidden_dim=16, activation='relu', rollout_batch_size=4, warm_up_steps=0, train_interval=10 ** 9, max_buffer_size=120,
                    extra=extra)
    sac = SAC(env, cfg, seed=0)
    acts = torch.as_tensor(tr['act'], dtype=torch.float32, device=device)
    sac.agent.ac.act = lambda obs, deterministic=False: acts[env.t % env.T]      # the actions the reference fed (warm-up draws + its samples)
    for _ in range(acts.shape[0]):
        res = sac.train_step()
        assert 'updates' not in res
    assert sac.total_steps == int(G['total_steps']) and [sac.buffer.pos, sac.buffer.size] == G['buffer/pos_size'].tolist()
    torch.testing.assert_close(env.seen_act, acts, rtol=0, atol=0)
    for k in ('obs', 'act', 'rew', 'next_obs', 'mask'):
        got = getattr(sac.buffer, k).cpu().numpy().reshape(G[f'buffer/{k}'].shape)
        np.testing.assert_allclose(got, G[f'buffer/{k}'], rtol=0, atol=1e-6, err_msg=k)
    m = G['buffer/mask'].reshape(-1)
    assert (m == 0).sum() > 0 and (tr['trunc'].sum() > 0)                       # the fixture holds both kinds of episode end
    if norm:
        o, r = sac.obs_normalizer, sac.reward_normalizer
        kw = dict(rtol=5e-6, atol=1e-7)             # (the collector's observations / rewards are float32; the reference's were float64)
        np.testing.assert_allclose(o.rms.mean.cpu().numpy(), G['norm/obs_mean'], **kw)
        np.testing.assert_allclose(o.rms.var.cpu().numpy(), G['norm/obs_var'], **kw)
        np.testing.assert_allclose(float(o.rms.count), float(G['norm/obs_count']), rtol=1e-12)
        np.testing.assert_allclose(float(r.rms.var), float(G['norm/rew_var']), **kw)
        np.testing.assert_allclose(float(r.rms.count), float(G['norm/rew_count']), rtol=1e-12)
        np.testing.assert_allclose(r.ret.cpu().numpy(), G['norm/ret'], **kw)
        np.testing.assert_allclose(sac.obs.cpu().numpy(), G['final_obs'], rtol=0, atol=1e-6)
        assert np.abs(G['buffer/obs']).max() <= 4.0 + 1e-6 and np.abs(G['buffer/rew']).max() <= 2.0 + 1e-6

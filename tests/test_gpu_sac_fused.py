"""scg_sac_update (csrc/scg_sac.hip, include/scg_sac.h): the fused SAC gradient step against
 (a) the REFERENCE's own SACAgent.update — tests/golden/learner.npz, produced by tests/golden/make_learner.py from
     controllers/sac/sac_utils.py: same initial weights, same index batches, same noise -> the reference's final weights,
     target weights, temperature and loss statistics after three updates;
 (b) this package's eager PyTorch update (itself pinned to the reference by tests/test_learner_golden.py) at the production
     shape (24 -> 128 -> 128, 4 actions, batch 4096): gradients of one step and parameters after several;
 (c) itself: bitwise reproducibility, in-kernel sampling, HIP-graph capture through SACAgent.update_from_buffer."""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'learner.npz'))


def sd(prefix):
    return {k[len(prefix) + 1:]: torch.as_tensor(G[k]) for k in G.files if k.startswith(prefix + '/')}


class RecordNoise:
    """Context: records every torch.randn_like draw (the actor's reparameterisation noise) / replays a given list."""

    def __init__(self, replay=None):
        self.draws, self.replay, self._orig = [], list(replay) if replay is not None else None, None

    def __enter__(self):
        self._orig = torch.randn_like

        def fn(t, *a, **k):
            x = self.replay.pop(0).to(t.device) if self.replay is not None else self._orig(t, *a, **k)
            self.draws.append(x.detach().cpu().clone())
            return x
        torch.randn_like = fn
        return self

    def __exit__(self, *exc):
        torch.randn_like = self._orig


def _ring(buf_cls, dev):
    rng = np.random.default_rng(9)
    buf = buf_cls(200, 6, 2, dev)
    for _ in range(5):
        n = 50
        b = {'obs': rng.normal(0, 1, (n, 6)), 'act': rng.uniform([-1, 0], [1, 2], (n, 2)), 'rew': rng.normal(0, 1, (n,)),
             'next_obs': rng.normal(0, 1, (n, 6)), 'mask': (rng.uniform(size=n) > 0.1).astype(np.float32)}
        t = {k: torch.as_tensor(v, dtype=torch.float32, device=dev) for k, v in b.items()}
        buf.push(t['obs'], t['act'], t['rew'], t['next_obs'], t['mask'])
    return buf


def test_fused_update_reproduces_the_reference_sac_agent():
    from safe_control_gym_amd.sac import DeviceReplay, SACAgent, SACConfig
    kw = dict(hidden_dim=32, activation='relu', gamma=0.98, tau=0.01, init_temperature=0.3, use_entropy_tuning=True, actor_lr=1e-3,
              critic_lr=2e-3, entropy_lr=3e-3)
    # the noise the reference drew: replay its update on the eager CPU path under the generator's seed and record it
    cpu = SACAgent(6, 2, torch.tensor([-1.0, 0.0]), torch.tensor([1.0, 2.0]), SACConfig(**kw, extra={'cuda_graphs': False}), 'cpu')
    cpu.ac.load_state_dict(sd('sac/init'), strict=False); cpu.ac_targ.load_state_dict(sd('sac/init'), strict=False)
    cbuf = _ring(DeviceReplay, 'cpu')
    torch.manual_seed(13)
    with RecordNoise() as rec:
        for idx in G['sac/indices']:
            idx = torch.as_tensor(idx)
            cpu.update({k: getattr(cbuf, k)[idx] for k in ('obs', 'act', 'rew', 'next_obs', 'mask')})
    assert len(rec.draws) == 6
    dev = torch.device('cuda', 0)
    ag = SACAgent(6, 2, torch.tensor([-1.0, 0.0], device=dev), torch.tensor([1.0, 2.0], device=dev), SACConfig(**kw), dev)
    assert ag.use_fused
    ag.ac.load_state_dict(sd('sac/init'), strict=False); ag.ac_targ.load_state_dict(sd('sac/init'), strict=False)
    buf = _ring(DeviceReplay, dev)
    res = []
    for k, idx in enumerate(G['sac/indices']):
        F = ag._fused_args(buf, 64, idx=torch.as_tensor(idx, dtype=torch.int32, device=dev), eps=rec.draws[2 * k].to(dev).contiguous(),
                           eps_next=rec.draws[2 * k + 1].to(dev).contiguous())
        ag._fused_step(F)
        torch.cuda.synchronize()
        res.append(F['stats'][:3].tolist())
    np.testing.assert_allclose(res, G['sac/results'], rtol=5e-5, atol=5e-6)
    for prefix, net in (('sac/final', ag.ac), ('sac/final_targ', ag.ac_targ)):
        final = sd(prefix)
        for k, v in net.state_dict().items():
            if k in final:
                torch.testing.assert_close(v.cpu(), final[k], rtol=2e-4, atol=5e-6, msg=lambda m, k=k: f'{prefix} {k}: {m}')
    np.testing.assert_allclose(float(ag.log_alpha), float(G['sac/final_log_alpha']), rtol=1e-5)
    assert ag._flat['steps'].tolist() == [3.0, 3.0, 3.0] and int(ag._flat['counter']) == 3


@pytest.mark.parametrize('tuning', [False, True])
def test_fused_update_equals_the_eager_update_at_the_production_shape(tuning):
    _fused_vs_eager(24, 4, 128, 'relu', tuning, 4096)


# shapes no shipped task has: three feature tiles (hidden 96), input widths that are not multiples of 4 / 8, obs + act = 31 (the
# widest Q input the library serves), one-float observations, three actions; batches of 1 to 20 tiles, and one above the 512-workgroup cap
@pytest.mark.parametrize('obs_dim,act_dim,hidden,act,B', [(7, 1, 96, 'relu', 256), (17, 2, 96, 'tanh', 640), (27, 4, 64, 'leaky_relu', 32),
                                                         (1, 1, 32, 'tanh', 96), (29, 2, 128, 'relu', 512), (3, 3, 32, 'relu', 224),
                                                         (12, 2, 64, 'relu', 20480)])       # 640 tiles on 512 workgroups: the multi-tile loops
def test_fused_update_equals_the_eager_update_on_other_shapes(obs_dim, act_dim, hidden, act, B):
    _fused_vs_eager(obs_dim, act_dim, hidden, act, bool(obs_dim % 2), B)


def _fused_vs_eager(nobs, nu, hidden, act, tuning, B):
    from safe_control_gym_amd.sac import DeviceReplay, SACAgent, SACConfig
    dev = torch.device('cuda', 0)
    low, high = -torch.ones(nu, device=dev), torch.ones(nu, device=dev)
    kw = dict(hidden_dim=hidden, activation=act, use_entropy_tuning=tuning, actor_lr=1e-3, critic_lr=1e-3, entropy_lr=1e-3)
    torch.manual_seed(3)
    eager = SACAgent(nobs, nu, low, high, SACConfig(**kw, extra={'cuda_graphs': False}), dev)
    fused = SACAgent(nobs, nu, low, high, SACConfig(**kw), dev)
    assert fused.use_fused and not eager.use_fused
    fused.ac.load_state_dict(eager.ac.state_dict()); fused.ac_targ.load_state_dict(eager.ac_targ.state_dict())
    cap = 20000
    buf = DeviceReplay(cap, nobs, nu, dev)
    g = torch.Generator(device=dev).manual_seed(5)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)                     # noqa: E731
    buf.push(r(cap, nobs), torch.tanh(r(cap, nu)), r(cap), r(cap, nobs), (torch.rand(cap, device=dev, generator=g) > 0.05).float())
    for step in range(3):
        idx = torch.randint(0, cap, (B,), device=dev, generator=g)
        eps, eps2 = r(B, nu), r(B, nu)
        batch = {k: getattr(buf, k)[idx] for k in ('obs', 'act', 'rew', 'next_obs', 'mask')}
        # reference gradients of this step from autograd on the eager agent's CURRENT weights
        with RecordNoise(replay=[eps, eps2]):
            res_e = eager.update(batch)
        ga = torch.cat([p.grad.reshape(-1) for p in eager.ac.actor.parameters()])
        F = fused._fused_args(buf, B, idx=idx.to(torch.int32), eps=eps, eps_next=eps2)
        fused._fused_step(F)
        torch.cuda.synchronize()
        st = F['stats'].tolist()
        np.testing.assert_allclose(st[:2], [float(res_e['policy_loss']), float(res_e['critic_loss'])], rtol=2e-4, atol=1e-5)
        if step == 0:       # same weights on both sides: gradients comparable element by element
            fl = fused._flat
            names = ['net.fcs.0.weight', 'net.fcs.0.bias', 'net.fcs.1.weight', 'net.fcs.1.bias', 'mu_layer.weight', 'log_std_layer.weight',
                     'mu_layer.bias', 'log_std_layer.bias']
            pe = dict(eager.ac.actor.named_parameters())
            ref = torch.cat([pe[n].grad.reshape(-1) for n in names])
            got = fl['g'][:fl['n_actor']]
            torch.testing.assert_close(got, ref, rtol=2e-3, atol=2e-6)
            qe = torch.cat([p.grad.reshape(-1) for q in (eager.ac.q1, eager.ac.q2) for p in q.parameters()])
            torch.testing.assert_close(fl['g'][fl['n_actor']:fl['n']], qe, rtol=2e-3, atol=2e-6)
            assert ga.abs().max() > 0
    # after three Adam steps of lr 1e-3 the two parameter sets stay together (Adam's first steps are sign-like: elements with
    # |g| ~ 1e-8 may differ by a whole lr, hence the absolute tolerance of two steps)
    for (k, a), b in zip(fused.ac.state_dict().items(), eager.ac.state_dict().values()):
        assert (a - b).abs().max() <= 2.1e-3, k
        assert (a - b).abs().mean() <= 2e-5, k
    for (k, a), b in zip(fused.ac_targ.state_dict().items(), eager.ac_targ.state_dict().values()):
        assert (a - b).abs().max() <= 1e-4, k
    if tuning:
        assert abs(float(fused.log_alpha) - float(eager.log_alpha)) < 1e-5


def test_fused_update_samples_in_the_kernel_is_reproducible_and_learns():
    from safe_control_gym_amd.sac import DeviceReplay, SACAgent, SACConfig
    dev = torch.device('cuda', 0)
    low, high = -torch.ones(4, device=dev), torch.ones(4, device=dev)

    def run(n_calls):
        torch.manual_seed(11)
        ag = SACAgent(24, 4, low, high, SACConfig(hidden_dim=128, activation='relu'), dev)
        buf = DeviceReplay(50000, 24, 4, dev)
        g = torch.Generator(device=dev).manual_seed(2)
        obs = torch.randn(30000, 24, device=dev, generator=g)
        act = torch.rand(30000, 4, device=dev, generator=g) * 2 - 1
        # a learnable signal: reward = -|a - tanh(obs[:4])|^2, episodes end immediately (mask 0): Q -> reward, actor -> tanh(obs[:4])
        rew = -((act - torch.tanh(obs[:, :4])) ** 2).sum(-1)
        buf.push(obs, act, rew, torch.randn(30000, 24, device=dev, generator=g), torch.zeros(30000, device=dev))
        out = []
        for _ in range(n_calls):
            out.append(ag.update_from_buffer(buf, 4096, 8))         # 8 gradient steps per call, one captured graph
        return ag, out, obs

    a1, o1, obs = run(40)
    a2, o2, _ = run(40)
    assert o1 == o2                                                             # bitwise: fixed-order reductions, counter-based draws
    assert torch.equal(a1._flat['p'], a2._flat['p']) and int(a1._flat['counter']) == 320 and a1._flat['steps'].tolist()[:2] == [320.0, 320.0]
    assert o1[-1]['critic_loss'] < 0.2 * o1[0]['critic_loss']
    with torch.no_grad():
        a = a1.ac.act(obs[:2048], deterministic=True)
        err0 = ((torch.zeros_like(a) - torch.tanh(obs[:2048, :4])) ** 2).sum(-1).mean()
        err = ((a - torch.tanh(obs[:2048, :4])) ** 2).sum(-1).mean()
    assert err < 0.5 * err0, (float(err), float(err0))
    # the batched deterministic actor of the library == the torch module on the same flat weights
    import ctypes as C
    from safe_control_gym_amd import _sac
    D = _sac.lib(24, 128, 4, 'relu')
    out = torch.empty(2048, 4, device=dev)
    lo, hi = (C.c_float * 4)(-1, -1, -1, -1), (C.c_float * 4)(1, 1, 1, 1)
    _sac.check(D, D.scg_sac_act(a1._flat['p'].data_ptr(), C.byref(a1._flat['actor']), lo, hi, obs[:2048].contiguous().data_ptr(), 2048,
                                out.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    torch.testing.assert_close(out, a, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('nobs,nu,hidden,act,B', [(24, 4, 128, 'relu', 4096), (17, 2, 96, 'tanh', 640), (6, 2, 32, 'relu', 64), (12, 2, 64, 'relu', 20480)])
def test_update_n_is_bit_identical_to_single_step_calls(nobs, nu, hidden, act, B):
    """scg_sac_update_n (what SACAgent.update_from_buffer captures: step k + 1's first launch — rows drawn, a, log pi at obs, the tiles
    actor_grad_kernel reads back — rides in step k's target-action launch, 7 n + 1 launches) against n scg_sac_update calls (8 launches
    each) from the same state: parameters, target copy, Adam moments, step counts, Philox counter and the loss sums, bit for bit — with
    entropy tuning on (log_alpha moves between the two jobs' reads) and an odd n (both copies of the row / log_alpha buffers in use)."""
    from safe_control_gym_amd.sac import DeviceReplay, SACAgent, SACConfig
    dev = torch.device('cuda', 0)
    low, high = -torch.ones(nu, device=dev), torch.ones(nu, device=dev)
    out = []
    for mode in ('n', 'single'):
        torch.manual_seed(4)
        ag = SACAgent(nobs, nu, low, high, SACConfig(hidden_dim=hidden, activation=act, use_entropy_tuning=True), dev)
        assert ag.use_fused
        buf = DeviceReplay(20000, nobs, nu, dev)
        g = torch.Generator(device=dev).manual_seed(7)
        n = 9000
        buf.push(torch.randn(n, nobs, device=dev, generator=g), torch.rand(n, nu, device=dev, generator=g) * 2 - 1, torch.randn(n, device=dev, generator=g),
                 torch.randn(n, nobs, device=dev, generator=g), (torch.rand(n, device=dev, generator=g) < 0.8).float())
        if mode == 'n':
            stats = [ag.update_from_buffer(buf, B, 5) for _ in range(3)]           # three replays of the captured 5-step graph
            acc = None
        else:
            F = ag._fused_args(buf, B)
            stats = []
            for _ in range(3):
                F['acc'].zero_()
                for _ in range(5):
                    ag._fused_step(F)
                stats.append(dict(zip(('policy_loss', 'critic_loss', 'entropy_loss'), (F['acc'] / 5).tolist()[:3])))
        torch.cuda.synchronize()
        fl = ag._flat
        out.append(({k: fl[k].clone() for k in ('p', 'targ', 'm', 'v', 'steps', 'counter')}, stats))
    (a, sa), (b, sb) = out
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert int(a['counter']) == 15 and a['steps'].tolist() == [15.0, 15.0, 15.0]
    assert sa == sb


def _sac_variant_names():
    from tests.golden.learner_cases import SAC_CASES
    return sorted(SAC_CASES)


@pytest.mark.parametrize('name', _sac_variant_names())
def test_fused_update_reproduces_the_reference_sac_agent_in_the_corners(name):
    """tests/golden/learner_variants.npz (the reference's SACAgent.update with a fixed temperature / tanh trunk / one-sided action
    interval, and with a tuned temperature / four actions): four fused steps on the reference's index batches and noise."""
    from safe_control_gym_amd.sac import DeviceReplay, SACAgent, SACConfig
    from tests.golden.learner_cases import SAC_CASES
    c = SAC_CASES[name]
    p = f'sac/{name}'
    V = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'learner_variants.npz'))
    vsd = lambda pre: {k[len(pre) + 1:]: torch.as_tensor(V[k]) for k in V.files if k.startswith(pre + '/')}       # noqa: E731
    nu = len(c['low'])

    def fill(dev):
        buf = DeviceReplay(V[p + '/buffer/obs'].shape[0], c['obs'], nu, dev)
        t = {k: torch.as_tensor(V[f'{p}/buffer/{k}'], dtype=torch.float32, device=dev) for k in ('obs', 'act', 'rew', 'next_obs', 'mask')}
        buf.push(t['obs'], t['act'], t['rew'].reshape(-1), t['next_obs'], t['mask'].reshape(-1))
        return buf
    cpu = SACAgent(c['obs'], nu, torch.tensor(c['low']), torch.tensor(c['high']), SACConfig(**c['kw'], extra={'cuda_graphs': False}), 'cpu')
    cpu.ac.load_state_dict(vsd(p + '/init'), strict=False); cpu.ac_targ.load_state_dict(vsd(p + '/init'), strict=False)
    cbuf = fill('cpu')
    torch.manual_seed(29)
    with RecordNoise() as rec:                      # the noise the reference drew (its generator's seed on the eager CPU path)
        for idx in V[p + '/indices']:
            idx = torch.as_tensor(idx)
            cpu.update({k: getattr(cbuf, k)[idx] for k in ('obs', 'act', 'rew', 'next_obs', 'mask')})
    dev = torch.device('cuda', 0)
    ag = SACAgent(c['obs'], nu, torch.tensor(c['low'], device=dev), torch.tensor(c['high'], device=dev), SACConfig(**c['kw']), dev)
    assert ag.use_fused
    ag.ac.load_state_dict(vsd(p + '/init'), strict=False); ag.ac_targ.load_state_dict(vsd(p + '/init'), strict=False)
    buf = fill(dev)
    res = []
    for k, idx in enumerate(V[p + '/indices']):
        F = ag._fused_args(buf, 64, idx=torch.as_tensor(idx, dtype=torch.int32, device=dev), eps=rec.draws[2 * k].to(dev).contiguous(),
                           eps_next=rec.draws[2 * k + 1].to(dev).contiguous())
        ag._fused_step(F)
        torch.cuda.synchronize()
        res.append(F['stats'][:3].tolist())
    np.testing.assert_allclose(res, V[p + '/results'], rtol=1e-4, atol=1e-5)
    for prefix, net in ((p + '/final', ag.ac), (p + '/final_targ', ag.ac_targ)):
        final = vsd(prefix)
        for k, v in net.state_dict().items():
            if k in final:
                torch.testing.assert_close(v.cpu(), final[k], rtol=2e-4, atol=1e-5, msg=lambda m, k=k: f'{prefix} {k}: {m}')
    np.testing.assert_allclose(float(ag.log_alpha.detach()), float(V[p + '/final_log_alpha']), rtol=1e-5)


def test_adam_state_travels_between_the_fused_and_the_torch_layout():
    """Checkpoints carry the Adam moments in torch.optim's per-parameter layout whichever update produced them: a fused agent's
    checkpoint resumes in an agent that steps torch.optim (and in the reference's SACAgent, sac_utils.py:85-108), and torch-only
    checkpoints — the reference's own training checkpoints — resume in the fused update (scattered into the flat m / v / steps)."""
    from safe_control_gym_amd.sac import DeviceReplay, SACAgent, SACConfig
    dev = torch.device('cuda', 0)
    low, high = -torch.ones(2, device=dev), torch.ones(2, device=dev)
    kw = dict(hidden_dim=64, activation='relu', use_entropy_tuning=True, actor_lr=1e-3, critic_lr=2e-3, entropy_lr=1e-3)
    torch.manual_seed(4)
    fused = SACAgent(6, 2, low, high, SACConfig(**kw), dev)
    buf = DeviceReplay(4096, 6, 2, dev)
    g = torch.Generator(device=dev).manual_seed(5)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)                     # noqa: E731
    buf.push(r(4096, 6), torch.tanh(r(4096, 2)), r(4096), r(4096, 6), (torch.rand(4096, device=dev, generator=g) > 0.05).float())
    fused.update_from_buffer(buf, 256, 5)
    torch.cuda.synchronize()
    sd_f = fused.state_dict()
    assert set(sd_f['flat_adam']) == {'counter'} and len(sd_f['actor_opt']['state']) == 8 and len(sd_f['critic_opt']['state']) == 12
    # (1) -> an agent that steps torch.optim: per-parameter moments == the flat ones, step counts carried
    eager = SACAgent(6, 2, low, high, SACConfig(**kw, extra={'cuda_graphs': False}), dev)
    eager.load_state_dict(sd_f)
    fl = fused._flat
    for opt_f, opt_e in ((fused.actor_opt, eager.actor_opt), (fused.critic_opt, eager.critic_opt), (fused.alpha_opt, eager.alpha_opt)):
        for pf, pe in zip(opt_f.param_groups[0]['params'], opt_e.param_groups[0]['params']):
            o, k = fused._flat_offset(pf), pf.numel()
            assert torch.equal(opt_e.state[pe]['exp_avg'].reshape(-1), fl['m'][o:o + k])
            assert torch.equal(opt_e.state[pe]['exp_avg_sq'].reshape(-1), fl['v'][o:o + k])
            assert float(opt_e.state[pe]['step']) == 5.0
            assert torch.equal(pe.data, pf.data)
    # (2) torch-only checkpoint (what the reference / a non-fused agent saves) -> the fused update's flat buffers
    sd_e = eager.state_dict()
    assert 'flat_adam' not in sd_e
    again = SACAgent(6, 2, low, high, SACConfig(**kw), dev)
    again.load_state_dict(sd_e)
    assert again.use_fused and again._flat['steps'].tolist() == [5.0, 5.0, 5.0]
    n = fl['n']
    assert torch.equal(again._flat['m'][:n + 1], fl['m'][:n + 1]) and torch.equal(again._flat['v'][:n + 1], fl['v'][:n + 1])
    assert torch.equal(again._flat['p'], fl['p']) and torch.equal(again._flat['targ'], fl['targ'])
    # ... and it continues exactly where the first agent continues (same sampling counter once that is carried too)
    again._flat['counter'].copy_(fl['counter'])
    fused.update_from_buffer(buf, 256, 3); again.update_from_buffer(buf, 256, 3)
    torch.cuda.synchronize()
    assert torch.equal(again._flat['p'], fused._flat['p'])


def test_fused_collector_pieces_equal_the_torch_collector():
    """scg_sac_sample / scg_sac_push (the collector's device-side pieces, SAC._collect_body_fused) against the PyTorch collector they
    replace: (1) the sampled action with the SAME noise equals MLPActor.forward's; the in-kernel noise is N(0, 1) and differs from
    step to step; the warm-up action is U[low, high); (2) one vector step pushed by the kernel == DeviceReplay.push_device of the
    fixed-up batch (terminal observation and mask 1 where truncated), across a ring wrap, incl. position / size words and the current
    observation; (3) SAC.train_step on the fused collector fills the ring consistently with the env's own outputs."""
    import ctypes as C
    from safe_control_gym_amd import _sac
    from safe_control_gym_amd.sac import SAC, DeviceReplay, SACAgent, SACConfig
    dev = torch.device('cuda', 0)
    nobs, nu, H = 24, 4, 128
    low, high = torch.tensor([-1.0, -0.5, 0.0, 0.1], device=dev), torch.tensor([1.0, 0.5, 2.0, 0.3], device=dev)
    torch.manual_seed(3)
    ag = SACAgent(nobs, nu, low, high, SACConfig(hidden_dim=H, activation='relu'), dev)
    assert ag.use_fused
    fl = ag._flat
    D = _sac.lib(nobs, H, nu, 'relu')
    lo, hi = (C.c_float * 4)(*low.tolist()), (C.c_float * 4)(*high.tolist())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    m = 1000                                                    # (ragged last tile)
    obs = torch.randn(m, nobs, device=dev)
    eps = torch.randn(m, nu, device=dev)
    out = torch.empty(m, nu, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    _sac.check(D, D.scg_sac_sample(fl['p'].data_ptr(), C.byref(fl['actor']), lo, hi, obs.data_ptr(), m, 77, cnt.data_ptr(), 0, eps.data_ptr(), out.data_ptr(), st))
    a = ag.ac.actor
    with torch.no_grad():
        h = a.net(obs)
        u = a.mu_layer(h) + torch.clamp(a.log_std_layer(h), -20, 2).exp() * eps
        ref = low + 0.5 * (torch.tanh(u) + 1.0) * (high - low)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=2e-6)
    # in-kernel noise: implied eps = (atanh(2 (a - low) / (high - low) - 1) - mu) / sigma is standard normal, fresh per counter value
    big = torch.randn(65536, nobs, device=dev) * 0.1
    outs = []
    for c in (0, 1):
        cnt.fill_(c)
        o = torch.empty(65536, nu, device=dev)
        _sac.check(D, D.scg_sac_sample(fl['p'].data_ptr(), C.byref(fl['actor']), lo, hi, big.data_ptr(), 65536, 77, cnt.data_ptr(), 0, None, o.data_ptr(), st))
        outs.append(o)
    with torch.no_grad():
        h = a.net(big)
        mu, sg = a.mu_layer(h), torch.clamp(a.log_std_layer(h), -20, 2).exp()
        e0 = (torch.atanh((2 * (outs[0] - low) / (high - low) - 1).clamp(-1 + 1e-6, 1 - 1e-6)) - mu) / sg
    assert abs(float(e0.mean())) < 0.01 and abs(float(e0.std()) - 1.0) < 0.01 and abs(float((e0 ** 3).mean())) < 0.05
    assert float((outs[0] - outs[1]).abs().mean()) > 1e-3      # another counter word, other draws
    uo = torch.empty(65536, nu, device=dev)
    _sac.check(D, D.scg_sac_sample(None, None, lo, hi, None, 65536, 77, cnt.data_ptr(), 1, None, uo.data_ptr(), st))
    assert bool((uo >= low).all()) and bool((uo < high + 1e-6).all())
    torch.testing.assert_close(uo.mean(0), 0.5 * (low + high), rtol=0, atol=0.01 * float((high - low).max()))
    # ---- (2) ring push
    n, cap = 300, 700
    b_k, b_t = DeviceReplay(cap, nobs, nu, dev), DeviceReplay(cap, nobs, nu, dev)
    cur_k = torch.randn(n, nobs, device=dev)
    cur_t = cur_k.clone()
    ring = _sac.SacRing(d_obs=b_k.obs.data_ptr(), d_act=b_k.act.data_ptr(), d_rew=b_k.rew.data_ptr(), d_next_obs=b_k.next_obs.data_ptr(),
                        d_mask=b_k.mask.data_ptr(), capacity=cap, d_pos=b_k.pos_t.data_ptr(), d_size_f=b_k.size_t.data_ptr(),
                        d_size_i32=b_k.size_i32.data_ptr(), d_counter=cnt.data_ptr())
    cnt.zero_()
    g = torch.Generator(device='cpu').manual_seed(4)
    for step in range(4):                                       # 1200 rows into 700 slots: wraps
        act = torch.randn(n, nu, generator=g).to(dev)
        rew, nxt, term = torch.randn(n, generator=g).to(dev), torch.randn(n, nobs, generator=g).to(dev), torch.randn(n, nobs, generator=g).to(dev)
        done = (torch.rand(n, generator=g) < 0.3).to(torch.uint8).to(dev)
        flags = (torch.randint(0, 4, (n,), generator=g)).to(torch.uint8).to(dev)          # bit 0 = truncated
        _sac.check(D, D.scg_sac_push(C.byref(ring), cur_k.data_ptr(), act.data_ptr(), rew.data_ptr(), nxt.data_ptr(), term.data_ptr(),
                                     done.data_ptr(), flags.data_ptr(), n, st))
        trunc = (flags & 1).bool() & done.bool()
        b_t.push_device(cur_t, act, rew, torch.where(trunc[:, None], term, nxt), torch.where(trunc, torch.ones_like(rew), 1.0 - done.float()))
        cur_t.copy_(nxt)
    for k in ('obs', 'act', 'rew', 'next_obs', 'mask'):
        assert torch.equal(getattr(b_k, k), getattr(b_t, k)), k
    assert torch.equal(cur_k, cur_t) and int(b_k.pos_t) == int(b_t.pos_t) == (4 * n) % cap and int(b_k.size_i32) == cap and float(b_k.size_t) == cap
    assert int(cnt) == 4
    # ---- (3) the collector inside SAC.train_step: warm-up and policy phases, graph replays, truncations
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    env_id, tc = load_task('quadrotor_3D_track')
    env = HipVecEnv(env_id, 256, seed=5, return_numpy=False, **dict(tc, episode_len_sec=0.2))
    sac = SAC(env, SACConfig(hidden_dim=128, activation='relu', warm_up_steps=4 * 256, train_interval=10 ** 9, max_buffer_size=64 * 256), seed=5)
    assert sac._fused_collect
    prev = sac.obs.clone()
    for t in range(12):
        sac.train_step()
        torch.cuda.synchronize()
        rows = slice(t * 256, (t + 1) * 256)
        assert torch.equal(sac.buffer.obs[rows], prev) and torch.equal(sac.obs, env.out.obs)
        assert torch.equal(sac.buffer.act[rows], sac._act) and torch.equal(sac.buffer.rew[rows, 0], env.out.reward)
        tr = (env.out.flags & 1).bool() & env.out.done.bool()
        assert torch.equal(sac.buffer.next_obs[rows], torch.where(tr[:, None], env.out.terminal_obs, env.out.obs))
        assert torch.equal(sac.buffer.mask[rows, 0], torch.where(tr, torch.ones_like(env.out.reward), 1.0 - env.out.done.float()))
        lo3, hi3 = sac.low, sac.high
        assert bool((sac._act >= lo3 - 1e-6).all()) and bool((sac._act <= hi3 + 1e-6).all())
        prev = sac.obs.clone()
    assert int(sac._collect_counter) == 12 and sac.buffer.size == 12 * 256 == int(sac.buffer.size_i32)
    env.close()

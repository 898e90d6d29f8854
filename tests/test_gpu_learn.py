"""The fused MFMA learner kernels (csrc/scg_learn.hip, include/scg_learn.h) against plain PyTorch float32 autograd of the
reference's loss definitions (controllers/ppo/ppo_utils.py:82-146 as restated in safe_control_gym_amd/ppo.py, which the CPU
tests pin to fixtures produced by the reference's own PPOAgent)."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

SHAPES = [(12, 128, 2, 'tanh'), (4, 64, 1, 'leaky_relu'), (24, 128, 4, 'relu'), (12, 32, 2, 'tanh'),
          # shapes no shipped task has: three feature tiles (96: the W2 fill did not compile before round 3's fill_chunk), input
          # widths that are not multiples of 4 / 8, the widest input, a one-float observation
          (7, 96, 1, 'relu'), (17, 96, 2, 'tanh'), (31, 64, 4, 'leaky_relu'), (1, 32, 1, 'tanh'), (9, 64, 4, 'tanh')]


def _agent(obs_dim, hidden, act_dim, act, **extra):
    from safe_control_gym_amd.ppo import PPOAgent, PPOConfig
    torch.manual_seed(3)
    cfg = PPOConfig(hidden_dim=hidden, activation=act, mini_batch_size=2048, opt_epochs=2, target_kl=0.02,
                    actor_lr=1e-3, critic_lr=2e-3, entropy_coef=0.01, extra=extra)
    ag = PPOAgent(obs_dim, act_dim, cfg, 'cuda:0')
    with torch.no_grad():
        ag.ac.actor.logstd.copy_(torch.linspace(-0.7, -0.3, act_dim))
    return ag


def _data(obs_dim, act_dim, M, ag, seed=0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    obs = torch.randn(M, obs_dim, device='cuda', generator=g) * 1.5
    with torch.no_grad():
        mean, logstd = ag.ac.actor(obs)
        act = mean + torch.exp(logstd) * torch.randn(M, act_dim, device='cuda', generator=g)
        from safe_control_gym_amd.ppo import normal_log_prob
        logp = normal_log_prob(mean, logstd, act) + 0.15 * torch.randn(M, device='cuda', generator=g)     # off-policy on purpose
        v = ag.ac.critic(obs).squeeze(-1)
    adv = torch.randn(M, device='cuda', generator=g)
    ret = v + torch.randn(M, device='cuda', generator=g)
    v_old = v + 0.3 * torch.randn(M, device='cuda', generator=g)
    return {'obs': obs, 'act': act, 'logp': logp, 'adv': adv, 'ret': ret, 'v': v_old}


@pytest.mark.parametrize('shape', SHAPES)
def test_mfma_forward_equals_torch(shape):
    from safe_control_gym_amd import _learn
    obs_dim, hidden, act_dim, act = shape
    ag = _agent(obs_dim, hidden, act_dim, act)
    D = _learn.lib(obs_dim, hidden, act_dim, act)
    a_lay, c_lay, _, n = ag._layouts()
    M = 1000                                    # not a multiple of 32: tail tile
    x = torch.randn(M, obs_dim, device='cuda') * 2
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for lay, nout, net in ((a_lay, act_dim, ag.ac.actor.pi_net), (c_lay, 1, ag.ac.critic.v_net)):
        out = torch.full((M, nout), float('nan'), device='cuda')
        _learn.check(D, D.scg_mlp_forward(ag._flat['p'].data_ptr(), C.byref(lay), nout, x.data_ptr(), M, out.data_ptr(), None, st))
        ref = net(x)
        torch.testing.assert_close(out, ref, rtol=2e-5, atol=2e-5)
        # sparse evaluation: flagged rows get their value, every other row is zero (tiles without a flagged row are skipped)
        mask = torch.zeros(M, dtype=torch.uint8, device='cuda')
        mask[[5, 40, 700, 999]] = 1
        out2 = torch.full((M, nout), float('nan'), device='cuda')
        _learn.check(D, D.scg_mlp_forward(ag._flat['p'].data_ptr(), C.byref(lay), nout, x.data_ptr(), M, out2.data_ptr(),
                                          mask.data_ptr(), st))
        torch.testing.assert_close(out2[mask.bool()], ref[mask.bool()], rtol=2e-5, atol=2e-5)
        assert torch.isfinite(out2).all() and (out2[~mask.bool()] == 0).all()       # every row that is not flagged returns 0


@pytest.mark.parametrize('shape', [(12, 128, 2, 'tanh'), (17, 96, 2, 'tanh'), (4, 64, 1, 'relu')])
def test_one_tile_form_equals_the_accumulating_form(shape):
    """scg_ppo_grad at one tile per wave: the one-tile form (ppo_grad_kernel<true>: every wave publishes its transposed h1 / dz2 tiles, wave w
    forms tile row w of dW2 over all the workgroup's samples inside the MFMA accumulators) against the form that accumulates a wave's own
    products and sums the waves afterwards (forced through scg_learn_force_accumulating_form): every gradient but dW2 and the statistics bit
    for bit, dW2 to float32 summation-order noise; full and partial last workgroups, waves without a tile."""
    from safe_control_gym_amd import _learn
    obs_dim, hidden, act_dim, act = shape
    ag = _agent(obs_dim, hidden, act_dim, act)
    D = _learn.lib(obs_dim, hidden, act_dim, act)
    D.scg_learn_force_accumulating_form.argtypes = [C.c_int]
    data = _data(obs_dim, act_dim, 40000, ag)
    try:
        for mb in (16256, 2048, 96):
            F = ag._build_fused(data, mb)
            assert mb // 32 <= F['args'].n_workgroups * 4                  # one tile per wave at most
            F['idx'].copy_(torch.randperm(40000, device='cuda')[:mb].to(torch.int32))
            outs = []
            for on in (0, 1):
                D.scg_learn_force_accumulating_form(on)
                ag._flat['g'].zero_()
                ag._fused_grad(F)
                torch.cuda.synchronize()
                outs.append((ag._flat['g'].clone(), F['stats'].clone()))
            (g1, s1), (g0, s0) = outs[1], outs[0]
            assert torch.equal(s0, s1), mb
            scale = float(g0.abs().max())
            assert float((g0 - g1).abs().max()) <= 2e-6 * scale + 1e-12, (mb, float((g0 - g1).abs().max()), scale)
            a_lay, c_lay, _, _ = ag._layouts()
            w2 = torch.zeros_like(g0, dtype=torch.bool)
            for lay in (a_lay, c_lay):
                w2[lay.W2:lay.W2 + hidden * hidden] = True
            assert torch.equal(g0[~w2], g1[~w2]), mb                      # everything but the two dW2 blocks: the same code
            assert float(g0.abs().sum()) > 0
    finally:
        D.scg_learn_force_accumulating_form(0)


def test_returns_post_processing_launches_equal_the_torch_ops():
    """scg_ppo_returns_prepare / _moments / _normalise (the collector's work between rollout and update, ppo.py:276-300) against the
    PyTorch expressions they replace: flags and copies exactly, sums to float32 accumulation-order noise, the episode accumulators added to
    the running totals and zeroed."""
    from safe_control_gym_amd import _learn
    D = _learn.lib(12, 128, 2, 'tanh')
    T, N = 7, 1000
    g = torch.Generator(device='cuda').manual_seed(3)
    done = (torch.rand(T, N, device='cuda', generator=g) < 0.2).to(torch.uint8)
    flags = torch.randint(0, 8, (T, N), device='cuda', generator=g, dtype=torch.uint8)
    rew, v_all = torch.randn(T, N, device='cuda', generator=g), torch.randn(T + 1, N, device='cuda', generator=g)
    trunc, mask = torch.empty(T, N, dtype=torch.uint8, device='cuda'), torch.empty(T, N, device='cuda')
    rew_c, v = torch.empty(T, N, device='cuda'), torch.empty(T, N, device='cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())                              # noqa: E731
    _learn.check(D, D.scg_ppo_returns_prepare(p(done), p(flags), p(rew), p(v_all), T, N, p(trunc), p(mask), p(rew_c), p(v), st))
    assert torch.equal(trunc, (flags & 1) & done) and torch.equal(mask, 1.0 - done.float()) and torch.equal(rew_c, rew) and torch.equal(v, v_all[:T])
    adv = torch.randn(T, N, device='cuda', generator=g) * 3 + 0.5
    acc = torch.rand(N, 8, device='cuda', generator=g)
    acc0, totals = acc.clone(), torch.tensor([1.0, 2.0, 3.0, 4.0], device='cuda')
    scratch = torch.zeros(int(D.scg_ppo_returns_scratch_bytes()) // 4, device='cuda')
    mom, out = torch.zeros(3, device='cuda'), torch.empty(T, N, device='cuda')
    _learn.check(D, D.scg_ppo_returns_moments(p(adv), T, N, p(acc), p(scratch), p(mom), p(totals), st))
    _learn.check(D, D.scg_ppo_returns_normalise(p(adv), p(mom), T, N, p(out), st))
    torch.cuda.synchronize()
    torch.testing.assert_close(mom, torch.stack([adv.sum(), (adv * adv).sum(), torch.tensor(float(T * N), device='cuda')]), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(totals, torch.tensor([1.0, 2.0, 3.0, 4.0], device='cuda') + acc0.sum(0)[:4], rtol=1e-5, atol=1e-4)
    assert (acc == 0).all()
    torch.testing.assert_close(out, (adv - adv.mean()) / (adv.std(unbiased=False) + 1e-6), rtol=1e-4, atol=1e-5)
    # without episode accumulators: the totals are left alone; in place
    _learn.check(D, D.scg_ppo_returns_moments(p(adv), T, N, None, p(scratch), p(mom), p(totals), st))
    _learn.check(D, D.scg_ppo_returns_normalise(p(adv), p(mom), T, N, p(adv), st))
    torch.cuda.synchronize()
    torch.testing.assert_close(adv, out, rtol=0, atol=0)


def test_fused_minibatch_gradients_equal_autograd_with_several_tiles_per_wave():
    """The accumulating form of the gradient kernel (more than one 32-row tile per wave: minibatches above 127 x 128 rows) against autograd."""
    test_fused_minibatch_gradients_equal_autograd((12, 128, 2, 'tanh'), True, M=40000, mb=32768)


@pytest.mark.parametrize('clipped_value', [False, True])
@pytest.mark.parametrize('shape', SHAPES)
def test_fused_minibatch_gradients_equal_autograd(shape, clipped_value, M=8192, mb=4096):
    from safe_control_gym_amd.ppo import policy_loss_terms, value_loss_term
    obs_dim, hidden, act_dim, act = shape
    ag = _agent(obs_dim, hidden, act_dim, act)
    ag.cfg.use_clipped_value = clipped_value
    data = _data(obs_dim, act_dim, M, ag)
    F = ag._build_fused(data, mb)
    idx = torch.randperm(M, device='cuda')[:mb]
    F['idx'].copy_(idx.to(torch.int32))
    ag._flat['g'].zero_()
    ag._fused_grad(F)
    torch.cuda.synchronize()
    got = ag._flat['g'].clone()
    stats = F['stats'].tolist()
    # reference: autograd on the same minibatch
    batch = {k: v[idx] for k, v in data.items()}
    ag._flat['g'].zero_()
    pl, el, kl = policy_loss_terms(ag.ac, batch, ag.cfg.clip_param)
    vl = value_loss_term(ag.ac, batch, ag.cfg.clip_param, clipped_value)
    (pl + ag.cfg.entropy_coef * el).backward()
    vl.backward()
    ref = ag._flat['g'].clone()
    n, n_a = ag._flat['n'], ag._flat['n_a']
    for name, lo, hi in (('actor', 0, n_a), ('critic', n_a, n)):
        scale = ref[lo:hi].abs().max().item()
        err = (got[lo:hi] - ref[lo:hi]).abs().max().item()
        assert err <= 2e-4 * scale + 1e-8, (name, err, scale)
    assert abs(got[n].item() - kl.item()) < 1e-5 + 1e-4 * abs(kl.item())
    np.testing.assert_allclose(stats, [pl.item(), vl.item(), el.item(), kl.item()], rtol=2e-4, atol=1e-6)
    # the clip really bites on this data (both branches of the surrogate are exercised)
    with torch.no_grad():
        from safe_control_gym_amd.ppo import normal_log_prob
        mean, logstd = ag.ac.actor(batch['obs'])
        ratio = torch.exp(normal_log_prob(mean, logstd, batch['act']) - batch['logp'])
        frac = ((ratio < 0.8) | (ratio > 1.2)).float().mean().item()
    assert 0.05 < frac < 0.95


def test_gated_adam_kernel_equals_the_graphed_torch_adam():
    """scg_adam_gated == PPOAgent's torch formulation of the two Adam steps with the approx-KL gate (itself pinned to the
    eager torch.optim path by tests/test_gpu_rl.py), over several steps with the gate open and closed."""
    ag = _agent(12, 128, 2, 'tanh')
    fl = ag._flat
    n, n_a = fl['n'], fl['n_a']
    F = ag._build_fused(_data(12, 2, 256, ag), 256)
    p0 = fl['p'].clone()
    ref = {'p': p0.clone(), 'm': torch.zeros(n, device='cuda'), 'v': torch.zeros(n, device='cuda'), 'steps': [0.0, 0.0]}
    g = torch.Generator(device='cuda').manual_seed(1)
    for it, kl in enumerate([0.001, 0.5, 0.02, 0.029, 0.031]):
        grad = torch.randn(n + 1, device='cuda', generator=g) * 0.1
        grad[n] = kl
        fl['g'].copy_(grad)
        ag._fused_adam(F)
        gate = kl <= 1.5 * ag.cfg.target_kl
        for lo, hi, lr, k, take in ((0, n_a, ag.cfg.actor_lr, 0, gate), (n_a, n, ag.cfg.critic_lr, 1, True)):
            if not take:
                continue
            ref['steps'][k] += 1
            t = ref['steps'][k]
            gg = grad[lo:hi]
            ref['m'][lo:hi] = 0.9 * ref['m'][lo:hi] + 0.1 * gg
            ref['v'][lo:hi] = 0.999 * ref['v'][lo:hi] + 0.001 * gg * gg
            ref['p'][lo:hi] -= lr / (1 - 0.9 ** t) * ref['m'][lo:hi] / (ref['v'][lo:hi].sqrt() / (1 - 0.999 ** t) ** 0.5 + 1e-8)
    torch.cuda.synchronize()
    torch.testing.assert_close(fl['p'], ref['p'], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(fl["m"], ref["m"], rtol=1e-4, atol=1e-8)
    assert fl['steps'].tolist() == ref['steps']
    assert F['stats_acc'][4].item() == ref['steps'][0]


def test_scaled_adam_reads_the_summed_gradient_times_one_over_world():
    """scg_adam_gated_scaled (the data-parallel step: SUM all-reduce, then Adam reading g x 1 / world — gradients AND the approx-KL slot that
    gates the actor) == scg_adam_gated on the mean gradient: bit for bit with a power-of-two world (the scaling is exact), over steps with
    the gate open and closed, where the SUMMED approx-KL alone would close it."""
    a, b = _agent(12, 128, 2, 'tanh'), _agent(12, 128, 2, 'tanh')
    Fa, Fb = (x._build_fused(_data(12, 2, 256, x), 256) for x in (a, b))
    b._flat['p'].copy_(a._flat['p'])
    n = a._flat['n']
    g = torch.Generator(device='cuda').manual_seed(2)
    world = 4
    for kl in (0.001, 0.02, 0.5, 0.025):                     # gate: kl <= 1.5 target_kl = 0.03; 0.02 x 4 and 0.025 x 4 exceed it: only the MEAN may gate
        grad = torch.randn(n + 1, device='cuda', generator=g) * 0.1
        grad[n] = kl
        a._flat['g'].copy_(grad)
        a._fused_adam(Fa)
        b._flat['g'].copy_(grad * world)                     # what the SUM all-reduce of `world` identical ranks leaves
        b._fused_adam(Fb, 1.0 / world)
    torch.cuda.synchronize()
    for k in ('p', 'm', 'v', 'steps'):
        assert torch.equal(a._flat[k], b._flat[k]), k
    assert a._flat['steps'].tolist() == [3.0, 4.0]           # the actor skipped the kl = 0.5 step only


@pytest.mark.parametrize('cap', [None, 3], ids=['full_epochs', 'partial_epochs'])
def test_fused_update_matches_the_torch_update_statistically(cap):
    """A whole PPOAgent.update through the fused kernels vs the graphed PyTorch path from identical parameters, data and
    minibatch permutations: same number of gated actor steps, parameters equal to float32 accumulation-order noise.
    partial_epochs: extra['minibatches_per_epoch'] = 3 of the 8 minibatches of each shuffled epoch, on both paths."""
    data = None
    res, params = {}, {}
    for mode, extra in (('fused', {}), ('torch', {'fused_update': False})):
        if cap:
            extra = dict(extra, minibatches_per_epoch=cap)
        ag = _agent(12, 128, 2, 'tanh', **extra)
        assert ag.use_fused == (mode == 'fused')
        if data is None:
            data = _data(12, 2, 16384, ag)
        gen = torch.Generator(device='cuda').manual_seed(11)
        res[mode] = ag.update({k: v.clone() for k, v in data.items()}, generator=gen)
        torch.cuda.synchronize()
        params[mode] = ag._flat['p'].clone()
    assert res['fused']['actor_steps'] == res['torch']['actor_steps'] and res['fused']['minibatches'] == res['torch']['minibatches']
    assert res['fused']['minibatches'] == 2 * (cap or 8)
    for k in ('policy_loss', 'value_loss', 'entropy_loss', 'approx_kl'):
        assert abs(res['fused'][k] - res['torch'][k]) < 1e-4 + 1e-3 * abs(res['torch'][k]), (k, res)
    torch.testing.assert_close(params['fused'], params['torch'], rtol=0, atol=2e-4)


def test_random_permutation_kernel_is_a_keyed_bijection():
    """scg_random_permutation (the update loop's SubsetRandomSampler): first `count` images of a permutation of range(n),
    different per key, no fixed structure between consecutive indices."""
    import ctypes as C
    from safe_control_gym_amd import _learn
    D = _learn.lib(12, 128, 2, 'tanh')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n in (1, 2, 1000, 65536, 524288, 2097152 + 77):
        out = torch.full((n,), -1, dtype=torch.int32, device='cuda')
        _learn.check(D, D.scg_random_permutation(out.data_ptr(), n, n, 0x1234567890ABCDEF, st))
        assert torch.equal(torch.sort(out).values, torch.arange(n, dtype=torch.int32, device='cuda')), n
        if n >= 1000:
            out2 = torch.empty_like(out)
            _learn.check(D, D.scg_random_permutation(out2.data_ptr(), n, n, 0x1234567890ABCDF0, st))
            assert float((out == out2).float().mean()) < 0.01                       # another key, another permutation
            d = (out[1:].double() - out[:-1].double())
            assert abs(float(d.mean())) < 0.02 * n and float(d.abs().mean()) > 0.25 * n   # neighbours land far apart (uniform: n / 3)
            assert float((out.double() - torch.arange(n, device='cuda').double()).abs().mean()) > 0.25 * n
    part = torch.full((4096,), -1, dtype=torch.int32, device='cuda')
    _learn.check(D, D.scg_random_permutation(part.data_ptr(), 100000, 4096, 7, st))          # count < n: distinct, in range
    assert part.min() >= 0 and part.max() < 100000 and torch.unique(part).numel() == 4096
    assert D.scg_random_permutation(part.data_ptr(), 10, 11, 7, st) != 0


def test_gradient_kernel_is_bitwise_reproducible():
    """No atomics on the path of the 12-128-128-{2,1} pair (per-wave private accumulation, wave-ordered and fixed-order
    sums): the same minibatch gives the same bits, launch after launch."""
    ag = _agent(12, 128, 2, 'tanh')
    M, mb = 131072, 65024
    data = _data(12, 2, M, ag)
    F = ag._build_fused(data, mb)
    F['idx'].copy_(torch.randperm(M, device='cuda')[:mb].to(torch.int32))
    outs = []
    for _ in range(4):
        ag._flat['g'].zero_()
        ag._fused_grad(F)
        torch.cuda.synchronize()
        outs.append((ag._flat['g'].clone(), F['stats'].clone()))
    for g, st in outs[1:]:
        assert torch.equal(g, outs[0][0]) and torch.equal(st, outs[0][1])
    assert float(outs[0][0].abs().sum()) > 0


# ---- directly against the REFERENCE's PPOAgent.update (tests/golden/learner.npz, learner_variants.npz)
def _fixture_cases():
    from tests.golden.learner_cases import PPO_CASES
    cases = [('learner.npz', 'ppo', dict(obs=12, act=2, kw=dict(hidden_dim=32, activation='tanh', use_clipped_value=False, clip_param=0.2,
                                                                 target_kl=0.02, entropy_coef=0.01, actor_lr=3e-3, critic_lr=1e-3, opt_epochs=3,
                                                                 mini_batch_size=64)))]
    for name, c in sorted(PPO_CASES.items()):
        if c['kw']['mini_batch_size'] % 32 == 0:            # (40-row minibatches take the PyTorch update: PPOAgent.update routes them)
            cases.append(('learner_variants.npz', f'ppo/{name}', c))
    return cases


@pytest.mark.parametrize('file,prefix,c', _fixture_cases(), ids=[p.split('/')[-1] for _, p, _ in _fixture_cases()])
def test_fused_update_reproduces_the_reference_ppo_agent(file, prefix, c):
    """scg_ppo_grad + scg_adam_gated, minibatch by minibatch on the index rows the reference's sampler drew, from the reference's
    initial weights: the reference's final weights, Adam step counts (the approx-KL gate) and averaged loss statistics."""
    import os
    from safe_control_gym_amd.ppo import PPOAgent, PPOConfig
    G = np.load(os.path.join(os.path.dirname(__file__), 'golden', file))
    sd = lambda pre: {k[len(pre) + 1:]: torch.as_tensor(G[k]) for k in G.files if k.startswith(pre + '/')}      # noqa: E731
    cfg = PPOConfig(**c['kw'])
    ag = PPOAgent(c['obs'], c['act'], cfg, 'cuda:0')
    assert ag.use_fused
    ag.ac.load_state_dict(sd(prefix + '/init'))
    data = {k: torch.as_tensor(G[f'{prefix}/data/{k}'], dtype=torch.float32, device='cuda') for k in ('obs', 'act', 'logp', 'adv', 'ret', 'v')}
    data = {k: (v.reshape(-1) if k in ('logp', 'adv', 'ret', 'v') else v).contiguous() for k, v in data.items()}
    M = data['obs'].shape[0]
    mb = min(cfg.mini_batch_size, M)
    n_mb = M // mb
    F = ag._build_fused(data, mb)
    F['stats_acc'].zero_()
    for perm in G[prefix + '/perms']:
        for j in range(n_mb):
            F['idx'].copy_(torch.as_tensor(perm[j * mb:(j + 1) * mb].astype(np.int32), device='cuda'))
            ag._fused_grad(F)
            ag._fused_adam(F)
    torch.cuda.synchronize()
    n_steps = cfg.opt_epochs * n_mb
    st = (F['stats_acc'] / n_steps).tolist()
    assert int(round(st[4] * n_steps)) == int(G[prefix + '/actor_adam_steps']) and n_steps == int(G[prefix + '/critic_adam_steps'])
    np.testing.assert_allclose(st[:4], G[prefix + '/results'], rtol=2e-4, atol=2e-5)
    final = sd(prefix + '/final')
    for k, v in ag.ac.state_dict().items():
        torch.testing.assert_close(v.cpu(), final[k], rtol=2e-4, atol=1e-5, msg=lambda m, k=k: f'{prefix} {k}: {m}')


@pytest.mark.parametrize('mb,epochs', [(2048, 2), (4096, 3)], ids=['16_steps', '9_steps_odd'])
def test_fused_optimiser_step_equals_gradient_reduction_then_gated_adam(mb, epochs):
    """scg_ppo_step (gradient kernel + ONE kernel that sums the partials and steps every parameter, double-buffered step counts)
    against scg_ppo_grad + scg_adam_gated (the path data-parallel runs keep): same minibatch permutations -> the same parameters, moments
    and step counts — the per-element sums and the Adam arithmetic are the same code, only the approx-KL word is summed in another
    (fixed) order — and the same loss statistics.  The odd step count leaves the counts in the scratch bank: `_flat['steps']` must hold
    them after the update."""
    from safe_control_gym_amd.ppo import PPOAgent, PPOConfig
    out = {}
    data = None
    for mode, extra in (('step', {}), ('split', {'fused_step': False})):
        torch.manual_seed(3)
        ag = PPOAgent(12, 2, PPOConfig(hidden_dim=128, activation='tanh', mini_batch_size=mb, opt_epochs=epochs, target_kl=0.02, actor_lr=1e-3,
                                       critic_lr=2e-3, entropy_coef=0.01, extra=extra), 'cuda:0')
        assert ag.use_fused and ag._fused_step_ok == (mode == 'step')
        if data is None:
            data = _data(12, 2, 12288 if mb == 4096 else 16384, ag)
        gen = torch.Generator(device='cuda').manual_seed(5)
        res = ag.update({k: v.clone() for k, v in data.items()}, generator=gen)
        torch.cuda.synchronize()
        out[mode] = (res, ag._flat['p'].clone(), ag._flat['m'].clone(), ag._flat['v'].clone(), ag._flat['steps'].tolist(), ag._flat.get('bank', 0))
    (ra, pa, ma, va, sa, bank), (rb, pb, mb_, vb, sb, _) = out['step'], out['split']
    n_steps = epochs * (data['obs'].shape[0] // mb)
    assert ra['minibatches'] == rb['minibatches'] == n_steps and ra['actor_steps'] == rb['actor_steps']
    assert sa == sb == [float(ra['actor_steps']), float(n_steps)] and bank == 0
    assert 0 < ra['actor_steps'] <= n_steps
    torch.testing.assert_close(pa, pb, rtol=0, atol=0)
    torch.testing.assert_close(ma, mb_, rtol=0, atol=0)
    torch.testing.assert_close(va, vb, rtol=0, atol=0)
    for k in ('policy_loss', 'value_loss', 'entropy_loss', 'approx_kl'):
        assert ra[k] == pytest.approx(rb[k], rel=1e-5, abs=1e-7), k

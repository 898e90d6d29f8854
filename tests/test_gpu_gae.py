"""scg_gae (both kernels) against the oracle restatement of ppo_utils.py:374-400 and the reference's
own known answers (tests/golden/gae.npz)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def _run(rews, vals, masks, term, last, gamma, lam, use_gae, dtype):
    from safe_control_gym_amd.rollout import gae_returns
    dev = torch.device('cuda:0')
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=dev)   # noqa: E731
    r = t(rews[..., 0])
    ret, adv = gae_returns(r, t(vals[..., 0]), t(masks[..., 0]), t(term[..., 0]), t(last[..., 0]), gamma, lam, use_gae)
    return ret.cpu().numpy()[..., None], adv.cpu().numpy()[..., None], r.cpu().numpy()[..., None]


@pytest.mark.parametrize('k', [0, 1, 2])
def test_gae_reference_known_answers(k):
    g = np.load(os.path.join(GOLDEN, 'gae.npz'))
    ret, adv, rew_after = _run(g[f'rews{k}'], g[f'vals{k}'], g[f'masks{k}'], g[f'term{k}'], g[f'last{k}'],
                               0.99, 0.95, bool(g[f'use_gae{k}']), torch.float32)
    np.testing.assert_allclose(ret, g[f'rets{k}'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(adv, g[f'advs{k}'], rtol=1e-5, atol=1e-5)
    # the reference adds gamma * terminal_v into the reward buffer in place (ppo_utils.py:389)
    np.testing.assert_allclose(rew_after, g[f'rews{k}'] + 0.99 * g[f'term{k}'], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('T,N', [(33, 4096), (16, 65536), (1000, 4), (257, 7), (64, 1), (5, 3)])
@pytest.mark.parametrize('use_gae', [True, False])
@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_gae_matches_oracle(T, N, use_gae, dtype):
    from oracle.vec import compute_returns_and_advantages
    rng = np.random.default_rng(T * 131 + N)
    rews = rng.standard_normal((T, N, 1))
    vals = rng.standard_normal((T, N, 1))
    masks = (rng.uniform(size=(T, N, 1)) > 0.05).astype(np.float64)
    term = rng.standard_normal((T, N, 1)) * (masks == 0) * (rng.uniform(size=(T, N, 1)) > 0.5)
    last = rng.standard_normal((N, 1))
    ret_o, adv_o = compute_returns_and_advantages(rews, vals, masks, term, last, 0.99, use_gae, 0.95)
    ret, adv, _ = _run(rews, vals, masks, term, last, 0.99, 0.95, use_gae, getattr(torch, dtype))
    tol = dict(rtol=1e-10, atol=1e-10) if dtype == 'float64' else dict(rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(ret, ret_o, **tol)
    np.testing.assert_allclose(adv, adv_o, **tol)


# shapes around every dispatch boundary of launch_gae (scg_kernels.hip: gae_seg_kernel for N >= 1024 and 9 <= T <= 128,
# gae_env_kernel for N >= 1024 or T < 64, gae_wave_kernel otherwise), ragged last waves / last time chunks included
BOUNDARY_SHAPES = [(8, 1024), (9, 1024), (9, 1030), (15, 2049), (16, 1025), (17, 1087), (127, 1100), (128, 1024), (129, 1024), (63, 1023),
                   (64, 1023), (65, 100), (64, 64), (63, 64), (1, 1), (1, 70000), (2, 1), (300, 63), (513, 2), (40, 5000)]


@pytest.mark.parametrize('T,N', BOUNDARY_SHAPES)
def test_gae_dispatch_boundaries_random_parameters(T, N):
    """Random gamma / lambda (0 and 1 included), mask densities from all-terminal to none, with and without a terminal-value
    buffer, both advantage estimators, float64 (1e-10) and float32."""
    from oracle.vec import compute_returns_and_advantages
    from safe_control_gym_amd.rollout import gae_returns
    rng = np.random.default_rng(T * 7919 + N)
    for trial in range(3):
        gamma = float(rng.choice([0.0, 0.9, 0.99, 1.0]))
        lam = float(rng.choice([0.0, 0.5, 0.95, 1.0]))
        use_gae = bool(rng.integers(0, 2))
        p_term = float(rng.choice([0.0, 0.02, 0.3, 1.0]))
        rews = rng.standard_normal((T, N, 1))
        vals = rng.standard_normal((T, N, 1))
        masks = (rng.uniform(size=(T, N, 1)) >= p_term).astype(np.float64)
        with_term = bool(rng.integers(0, 2))
        term = rng.standard_normal((T, N, 1)) * (masks == 0) * (rng.uniform(size=(T, N, 1)) > 0.5) if with_term else np.zeros((T, N, 1))
        last = rng.standard_normal((N, 1))
        ret_o, adv_o = compute_returns_and_advantages(rews.copy(), vals, masks, term, last, gamma, use_gae, lam)
        for dtype, tol in ((torch.float64, dict(rtol=1e-10, atol=1e-10)), (torch.float32, dict(rtol=3e-4, atol=3e-4))):
            dev = torch.device('cuda:0')
            t = lambda a: torch.as_tensor(np.ascontiguousarray(a[..., 0]), dtype=dtype, device=dev)   # noqa: E731
            r = t(rews)
            ret, adv = gae_returns(r, t(vals), t(masks), t(term) if with_term else None, t(last), gamma, lam, use_gae)
            msg = f'T={T} N={N} gamma={gamma} lam={lam} use_gae={use_gae} p_term={p_term} term={with_term} {dtype}'
            scale = max(1.0, float(np.abs(ret_o).max()))           # (gamma = 1: returns grow with T)
            np.testing.assert_allclose(ret.cpu().numpy() / scale, ret_o[..., 0] / scale, err_msg=msg, **tol)
            np.testing.assert_allclose(adv.cpu().numpy() / scale, adv_o[..., 0] / scale, err_msg=msg, **tol)
            np.testing.assert_allclose(r.cpu().numpy(), (rews + gamma * term)[..., 0], rtol=1e-6, atol=1e-6, err_msg=msg)

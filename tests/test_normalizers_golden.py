"""Observation / reward normalisers against known answers produced by the REFERENCE's own classes
(tests/golden/make_normalizers.py: math_and_models/normalization.py) — the device-resident torch implementation the collectors
use (safe_control_gym_amd/normalization.py) and the oracle's NumPy restatement, step by step over 40 vector steps: normalised
outputs, running mean / variance / count, the running discounted returns after upstream's index-array reset
(`ret[dones.astype(np.long)] = 0`: rows 0 / 1, not the done envs), frozen statistics in read-only mode, the state dict."""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')

from tests.devices import DEVICES  # noqa: E402
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'normalizers.npz'))
TOL = dict(rtol=1e-12, atol=1e-12)


def _run(mod, as_in, as_np, device='cpu'):
    D = G['obs'].shape[2]
    if mod.__name__.startswith('oracle'):
        o, r = mod.MeanStdNormalizer(shape=(D,), clip=2.5), mod.RewardStdNormalizer(gamma=0.97, clip=3.0)
    else:
        o, r = mod.MeanStdNormalizer((D,), device, clip=2.5), mod.RewardStdNormalizer(0.97, device, clip=3.0)
    for t in range(G['obs'].shape[0]):
        if t == 30:
            o.set_read_only(); r.set_read_only()
        np.testing.assert_allclose(as_np(o(as_in(G['obs'][t]))), G['obs_norm'][t], err_msg=f'obs t={t}', **TOL)
        np.testing.assert_allclose(as_np(r(as_in(G['rew'][t]), as_in(G['done'][t]))), G['rew_norm'][t], err_msg=f'rew t={t}', **TOL)
        np.testing.assert_allclose(as_np(r.ret), G['ret'][t], err_msg=f'ret t={t}', **TOL)
        np.testing.assert_allclose(as_np(o.rms.mean), G['obs_mean'][t], **TOL)
        np.testing.assert_allclose(as_np(o.rms.var), G['obs_var'][t], **TOL)
        np.testing.assert_allclose(float(as_np(r.rms.var)), float(G['rew_var'][t]), **TOL)
        np.testing.assert_allclose(float(as_np(o.rms.count)), float(G['obs_count'][t]), rtol=1e-12)
    return o


@pytest.mark.parametrize('device', DEVICES)
def test_device_normalizers_reproduce_the_reference(device):
    from safe_control_gym_amd import normalization as dev
    o = _run(dev, lambda a: torch.as_tensor(a, device=device), lambda t: t.cpu().numpy() if torch.is_tensor(t) else np.asarray(t), device)
    sd = o.state_dict()
    np.testing.assert_allclose(sd['mean'], G['sd_mean'], **TOL)
    np.testing.assert_allclose(sd['var'], G['sd_var'], **TOL)
    assert (G['ret'][7][2:] != 0).all() and G['ret'][7][0] != 0            # every env done on step 7: upstream clears row 1 only


def test_oracle_normalizers_reproduce_the_reference():
    from oracle import normalization as ref
    _run(ref, np.asarray, np.asarray)

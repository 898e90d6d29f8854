"""SURVEY §8f-3 pinned to the reference: the safety layer and the Safe-Explorer PPO update against known answers produced by
the reference's OWN classes (tests/golden/make_adversarial.py -> adversarial.npz: safe_explorer_utils.SafetyLayer /
ConstraintBuffer, safe_ppo_utils.SafePPOAgent on fixed data), on the CPU and (gpu mark) on the MI355X with device tensors; the RARL / RAP
collectors (which need the device GAE kernel) replay the reference's recorded transitions in tests/test_gpu_adversarial.py."""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')

from tests.devices import DEVICES  # noqa: E402

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'adversarial.npz'))


def sd(prefix):
    return {k[len(prefix) + 1:]: torch.as_tensor(G[k]) for k in G.files if k.startswith(prefix + '/')}


def _layer(device='cpu'):
    from safe_control_gym_amd.safe_explorer import SafetyLayer
    layer = SafetyLayer(5, 2, 3, hidden_dim=16, lr=3e-3, slack=[0.05, 0.0, 0.1], device=device)
    layer.constraint_models.load_state_dict(sd('safety/init'))
    return layer


@pytest.mark.parametrize('device', DEVICES)
def test_safety_layer_projection_equals_the_reference(device):
    layer = _layer(device)
    obs, act, c = (torch.as_tensor(G[f'safety/proj/{k}'], device=device) for k in ('obs', 'act', 'c'))
    with torch.no_grad():
        got = layer.get_safe_action(obs, act, c)
    np.testing.assert_allclose(got.cpu().numpy(), G['safety/proj/safe'], rtol=1e-5, atol=1e-6)
    assert np.abs(G['safety/proj/safe'] - G['safety/proj/act']).max() > 1e-3          # the projection did act on this batch


@pytest.mark.parametrize('device', DEVICES)
def test_constraint_buffer_ring_and_pretraining_updates_equal_the_reference(device):
    from safe_control_gym_amd.safe_explorer import ConstraintBuffer
    layer = _layer(device)
    rng = np.random.default_rng(17)
    rng.normal(0, 1, (48, 5)); rng.normal(0, 1, (48, 2)); rng.normal(0, 1, (48, 3))       # the generator's projection batch came first
    buf = ConstraintBuffer(100, 5, 2, 3, device)
    for _ in range(3):                                                                      # 120 > 100: the ring wraps
        n = 40
        b = {'obs': rng.normal(0, 1, (n, 5)), 'act': rng.normal(0, 1, (n, 2)), 'c': rng.normal(0, 0.3, (n, 3)), 'c_next': rng.normal(0, 0.3, (n, 3))}
        buf.push(**{k: torch.as_tensor(v, dtype=torch.float32, device=device) for k, v in b.items()})
    assert [buf.pos, buf.size] == G['safety/buffer/pos_size'].tolist()
    for k in ('obs', 'act', 'c', 'c_next'):
        np.testing.assert_array_equal(buf.data[k].cpu().numpy(), G[f'safety/buffer/{k}'])
    losses = []
    for ind in G['safety/indices']:
        ind = torch.as_tensor(ind, device=device)
        losses.append(layer.update({k: t[ind] for k, t in buf.data.items()}).tolist())
    np.testing.assert_allclose(losses, G['safety/losses'], rtol=2e-5, atol=1e-6)
    final = sd('safety/final')
    for k, v in layer.constraint_models.state_dict().items():
        torch.testing.assert_close(v.cpu(), final[k], rtol=1e-4, atol=2e-6, msg=lambda m, k=k: f'{k}: {m}')


@pytest.mark.parametrize('device', DEVICES)
def test_safe_explorer_ppo_update_equals_the_reference(device):
    """SafePPOAgent.update (safe_ppo_utils.py:16-60): the safety layer filters the actor's mean, the constraint values are a
    policy input, the layer's own parameters do not move."""
    import warnings
    from safe_control_gym_amd.ppo import PPOAgent, PPOConfig, normal_log_prob
    from safe_control_gym_amd.safe_explorer import SafetyLayer
    layer = SafetyLayer(5, 2, 3, hidden_dim=16, lr=3e-3, slack=[0.05, 0.0, 0.1], device=device)
    layer.constraint_models.load_state_dict(sd('safety/final'))                              # (the generator trained it first)
    cfg = PPOConfig(hidden_dim=16, activation='tanh', use_clipped_value=False, clip_param=0.2, target_kl=0.05, entropy_coef=0.01,
                    actor_lr=3e-3, critic_lr=1e-3, opt_epochs=2, mini_batch_size=32, extra={'cuda_graphs': False})      # (explicit permutations: the eager update, on either device)
    ag = PPOAgent(5, 2, cfg, device)
    ag.ac.load_state_dict(sd('safeppo/init'))
    ag.ac.actor.action_modifier = layer.get_safe_action
    data = {k: torch.as_tensor(G[f'safeppo/data/{k}'], device=device) for k in ('obs', 'act', 'logp', 'adv', 'ret', 'v', 'c')}
    with torch.no_grad():       # the buffer's log-probs are those of the FILTERED distribution under the initial weights
        mean, logstd = ag.ac.actor(data['obs'], data['c'])
        torch.testing.assert_close(normal_log_prob(mean, logstd, data['act']), data['logp'].reshape(-1), rtol=1e-5, atol=1e-5)
        raw_mean = ag.ac.actor.pi_net(data['obs'])
        assert (mean - raw_mean).abs().max() > 1e-3
    flat = {k: (v.reshape(-1) if k in ('logp', 'adv', 'ret', 'v') else v) for k, v in data.items()}
    before = {k: v.clone() for k, v in layer.constraint_models.state_dict().items()}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        res = ag.update(flat, perms=G['safeppo/perms'])
    np.testing.assert_allclose([res['policy_loss'], res['value_loss'], res['entropy_loss'], res['approx_kl']], G['safeppo/results'],
                               rtol=2e-5, atol=2e-6)
    final = sd('safeppo/final')
    for k, v in ag.ac.state_dict().items():
        torch.testing.assert_close(v.cpu(), final[k], rtol=1e-4, atol=2e-6, msg=lambda m, k=k: f'{k}: {m}')
    for k, v in layer.constraint_models.state_dict().items():
        assert torch.equal(v, before[k])


def test_rap_groups_are_sorted_contiguous_like_the_reference():
    """rap.py:356-357: `sorted(np.random.randint(num_adversaries, size=batch))` + np.unique(..., return_index=True) — the
    fixture's draw and its split points, and this package's RAP.collect grouping rule on the same indices."""
    idx = G['rap/adv_indices']
    assert np.all(np.diff(idx) >= 0) and set(idx.tolist()) <= {0, 1, 2}
    groups, starts = np.unique(idx, return_index=True)
    ends = np.concatenate([starts[1:], [len(idx)]])
    np.testing.assert_array_equal(G['rap/split_ids'], groups)
    for g, s, e in zip(groups, starts, ends):
        assert G[f'rap/adversary{int(g)}/buffer/act'].shape[1] == e - s                  # the split rollouts are the contiguous slices

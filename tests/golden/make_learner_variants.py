#!/usr/bin/env python3
"""More known answers of the reference's LEARNERS (see make_learner.py for the protocol): the hyper-parameter corners the first
fixture leaves out — clipped value loss, an approx-KL gate that closes after the first minibatches and one that never closes, a
single whole-batch epoch, plain discounted returns instead of GAE, relu / leaky_relu trunks, one to four actions — run through the
reference's own PPOAgent.update / SACAgent.update (with and without temperature tuning, asymmetric action bounds).

    python tests/golden/make_learner_variants.py      (build container only: needs /root/reference) -> learner_variants.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden import ref_stubs  # noqa: E402

ref_stubs.install()

import torch  # noqa: E402
from gymnasium.spaces import Box  # noqa: E402

from safe_control_gym.controllers.ppo import ppo_utils  # noqa: E402
from safe_control_gym.controllers.sac import sac_utils  # noqa: E402

from tests.golden.learner_cases import PPO_CASES, SAC_CASES  # noqa: E402


def flat_sd(sd, prefix):
    return {f'{prefix}/{k}': v.detach().cpu().numpy().copy() for k, v in sd.items()}


def ppo_case(out, name, c):
    nobs, nact, T, N, kw = c['obs'], c['act'], c['T'], c['N'], c['kw']
    obs_space, act_space = Box(-1, 1, (nobs,)), Box(-1, 1, (nact,))
    torch.manual_seed(sum(map(ord, name)))
    agent = ppo_utils.PPOAgent(obs_space, act_space, **kw)
    p = f'ppo/{name}'
    out.update(flat_sd(agent.ac.state_dict(), p + '/init'))
    rng = np.random.default_rng(len(name))
    buf = ppo_utils.PPOBuffer(obs_space, act_space, T, N)
    obs = rng.normal(0, 1, (T + 1, N, nobs)).astype(np.float32)
    for t in range(T):
        torch.manual_seed(100 + t)
        with torch.no_grad():
            act, v, logp = agent.ac.step(torch.as_tensor(obs[t]))
        rew = rng.normal(0, 1, (N,))
        mask = (rng.uniform(size=N) > 0.15).astype(np.float32)
        term_v = np.where(mask == 0, rng.normal(0, 1, N) * (rng.uniform(size=N) > 0.5), 0.0)
        buf.push({'obs': obs[t], 'act': act, 'rew': rew, 'mask': mask, 'v': v, 'logp': logp, 'terminal_v': term_v})
    last_val = agent.ac.critic(torch.as_tensor(obs[T])).detach().numpy()
    out[p + '/raw/rew'], out[p + '/raw/v'], out[p + '/raw/mask'] = buf.rew.copy(), buf.v.copy(), buf.mask.copy()
    out[p + '/raw/terminal_v'], out[p + '/raw/last_val'] = buf.terminal_v.copy(), last_val.copy()
    ret, adv = ppo_utils.compute_returns_and_advantages(buf.rew, buf.v, buf.mask, buf.terminal_v, last_val, gamma=0.97,
                                                        use_gae=c['use_gae'], gae_lambda=0.9)
    out[p + '/raw/ret'], out[p + '/raw/adv'] = ret.copy(), adv.copy()
    buf.ret = ret
    buf.adv = (adv - adv.mean()) / (adv.std() + 1e-6)
    data = buf.get()
    for k in ('obs', 'act', 'logp', 'adv', 'ret', 'v'):
        out[f'{p}/data/{k}'] = data[k].numpy()
    np.random.seed(17)
    out[p + '/perms'] = np.stack([np.random.permutation(T * N) for _ in range(kw['opt_epochs'])])
    np.random.seed(17)
    res = agent.update(buf)
    out.update(flat_sd(agent.ac.state_dict(), p + '/final'))
    out[p + '/results'] = np.array([res['policy_loss'], res['value_loss'], res['entropy_loss'], res['approx_kl']])
    st = agent.actor_opt.state_dict()['state']
    out[p + '/actor_adam_steps'] = np.array(float(st[0]['step']) if st else 0.0)
    out[p + '/critic_adam_steps'] = np.array(float(agent.critic_opt.state_dict()['state'][0]['step']))
    print(name, 'actor steps', out[p + '/actor_adam_steps'], 'critic steps', out[p + '/critic_adam_steps'], out[p + '/results'])


def sac_case(out, name, c):
    nobs, low, high, kw = c['obs'], np.array(c['low'], dtype=np.float32), np.array(c['high'], dtype=np.float32), c['kw']
    nact = len(low)
    obs_space, act_space = Box(-1, 1, (nobs,)), Box(low, high)
    torch.manual_seed(sum(map(ord, name)))
    agent = sac_utils.SACAgent(obs_space, act_space, **kw)
    agent.train()
    p = f'sac/{name}'
    out.update(flat_sd(agent.ac.state_dict(), p + '/init'))
    rng = np.random.default_rng(len(name))
    n = 300
    buf = sac_utils.SACBuffer(obs_space, act_space, max_size=n, batch_size=64)
    buf.push({'obs': rng.normal(0, 1, (n, nobs)), 'act': rng.uniform(low, high, (n, nact)), 'rew': rng.normal(0, 1, (n,)),
              'next_obs': rng.normal(0, 1, (n, nobs)), 'mask': (rng.uniform(size=n) > 0.1).astype(np.float32)})
    for k in ('obs', 'act', 'rew', 'next_obs', 'mask'):
        out[f'{p}/buffer/{k}'] = buf.__dict__[k].copy()
    np.random.seed(23)
    out[p + '/indices'] = np.stack([np.random.randint(0, len(buf), size=64) for _ in range(4)])
    np.random.seed(23)
    torch.manual_seed(29)
    res = []
    for _ in range(4):
        r = agent.update(buf.sample(64))
        res.append([r['policy_loss'], r['critic_loss'], r['entropy_loss']])
    out[p + '/results'] = np.array(res, dtype=np.float64)
    out.update(flat_sd(agent.ac.state_dict(), p + '/final'))
    out.update(flat_sd(agent.ac_targ.state_dict(), p + '/final_targ'))
    out[p + '/final_log_alpha'] = agent.log_alpha.detach().numpy()
    print(name, out[p + '/results'].tolist(), float(agent.log_alpha))


def main():
    out = {}
    for name, c in PPO_CASES.items():
        ppo_case(out, name, c)
    for name, c in SAC_CASES.items():
        sac_case(out, name, c)
    np.savez_compressed(os.path.join(HERE, 'learner_variants.npz'), **out)
    print('learner_variants.npz written,', len(out), 'arrays')


if __name__ == '__main__':
    main()

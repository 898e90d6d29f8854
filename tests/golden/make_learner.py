#!/usr/bin/env python3
"""Known answers of the reference's LEARNERS, produced by running the reference's own classes.

    python tests/golden/make_learner.py             (build container only: needs /root/reference)

controllers/ppo/ppo_utils.py — MLPActorCritic.step, PPOBuffer.push / get / sampler, compute_returns_and_advantages,
PPOAgent.update (epochs x shuffled minibatches, approx-KL gate, two Adam optimisers) — and controllers/sac/sac_utils.py —
SACAgent.update (tanh-Gaussian actor, twin Q, temperature tuning, Polyak) and SACBuffer.push / sample — are pure
torch / NumPy and import under tests/golden/ref_stubs.py.  They are run on fixed random data; initial weights, the data as
the reference's buffers hold it, the index batches its samplers drew, the final weights / optimiser statistics and the
returned loss statistics go to tests/golden/learner.npz.  tests/test_learner_golden.py makes this repo's PPOAgent /
SACAgent (eager CPU path) reproduce them; the GPU tests pin the graphed / fused paths to the eager one.
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


def flat_sd(sd, prefix):
    return {f'{prefix}/{k}': v.detach().cpu().numpy().copy() for k, v in sd.items()}      # copy: the live tensors are updated in place later


def ppo_case(out):
    obs_space, act_space = Box(-1, 1, (12,)), Box(-1, 1, (2,))
    T, N = 16, 16
    torch.manual_seed(5)
    agent = ppo_utils.PPOAgent(obs_space, act_space, hidden_dim=32, use_clipped_value=False, clip_param=0.2, target_kl=0.02,
                               entropy_coef=0.01, actor_lr=3e-3, critic_lr=1e-3, opt_epochs=3, mini_batch_size=64, activation='tanh')
    out.update(flat_sd(agent.ac.state_dict(), 'ppo/init'))
    rng = np.random.default_rng(3)
    buf = ppo_utils.PPOBuffer(obs_space, act_space, T, N)
    obs = rng.normal(0, 1, (T + 1, N, 12)).astype(np.float32)
    for t in range(T):
        torch.manual_seed(100 + t)
        with torch.no_grad():                                                   # as PPO.train_step does (ppo.py:266-268)
            act, v, logp = agent.ac.step(torch.as_tensor(obs[t]))              # MLPActorCritic.step (ppo_utils.py:224-231)
        rew = rng.normal(0, 1, (N,))
        mask = (rng.uniform(size=N) > 0.1).astype(np.float32)
        term_v = np.where(mask == 0, rng.normal(0, 1, N) * (rng.uniform(size=N) > 0.5), 0.0)
        buf.push({'obs': obs[t], 'act': act, 'rew': rew, 'mask': mask, 'v': v, 'logp': logp, 'terminal_v': term_v})
    last_val = agent.ac.critic(torch.as_tensor(obs[T])).detach().numpy()
    ret, adv = ppo_utils.compute_returns_and_advantages(buf.rew, buf.v, buf.mask, buf.terminal_v, last_val, gamma=0.99,
                                                        use_gae=True, gae_lambda=0.95)
    buf.ret = ret
    buf.adv = (adv - adv.mean()) / (adv.std() + 1e-6)                           # ppo.py:300
    data = buf.get()                                                            # flat [T*N, .] as the sampler indexes it
    for k in ('obs', 'act', 'logp', 'adv', 'ret', 'v'):
        out[f'ppo/data/{k}'] = data[k].numpy()
    np.random.seed(11)
    out['ppo/perms'] = np.stack([np.random.permutation(T * N) for _ in range(3)])
    np.random.seed(11)
    res = agent.update(buf)
    out.update(flat_sd(agent.ac.state_dict(), 'ppo/final'))
    out['ppo/results'] = np.array([res['policy_loss'], res['value_loss'], res['entropy_loss'], res['approx_kl']])
    out['ppo/actor_adam_steps'] = np.array(float(agent.actor_opt.state_dict()['state'][0]['step']))
    out['ppo/critic_adam_steps'] = np.array(float(agent.critic_opt.state_dict()['state'][0]['step']))
    print('ppo: actor steps', out['ppo/actor_adam_steps'], 'critic steps', out['ppo/critic_adam_steps'], 'results', out['ppo/results'])


def sac_case(out):
    obs_space = Box(-1, 1, (6,))
    act_space = Box(np.array([-1.0, 0.0], dtype=np.float32), np.array([1.0, 2.0], dtype=np.float32))
    torch.manual_seed(7)
    agent = sac_utils.SACAgent(obs_space, act_space, hidden_dim=32, gamma=0.98, tau=0.01, init_temperature=0.3,
                               use_entropy_tuning=True, actor_lr=1e-3, critic_lr=2e-3, entropy_lr=3e-3, activation='relu')
    agent.train()
    out.update(flat_sd(agent.ac.state_dict(), 'sac/init'))
    rng = np.random.default_rng(9)
    buf = sac_utils.SACBuffer(obs_space, act_space, max_size=200, batch_size=64)
    for _ in range(5):                                                          # 250 > max_size: the ring wraps
        n = 50
        buf.push({'obs': rng.normal(0, 1, (n, 6)), 'act': rng.uniform([-1, 0], [1, 2], (n, 2)), 'rew': rng.normal(0, 1, (n,)),
                  'next_obs': rng.normal(0, 1, (n, 6)), 'mask': (rng.uniform(size=n) > 0.1).astype(np.float32)})
    for k in ('obs', 'act', 'rew', 'next_obs', 'mask'):
        out[f'sac/buffer/{k}'] = buf.__dict__[k].copy()
    out['sac/buffer/pos_size'] = np.array([buf.pos, buf.buffer_size])
    np.random.seed(21)
    idx = np.stack([np.random.randint(0, len(buf), size=64) for _ in range(3)])
    out['sac/indices'] = idx
    np.random.seed(21)
    torch.manual_seed(13)
    res = []
    for _ in range(3):
        r = agent.update(buf.sample(64))
        res.append([r['policy_loss'], r['critic_loss'], r['entropy_loss']])
    out['sac/results'] = np.array(res)
    out.update(flat_sd(agent.ac.state_dict(), 'sac/final'))
    out.update(flat_sd(agent.ac_targ.state_dict(), 'sac/final_targ'))
    out['sac/final_log_alpha'] = agent.log_alpha.detach().numpy()
    print('sac: results', out['sac/results'].tolist(), 'log_alpha', float(agent.log_alpha))


def main():
    out = {}
    ppo_case(out)
    sac_case(out)
    np.savez_compressed(os.path.join(HERE, 'learner.npz'), **out)
    print('learner.npz written,', len(out), 'arrays')


if __name__ == '__main__':
    main()

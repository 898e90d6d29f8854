#!/usr/bin/env python3
"""Known answers of the reference's PPO COLLECTOR, produced by running its own class: controllers/ppo/ppo.py `PPO.train_step`
(:259-303) on the reference's Quadrotor (2-D tracking, 10-step episodes so that time-limit truncations occur next to real
terminations), 4 envs x 30 steps; `PPOAgent.update` is intercepted, so what is recorded is the rollout the reference hands to it:
obs, act, rew (with gamma * terminal_v already added in place, ppo_utils.py:389), mask, v, logp, terminal_v (the critic's value of the
TERMINAL observation where TimeLimit.truncated), ret, adv (normalised over the batch, ppo.py:300) — plus every transition and the
initial weights, so that tests/replay_env.py can replay the collection through this repo's collector.

    python tests/golden/make_ppo_collector.py       (build container only: needs /root/reference) -> ppo_collector.npz
"""
import functools
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden import make_adversarial as A  # noqa: E402  (stubs, tensorboard stand-in, reference imports)

import torch  # noqa: E402
import yaml  # noqa: E402

OVER = dict(episode_len_sec=0.2, randomized_init=True, done_on_out_of_bound=True)


def main():
    import safe_control_gym.controllers.ppo.ppo as mod
    cfg = yaml.safe_load(open(os.path.join(A.REF, 'safe_control_gym/controllers/ppo/ppo.yaml')))
    cfg.update(hidden_dim=16, activation='tanh', rollout_batch_size=4, rollout_steps=30, use_gae=True, gamma=0.98, gae_lambda=0.9,
               num_workers=1, tensorboard=False, norm_obs=False, norm_reward=False)
    tc = yaml.safe_load(open(os.path.join(A.REF, 'examples/rl/config_overrides/quadrotor_2D/quadrotor_2D_track.yaml')))['task_config']
    tc.update(OVER)
    tc.pop('seed', None)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        env_func = functools.partial(A.make, 'quadrotor', output_dir=tmp, **tc)
        torch.manual_seed(8)
        ctrl = mod.PPO(env_func, training=True, output_dir=tmp, use_gpu=False, seed=2, **cfg)
        ctrl.reset()
        out['obs0'] = np.asarray(ctrl.obs, dtype=float).copy()
        out.update(A.flat_sd(ctrl.agent.ac.state_dict(), 'init'))
        out['gamma_lambda'] = np.array([ctrl.gamma, ctrl.gae_lambda])
        steps, env = [], ctrl.env
        orig = env.__class__.step

        def rec_step(act):
            nxt, rew, done, info = orig(env, act)
            trunc, term = np.zeros(len(done), dtype=bool), np.zeros_like(nxt)
            for i, inf in enumerate(info['n']):
                if 'terminal_info' in inf:
                    term[i] = inf['terminal_observation']
                    trunc[i] = bool(inf['terminal_info'].get('TimeLimit.truncated', False))
            steps.append({'act': np.asarray(act, dtype=float).copy(), 'next_obs': nxt.copy(), 'rew': np.asarray(rew, dtype=float).copy(),
                          'done': np.asarray(done).copy(), 'trunc': trunc, 'term_obs': term})
            return nxt, rew, done, info
        env.step = rec_step
        got = {}

        def fake_update(rollouts, device=None):
            got['buf'] = rollouts
            return {}
        ctrl.agent.update = fake_update
        torch.manual_seed(31)
        ctrl.train_step()
        for k in steps[0]:
            out[f'transitions/{k}'] = np.stack([s[k] for s in steps])
        out.update(A.buf_arrays(got['buf'], 'buffer'))
        out['total_steps'] = np.array(ctrl.total_steps)
    d, tr = out['transitions/done'], out['transitions/trunc']
    print('steps', len(steps), 'dones', int(d.sum()), 'truncations', int(tr.sum()), 'terminations', int((d & ~tr).sum()),
          'max |terminal_v|', float(np.abs(out['buffer/terminal_v']).max()))
    assert tr.sum() > 0 and (d & ~tr).sum() > 0 and np.abs(out['buffer/terminal_v']).max() > 0
    np.savez_compressed(os.path.join(HERE, 'ppo_collector.npz'), **out)
    print('ppo_collector.npz written,', len(out), 'arrays')


if __name__ == '__main__':
    main()

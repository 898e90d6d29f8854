#!/usr/bin/env python3
"""Known answers of the reference's SAC COLLECTOR, produced by running its own class: controllers/sac/sac.py `SAC.train_step`
(:269-335) on the reference's Quadrotor (2-D tracking, 10-step episodes so that time-limit truncations occur, `done_on_out_of_bound`
on so that real terminations occur too), 4 envs x 40 vector steps, no gradient updates (train_interval beyond the run).  Recorded:
every transition `env.step` returned (tests/replay_env.py replays them), the actions the reference fed (uniform warm-up draws, then
its policy's samples) and what its SACBuffer holds afterwards — obs, act, rew, and the TRUE next_obs / mask of the time-limit fix-up
(:287-305: a truncated episode stores the terminal observation with mask 1, a terminated one the post-reset observation with mask 0).

    python tests/golden/make_sac_collector.py       (build container only: needs /root/reference) -> sac_collector.npz
    python tests/golden/make_sac_collector.py --norm    the same run with `norm_obs: True, norm_reward: True` (sac.py:75-81: the running
        normalisers around env.step — next_obs, then the reward, then the truncated envs' terminal observations, each call updating the
        statistics) -> sac_collector_norm.npz, which also holds the normalisers' state afterwards (mean / var / count, running returns)
"""
import functools
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden import make_adversarial as A  # noqa: E402  (stubs, tensorboard stand-in, Recorder, reference imports)

import torch  # noqa: E402
import yaml  # noqa: E402

OVER = dict(episode_len_sec=0.2, randomized_init=True, done_on_out_of_bound=True)


def main(norm=False):
    import safe_control_gym.controllers.sac.sac as mod
    cfg = yaml.safe_load(open(os.path.join(A.REF, 'safe_control_gym/controllers/sac/sac.yaml')))
    cfg.update(hidden_dim=16, rollout_batch_size=4, warm_up_steps=8, train_interval=10 ** 9, max_buffer_size=120, num_workers=1,
               tensorboard=False, norm_obs=norm, norm_reward=norm, clip_obs=4.0, clip_reward=2.0)
    tc = yaml.safe_load(open(os.path.join(A.REF, 'examples/rl/config_overrides/quadrotor_2D/quadrotor_2D_track.yaml')))['task_config']
    tc.update(OVER)
    tc.pop('seed', None)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        env_func = functools.partial(A.make, 'quadrotor', output_dir=tmp, **tc)
        torch.manual_seed(4)
        ctrl = mod.SAC(env_func, training=True, output_dir=tmp, use_gpu=False, seed=6, **cfg)
        raw0 = []
        reset0 = ctrl.env.reset
        ctrl.env.reset = lambda *a, **k: (lambda r: (raw0.append(np.asarray(r[0], dtype=float).copy()), r)[1])(reset0(*a, **k))
        ctrl.reset()
        ctrl.env.reset = reset0
        out['obs0'] = raw0[0]                               # what env.reset returned (ctrl.obs is its normalised image under --norm)
        steps, T = [], 40
        env = ctrl.env
        orig = env.__class__.step

        def rec_step(act):
            nxt, rew, done, info = orig(env, act)
            trunc = np.zeros(len(done), dtype=bool)
            term = np.zeros_like(nxt)
            for i, inf in enumerate(info['n']):
                if 'terminal_info' in inf:
                    term[i] = inf['terminal_observation']
                    trunc[i] = bool(inf['terminal_info'].get('TimeLimit.truncated', False))
            steps.append({'act': np.asarray(act, dtype=float).copy(), 'next_obs': nxt.copy(), 'rew': np.asarray(rew, dtype=float).copy(),
                          'done': np.asarray(done).copy(), 'trunc': trunc, 'term_obs': term})
            return nxt, rew, done, info
        env.step = rec_step
        for _ in range(T):
            ctrl.train_step()
        for k in steps[0]:
            out[f'transitions/{k}'] = np.stack([s[k] for s in steps])
        b = ctrl.buffer
        for k in ('obs', 'act', 'rew', 'next_obs', 'mask'):
            out[f'buffer/{k}'] = np.asarray(b.__dict__[k], dtype=np.float64).copy()
        out['buffer/pos_size'] = np.array([b.pos, b.buffer_size])
        out['total_steps'] = np.array(ctrl.total_steps)
        if norm:
            o, r = ctrl.obs_normalizer, ctrl.reward_normalizer
            out['norm/obs_mean'], out['norm/obs_var'], out['norm/obs_count'] = o.rms.mean.copy(), o.rms.var.copy(), np.array(o.rms.count)
            out['norm/rew_var'], out['norm/rew_count'], out['norm/ret'] = np.array(r.rms.var), np.array(r.rms.count), r.ret.copy()
            out['final_obs'] = np.asarray(ctrl.obs, dtype=float).copy()
    d, tr = out['transitions/done'], out['transitions/trunc']
    print('vector steps', T, 'dones', int(d.sum()), 'truncations', int(tr.sum()), 'terminations', int((d & ~tr).sum()),
          'buffer pos/size', out['buffer/pos_size'].tolist())
    assert tr.sum() > 0 and (d & ~tr).sum() > 0
    name = 'sac_collector_norm.npz' if norm else 'sac_collector.npz'
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, 'written,', len(out), 'arrays')


if __name__ == '__main__':
    main(norm='--norm' in sys.argv[1:])

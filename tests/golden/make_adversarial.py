#!/usr/bin/env python3
"""Known answers for SURVEY §8f-3 (adversary / safety-layer collectors), produced by running the reference's OWN classes.

    python tests/golden/make_adversarial.py          (build container only: needs /root/reference)

* controllers/rarl/rarl.py — `RARL.collect_rollouts(adversary=False / True)` (:349-428) on the reference's Quadrotor (2-D
  tracking, `adversary_disturbance: dynamics`, 10-step episodes so that time-limit truncations and their terminal-value
  bootstrap occur), 4 envs x 12 steps, then `PPOAgent.update` of the side that collected.
* controllers/rarl/rap.py — `RAP.collect_rollouts` (:349-470): sorted adversary index per env, every adversary acts on
  its group, the split adversary rollouts.
* controllers/safe_explorer/safe_explorer_utils.py — `SafetyLayer.get_safe_action`, `.update` (pre-training loss, one
  Adam per constraint model), `ConstraintBuffer.push / sample`; safe_ppo_utils.py — `SafePPOAgent.update` with the safety
  layer inside the actor (constraint values as policy input).
The envs run on tests/golden/ref_stubs.py (pybullet stand-in = oracle/bullet.py); what is recorded is the COLLECTOR's
bookkeeping on whatever transitions came back — every transition is stored, so the checked code replays them
(tests/replay_env.py) and must reproduce the buffers: which policy's action / value / log-prob goes where, the sign of
the adversary's reward, whose critic bootstraps a truncated episode, returns / advantages and their normalisation.
Output: tests/golden/adversarial.npz.
"""
import functools
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden import ref_stubs  # noqa: E402

ref_stubs.install()
tb = types.ModuleType('torch.utils.tensorboard')
tb.SummaryWriter = type('SummaryWriter', (), {'__init__': lambda s, *a, **k: None, 'add_scalar': lambda s, *a, **k: None,
                                              'close': lambda s: None, 'flush': lambda s: None})
sys.modules.setdefault('torch.utils.tensorboard', tb)

import torch  # noqa: E402
import yaml  # noqa: E402
from gymnasium.spaces import Box  # noqa: E402

import safe_control_gym.envs  # noqa: E402,F401
from safe_control_gym.utils.registration import make  # noqa: E402

REF = ref_stubs.REFERENCE_ROOT


def flat_sd(sd, prefix):
    return {f'{prefix}/{k}': v.detach().cpu().numpy().copy() for k, v in sd.items()}


class Recorder:
    """Wraps the controller's vec env: stores every transition `env.step` returns and every adversary action handed to
    `env_method('set_adversary_control', ...)`, plus the processed value the envs ended up with (benchmark_env.py:216-228)."""

    def __init__(self, ctrl):
        self.env = ctrl.env
        self.steps, self.adv_raw, self.adv_applied, self.resets = [], [], [], []
        step, env_method, reset = self.env.step, self.env.env_method, self.env.reset
        inner = self.env.venv if hasattr(self.env, 'venv') else self.env

        def rec_step(act):
            self.adv_applied.append(np.stack([np.asarray(e.adv_action, dtype=float) for e in inner.envs]))
            out = step(act)
            next_obs, rew, done, info = out
            trunc = np.zeros(len(done), dtype=bool)
            term = np.zeros_like(next_obs)
            for i, inf in enumerate(info['n']):
                if 'terminal_info' in inf:
                    term[i] = inf['terminal_observation']
                    trunc[i] = bool(inf['terminal_info'].get('TimeLimit.truncated', False))
            self.steps.append({'act': np.asarray(act, dtype=float).copy(), 'next_obs': next_obs.copy(), 'rew': np.asarray(rew, dtype=float).copy(),
                               'done': np.asarray(done).copy(), 'trunc': trunc, 'term_obs': term})
            return out

        def rec_method(name, args_list, **kw):
            if name == 'set_adversary_control':
                self.adv_raw.append(np.stack([np.asarray(a[0], dtype=float) for a in args_list]))
            return env_method(name, args_list, **kw)

        def rec_reset(*a, **k):
            out = reset(*a, **k)
            self.resets.append(np.asarray(out[0], dtype=float).copy())
            return out

        self.env.step, self.env.env_method, self.env.reset = rec_step, rec_method, rec_reset

    def take(self):
        out = {k: np.stack([s[k] for s in self.steps]) for k in self.steps[0]}
        out['adv_raw'] = np.stack(self.adv_raw)
        out['adv_applied'] = np.stack(self.adv_applied)
        self.steps, self.adv_raw, self.adv_applied = [], [], []
        return out


def task_config():
    over = yaml.safe_load(open(os.path.join(REF, 'examples/rl/config_overrides/quadrotor_2D/quadrotor_2D_track.yaml')))['task_config']
    over.update(episode_len_sec=0.2, adversary_disturbance='dynamics', adversary_disturbance_scale=0.05, adversary_disturbance_offset=0.01)
    over.pop('seed', None)
    # start in bounds and stay there for 10 steps: truncations (not failures) must occur
    over.update(randomized_init=False, done_on_out_of_bound=True)
    return over


def buf_arrays(buf, prefix, keys=('obs', 'act', 'rew', 'mask', 'v', 'logp', 'terminal_v', 'ret', 'adv')):
    return {f'{prefix}/{k}': np.asarray(getattr(buf, k), dtype=np.float64).copy() for k in keys}


def rarl_case(out, tmp):
    import safe_control_gym.controllers.rarl.rarl as mod
    cfg = yaml.safe_load(open(os.path.join(REF, 'safe_control_gym/controllers/rarl/rarl.yaml')))
    cfg.update(hidden_dim=16, rollout_batch_size=4, rollout_steps=12, opt_epochs=2, mini_batch_size=16, use_gae=True, actor_lr=3e-3,
               critic_lr=1e-3, num_workers=1, tensorboard=False, agent_iterations=1, adversary_iterations=1)
    tc = task_config()
    env_func = functools.partial(make, 'quadrotor', output_dir=tmp, **tc)
    torch.manual_seed(11)
    ctrl = mod.RARL(env_func, training=True, output_dir=tmp, use_gpu=False, seed=3, **cfg)
    ctrl.reset()
    rec = Recorder(ctrl)
    out['rarl/obs0'] = np.asarray(ctrl.obs, dtype=float).copy()
    out.update(flat_sd(ctrl.agent.ac.state_dict(), 'rarl/agent_init'))
    out.update(flat_sd(ctrl.adversary.ac.state_dict(), 'rarl/adversary_init'))
    out['rarl/gamma_lambda'] = np.array([ctrl.gamma, ctrl.gae_lambda])
    out['rarl/adv_scale_offset'] = np.array([tc['adversary_disturbance_scale'], tc['adversary_disturbance_offset']])
    for tag, adversary in (('agent', False), ('adversary', True)):
        torch.manual_seed(100 + int(adversary))
        ro = ctrl.collect_rollouts(adversary=adversary)
        tr = rec.take()
        out.update({f'rarl/{tag}/transitions/{k}': v for k, v in tr.items()})
        out.update(buf_arrays(ro, f'rarl/{tag}/buffer'))
        np.random.seed(21 + int(adversary))
        n = ro.max_length * ro.batch_size
        out[f'rarl/{tag}/perms'] = np.stack([np.random.permutation(n) for _ in range(cfg['opt_epochs'])])
        np.random.seed(21 + int(adversary))
        learner = ctrl.adversary if adversary else ctrl.agent
        res = learner.update(ro)
        out[f'rarl/{tag}/results'] = np.array([res['policy_loss'], res['value_loss'], res['entropy_loss'], res['approx_kl']])
        out.update(flat_sd(learner.ac.state_dict(), f'rarl/{tag}/final'))
        print(f'rarl {tag}: truncated rows', int(tr['trunc'].sum()), 'done rows', int(tr['done'].sum()), 'results', out[f'rarl/{tag}/results'])
    return ctrl             # (closed by main() at the end: the pybullet stand-in numbers its clients per process)


def rap_case(out, tmp):
    import safe_control_gym.controllers.rarl.rap as mod
    cfg = yaml.safe_load(open(os.path.join(REF, 'safe_control_gym/controllers/rarl/rap.yaml')))
    cfg.update(hidden_dim=16, rollout_batch_size=6, rollout_steps=12, opt_epochs=1, mini_batch_size=8, use_gae=True, num_workers=1,
               tensorboard=False, num_adversaries=3)
    tc = task_config()
    env_func = functools.partial(make, 'quadrotor', output_dir=tmp, **tc)
    torch.manual_seed(12)
    # upstream's RAP does not define BaseController's abstract `select_action` (rap.py has no such method), so the class as
    # shipped cannot be instantiated; the collector under test is untouched by adding a stub for it
    runnable = type('RAPRunnable', (mod.RAP,), {'select_action': lambda self, obs, info=None: None})
    ctrl = runnable(env_func, training=True, output_dir=tmp, use_gpu=False, seed=4, **cfg)
    ctrl.reset()
    rec = Recorder(ctrl)
    out.update(flat_sd(ctrl.agent.ac.state_dict(), 'rap/agent_init'))
    for k, a in enumerate(ctrl.adversaries):
        out.update(flat_sd(a.ac.state_dict(), f'rap/adversary{k}_init'))
    np.random.seed(5)
    idx = sorted(np.random.randint(cfg['num_adversaries'], size=cfg['rollout_batch_size']))      # what collect_rollouts draws (rap.py:356)
    out['rap/adv_indices'] = np.asarray(idx)
    np.random.seed(5)
    torch.manual_seed(102)
    ro, splits = ctrl.collect_rollouts()
    tr = rec.take()
    out['rap/obs0'] = rec.resets[-1]
    out.update({f'rap/transitions/{k}': v for k, v in tr.items()})
    out.update(buf_arrays(ro, 'rap/agent/buffer'))
    for k, split in splits:
        out.update(buf_arrays(split, f'rap/adversary{int(k)}/buffer'))
    out['rap/split_ids'] = np.asarray([int(k) for k, _ in splits])
    print('rap: adversary per env', idx, 'truncated rows', int(tr['trunc'].sum()))
    return ctrl


def safety_case(out):
    from safe_control_gym.controllers.safe_explorer import safe_explorer_utils as seu
    from safe_control_gym.controllers.safe_explorer import safe_ppo_utils as spu
    O, A, Cn = 5, 2, 3
    obs_space, act_space = Box(-1, 1, (O,)), Box(-1, 1, (A,))
    torch.manual_seed(31)
    layer = seu.SafetyLayer(obs_space, act_space, hidden_dim=16, num_constraints=Cn, lr=3e-3, slack=[0.05, 0.0, 0.1])
    out.update(flat_sd(layer.constraint_models.state_dict(), 'safety/init'))
    rng = np.random.default_rng(17)
    obs, act, c = rng.normal(0, 1, (48, O)).astype(np.float32), rng.normal(0, 1, (48, A)).astype(np.float32), (0.3 * rng.normal(0, 1, (48, Cn))).astype(np.float32)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        safe = layer.get_safe_action(torch.as_tensor(obs), torch.as_tensor(act), torch.as_tensor(c))
    out['safety/proj/obs'], out['safety/proj/act'], out['safety/proj/c'], out['safety/proj/safe'] = obs, act, c, safe.detach().numpy()
    # ConstraintBuffer ring (wraps) + three updates on fixed index batches
    buf = seu.ConstraintBuffer(obs_space, act_space, Cn, max_size=100, batch_size=32)
    for _ in range(3):
        n = 40
        buf.push({'obs': rng.normal(0, 1, (n, O)), 'act': rng.normal(0, 1, (n, A)), 'c': rng.normal(0, 0.3, (n, Cn)),
                  'c_next': rng.normal(0, 0.3, (n, Cn))})
    for k in ('obs', 'act', 'c', 'c_next'):
        out[f'safety/buffer/{k}'] = buf.__dict__[k].copy()
    out['safety/buffer/pos_size'] = np.array([buf.pos, buf.buffer_size])
    idx = np.stack([rng.permutation(100)[:32] for _ in range(3)])
    out['safety/indices'] = idx
    layer.train()
    losses = []
    for ind in idx:
        batch = {k: torch.as_tensor(v) for k, v in buf.sample(ind).items()}
        res = layer.update(batch)
        losses.append([res[f'constraint_{i}_loss'] for i in range(Cn)])
    out['safety/losses'] = np.array(losses)
    out.update(flat_sd(layer.constraint_models.state_dict(), 'safety/final'))
    # SafePPOAgent.update with the (now trained) layer filtering the actor's mean, constraint values as input
    torch.manual_seed(32)
    agent = spu.SafePPOAgent(obs_space, act_space, hidden_dim=16, use_clipped_value=False, clip_param=0.2, target_kl=0.05, entropy_coef=0.01,
                             actor_lr=3e-3, critic_lr=1e-3, opt_epochs=2, mini_batch_size=32, action_modifier=layer.get_safe_action)
    out.update(flat_sd(agent.ac.state_dict(), 'safeppo/init'))
    T, N = 8, 8
    rb = spu.SafePPOBuffer(obs_space, act_space, Cn, T, N)
    ob = rng.normal(0, 1, (T + 1, N, O)).astype(np.float32)
    cc = (0.3 * rng.normal(0, 1, (T, N, Cn))).astype(np.float32)
    from safe_control_gym.controllers.ppo import ppo_utils
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for t in range(T):
            torch.manual_seed(200 + t)
            with torch.no_grad():
                a, v, logp = agent.ac.step(torch.as_tensor(ob[t]), c=torch.as_tensor(cc[t]))
            rew = rng.normal(0, 1, (N,))
            mask = (rng.uniform(size=N) > 0.1).astype(np.float32)
            rb.push({'obs': ob[t], 'act': a, 'rew': rew, 'mask': mask, 'v': v, 'logp': logp, 'terminal_v': np.zeros(N), 'c': cc[t]})
        last_val = agent.ac.critic(torch.as_tensor(ob[T])).detach().numpy()
        ret, adv = ppo_utils.compute_returns_and_advantages(rb.rew, rb.v, rb.mask, rb.terminal_v, last_val, gamma=0.99, use_gae=True, gae_lambda=0.95)
        rb.ret = ret
        rb.adv = (adv - adv.mean()) / (adv.std() + 1e-6)
        data = rb.get()
        for k in ('obs', 'act', 'logp', 'adv', 'ret', 'v', 'c'):
            out[f'safeppo/data/{k}'] = data[k].numpy()
        np.random.seed(41)
        out['safeppo/perms'] = np.stack([np.random.permutation(T * N) for _ in range(2)])
        np.random.seed(41)
        res = agent.update(rb)
    out['safeppo/results'] = np.array([res['policy_loss'], res['value_loss'], res['entropy_loss'], res['approx_kl']])
    out.update(flat_sd(agent.ac.state_dict(), 'safeppo/final'))
    print('safety: losses', out['safety/losses'].tolist(), 'safeppo results', out['safeppo/results'])


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        ctrls = [rarl_case(out, tmp), rap_case(out, tmp)]
        for c in ctrls:
            c.close()
    safety_case(out)
    np.savez_compressed(os.path.join(HERE, 'adversarial.npz'), **out)
    print('adversarial.npz written,', len(out), 'arrays')


if __name__ == '__main__':
    main()

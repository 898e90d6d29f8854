"""RARL and RAP (robust adversarial RL, single adversary / adversary population) on the HIP rollout engine.

Mirrors /root/reference/safe_control_gym/controllers/rarl/rarl.py and rap.py:
  rarl.py:267-281   train_step = update_agent (agent_iterations x [collect, PPO update]) then update_adversary
  rarl.py:349-428   collect_rollouts: BOTH policies act on every control step; the adversary's action goes through
                    `set_adversary_control` (benchmark_env.py:216-228: clip to [-1, 1], scale, offset) into the env's action
                    or dynamics disturbance channel; the protagonist learns from +reward, the adversary from -reward
  rarl.py:430-467   update_agent / update_adversary
  rap.py:257-281    train_step = ONE collection, then the protagonist's update and one update per sampled adversary
  rap.py:349-470    collect_rollouts: env reset, a SORTED adversary index per env (contiguous groups), every adversary acts on
                    its group, advantages normalised over the whole batch, the adversary rollout split by group
The env needs `adversary_disturbance: 'action' | 'dynamics'` in its task config (the kernel's adversary channel).

Both collectors are HIP-graph captured (static shapes): T x (protagonist step, adversary step(s), env kernel with the
adversary channel, statistics) + both agents' bootstrap / GAE — one replay per collection.  RAP evaluates every member of
the population on the whole batch and selects per env by index (population sizes are 2-4: rap.yaml:7), which keeps the
shapes static while the groups are re-drawn every iteration.
"""
import time

import numpy as np
import torch

from safe_control_gym_amd import parallel
from safe_control_gym_amd.ppo import PPO, PPOAgent, PPOConfig


def _normalised(adv, moments):
    parallel.all_reduce_sum_(moments)
    mean = moments[0] / moments[2]
    std = torch.sqrt(torch.clamp(moments[1] / moments[2] - mean * mean, min=0.0))
    return (adv - mean) / (std + 1e-6)


class _TwoSided(PPO):
    """Shared by RARL and RAP: a second set of (action, value, log-prob) buffers for the adversary side and the captured
    collection of both sides."""

    def __init__(self, env, cfg: PPOConfig, seed=0):
        if env.spec.adversary_disturbance is None:
            raise ValueError(f'{type(self).__name__} needs an env with adversary_disturbance set (benchmark_env.py:216-228)')
        cfg.extra = dict(cfg.extra, fused_rollout=False)      # two policies act per step: the torch-policy collector
        super().__init__(env, cfg, seed)
        self.adv_dim = env.spec.adversary_dim
        f = dict(device=self.device, dtype=torch.float32)
        self.act_adv = torch.zeros(self.T, self.N, self.adv_dim, **f)
        self.v_adv = torch.zeros(self.T, self.N, **f)
        self.logp_adv = torch.zeros(self.T, self.N, **f)
        self._two_graph = None

    def _adversary_step(self, obs):
        raise NotImplementedError

    def _adversary_critic(self, x):
        raise NotImplementedError

    def _collect_body(self):
        ac, env = self.agent.ac, self.env
        for t in range(self.T):
            act, v, logp = ac.step(self.obs[t])
            a_adv, v_adv, logp_adv = self._adversary_step(self.obs[t])
            self.act[t], self.v[t], self.logp[t] = act, v, logp
            self.act_adv[t], self.v_adv[t], self.logp_adv[t] = a_adv, v_adv, logp_adv
            env.set_adversary_control(self.act_adv[t])        # clip, scale, offset -> env._adv
            out, c_out = self._slots[t]
            env.step_tensors(self.act[t], env._adv, out=out, c_out=c_out)
            env._adv = None
            if self._normalise:
                self.obs[t + 1].copy_(self.obs_normalizer(self.obs[t + 1]))
                self.rew[t].copy_(self.reward_normalizer(self.rew[t], self.done[t]))
            d = self.done[t].to(torch.float32)
            self.ep_count += d.sum()
            self.ep_return_sum += (out.fin_return * d).sum()
            self.ep_length_sum += (out.fin_length * d).sum()
            self.ep_violation_sum += (out.fin_violation * d).sum()

    def _both_returns(self, dense):
        ag = self._returns_body(dense)
        ad = self._returns_body(dense, self._adversary_critic, -self.rew, self.v_adv)
        return ag, ad

    def _collect_both(self):
        """One collection + the (returns, advantages, moments) of both sides; a graph replay when graphs are on."""
        if self._graph_rollout:
            epoch = getattr(self.env, 'seed_epoch', 0)
            if self._two_graph is None or self._two_graph[2] != epoch:
                dev = self.device
                s = torch.cuda.Stream(dev)
                s.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(s):
                    for _ in range(3):          # warm up the torch ops only (no env step: the simulation must not advance)
                        self.agent.ac.step(self.obs[0])
                        self._adversary_step(self.obs[0])
                        self._both_returns(True)
                torch.cuda.current_stream(dev).wait_stream(s)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._collect_body()
                    out = self._both_returns(True)
                self._two_graph = (g, out, epoch)
            self._two_graph[0].replay()
            (r0, a0, m0), (r1, a1, m1) = self._two_graph[1]
            res = (r0, a0, m0.clone()), (r1, a1, m1.clone())
        else:
            self._collect_body()
            res = self._both_returns(False)
        self.total_steps += self.T * self.N * parallel.world_size()
        return res

    def _data(self, adversary, ret, adv, lo=0, hi=None):
        hi = self.N if hi is None else hi
        n = hi - lo
        M = self.T * n
        sl = lambda t: t[:, lo:hi]                          # noqa: E731
        return {'obs': sl(self.obs[:self.T]).reshape(M, self.obs_dim),
                'act': sl(self.act_adv if adversary else self.act).reshape(M, -1),
                'logp': sl(self.logp_adv if adversary else self.logp).reshape(M), 'adv': sl(adv).reshape(M),
                'ret': sl(ret).reshape(M), 'v': sl(self.v_adv if adversary else self.v).reshape(M)}


class RARL(_TwoSided):
    def __init__(self, env, cfg: PPOConfig, seed=0, agent_iterations=1, adversary_iterations=1):
        super().__init__(env, cfg, seed)
        self.agent_iterations, self.adversary_iterations = int(agent_iterations), int(adversary_iterations)
        self.adversary = PPOAgent(self.obs_dim, self.adv_dim, cfg, self.device)

    def _adversary_step(self, obs):
        return self.adversary.ac.step(obs)

    def _adversary_critic(self, x):
        return self.adversary.ac.critic(x)

    def collect(self):
        return self._collect_both()

    def _one_update(self, adversary):
        (ret, adv, moments) = self._collect_both()[1 if adversary else 0]
        with torch.no_grad():
            adv = _normalised(adv, moments)
        res = (self.adversary if adversary else self.agent).update(self._data(adversary, ret, adv))
        self.obs[0].copy_(self.obs[self.T])
        return res

    def train_step(self):
        t0 = time.perf_counter()
        out = {}
        for name, n, adversary in (('', self.agent_iterations, False), ('_adv', self.adversary_iterations, True)):
            acc = {}
            for _ in range(n):
                for k, v in self._one_update(adversary).items():
                    acc[k] = acc.get(k, 0.0) + v / n
            out.update({k + name: v for k, v in acc.items()})
        out.update({'step': self.total_steps, 'elapsed_time': time.perf_counter() - t0})
        return out


class RAP(_TwoSided):
    """rap.py: a population of adversaries; every collection re-draws which adversary faces which env."""

    def __init__(self, env, cfg: PPOConfig, seed=0, num_adversaries=2):
        super().__init__(env, cfg, seed)
        self.num_adversaries = int(num_adversaries)
        # the groups change size every iteration: the adversaries' minibatch updates run on the PyTorch path (no captured
        # shapes); the protagonist, whose batch is the whole rollout, keeps the fused / graphed update
        adv_cfg = PPOConfig(**{**cfg.__dict__, 'extra': dict(cfg.extra, cuda_graphs=False)})
        self.adversaries = [PPOAgent(self.obs_dim, self.adv_dim, adv_cfg, self.device) for _ in range(self.num_adversaries)]
        self._rng = np.random.RandomState(seed)               # rap.py:356 draws from NumPy's global stream
        self.adv_index = torch.zeros(self.N, dtype=torch.long, device=self.device)
        self.groups = []

    def _select(self, per_adv):
        """per_adv: list (one per adversary) of [N, ...] tensors (or [T*N, ...], time-major) -> rows of each env's adversary."""
        st = torch.stack(per_adv, 0)
        idx = self.adv_index.repeat(st.shape[1] // self.N)
        idx = idx.view(1, -1, *([1] * (st.dim() - 2))).expand(1, *st.shape[1:])
        return st.gather(0, idx)[0]

    def _adversary_step(self, obs):
        outs = [a.ac.step(obs) for a in self.adversaries]
        return tuple(self._select([o[k] for o in outs]) for k in range(3))

    def _adversary_critic(self, x):
        return self._select([a.ac.critic(x) for a in self.adversaries])

    def _both_returns(self, dense):
        return super()._both_returns(True)                  # (the per-env selection needs whole [T x N] batches)

    def collect(self):
        # rap.py:356-357: sorted adversary indices, one per env -> contiguous groups
        idx = np.sort(self._rng.randint(self.num_adversaries, size=self.N))
        groups, starts = np.unique(idx, return_index=True)
        ends = np.concatenate([starts[1:], [self.N]])
        self.groups = [(int(g), int(s), int(e)) for g, s, e in zip(groups, starts, ends)]
        self.adv_index.copy_(torch.as_tensor(idx, device=self.device))
        self.obs[0].copy_(self.obs_normalizer(self.env.reset_tensors()))      # rap.py:361: every collection starts from a reset
        return self._collect_both()

    def train_step(self):
        t0 = time.perf_counter()
        (ret, adv, mom), (ret_a, adv_a, mom_a) = self.collect()
        with torch.no_grad():
            adv, adv_a = _normalised(adv, mom), _normalised(adv_a, mom_a)       # both over the whole batch (rap.py:425,448)
        out = dict(self.agent.update(self._data(False, ret, adv)))
        for k, s, e in self.groups:
            res = self.adversaries[k].update(self._data(True, ret_a, adv_a, s, e))
            out.update({f'{name}_adv{k}': v for name, v in res.items()})
        out.update({'step': self.total_steps, 'elapsed_time': time.perf_counter() - t0, 'adv_indices': [g[0] for g in self.groups]})
        return out

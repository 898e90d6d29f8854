"""RARL (robust adversarial RL) on the HIP rollout engine.

Mirrors /root/reference/safe_control_gym/controllers/rarl/rarl.py:
  :267-281   train_step = update_agent (agent_iterations x [collect, PPO update]) then update_adversary
  :349-428   collect_rollouts: BOTH policies act on every control step; the adversary's action goes through
             `set_adversary_control` (benchmark_env.py:216-228: clip to [-1, 1], scale, offset) into the env's action or
             dynamics disturbance channel; the protagonist learns from +reward, the adversary from -reward
  :430-467   update_agent / update_adversary
The env needs `adversary_disturbance: 'action' | 'dynamics'` in its task config (the kernel's adversary channel).
"""
import time

import torch

from safe_control_gym_amd import parallel
from safe_control_gym_amd.ppo import PPO, PPOAgent, PPOConfig


class RARL(PPO):
    def __init__(self, env, cfg: PPOConfig, seed=0, agent_iterations=1, adversary_iterations=1):
        if env.spec.adversary_disturbance is None:
            raise ValueError('RARL needs an env with adversary_disturbance set (benchmark_env.py:216-228)')
        cfg.extra = dict(cfg.extra, cuda_graphs=False)        # two policies share the rollout: collected eagerly
        super().__init__(env, cfg, seed)
        self.agent_iterations, self.adversary_iterations = int(agent_iterations), int(adversary_iterations)
        self.adv_dim = env.spec.adversary_dim
        self.adversary = PPOAgent(self.obs_dim, self.adv_dim, cfg, self.device)
        f = dict(device=self.device, dtype=torch.float32)
        self.act_adv = torch.zeros(self.T, self.N, self.adv_dim, **f)
        self.v_adv = torch.zeros(self.T, self.N, **f)
        self.logp_adv = torch.zeros(self.T, self.N, **f)

    def _collect_body(self):
        ac, adv_ac, env = self.agent.ac, self.adversary.ac, self.env
        for t in range(self.T):
            act, v, logp = ac.step(self.obs[t])
            a_adv, v_adv, logp_adv = adv_ac.step(self.obs[t])
            self.act[t], self.v[t], self.logp[t] = act, v, logp
            self.act_adv[t], self.v_adv[t], self.logp_adv[t] = a_adv, v_adv, logp_adv
            env.set_adversary_control(a_adv)                  # clip, scale, offset -> env._adv
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

    def _one_update(self, adversary):
        self.collect()
        if adversary:
            ret, adv, moments = self._returns_body(False, self.adversary.ac.critic, -self.rew, self.v_adv)
        else:
            ret, adv, moments = self._returns_body(False)
        with torch.no_grad():
            parallel.all_reduce_sum_(moments)
            mean = moments[0] / moments[2]
            std = torch.sqrt(torch.clamp(moments[1] / moments[2] - mean * mean, min=0.0))
            adv = (adv - mean) / (std + 1e-6)
        M = self.T * self.N
        who = self.adversary if adversary else self.agent
        data = {'obs': self.obs[:self.T].reshape(M, self.obs_dim),
                'act': (self.act_adv if adversary else self.act).reshape(M, -1),
                'logp': (self.logp_adv if adversary else self.logp).reshape(M), 'adv': adv.reshape(M), 'ret': ret.reshape(M),
                'v': (self.v_adv if adversary else self.v).reshape(M)}
        res = who.update(data)
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

"""Safe-Explorer PPO (safety layer of Dalal et al. 2018) on the HIP rollout engine.

Mirrors /root/reference/safe_control_gym/controllers/safe_explorer/:
  safe_explorer_utils.py:15-176   SafetyLayer: one MLP g_i(obs) per state constraint with  c_i' ~ c_i + g_i(obs)·a ,
                                  closed-form projection  a* = a - lambda* g_i*  for the most violated constraint
  safe_explorer_utils.py:179-299  ConstraintBuffer
  safe_ppo.py:196-296,425-449     pre-training on random-action transitions (c, c_next with the TERMINAL constraint values
                                  for finished episodes)
  safe_ppo.py:299-345             train_step: the constraint values `c` of the current state are a policy input; the
                                  layer filters the MEAN of the action distribution (safe_ppo_utils.py:88-110)
What the env kernel provides: `c_values` of every step (pre-reset values where the episode ended — exactly upstream's
`terminal_info['constraint_values']`); the constraint values of a freshly reset env are evaluated from the returned state
by `EnvSpec.state_constraint_values`.
"""
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

from safe_control_gym_amd import parallel
from safe_control_gym_amd.ppo import MLP, PPO, PPOConfig


class SafetyLayer:
    def __init__(self, obs_dim, act_dim, num_constraints, hidden_dim=64, lr=1e-3, slack=0.05, device='cpu'):
        hidden = [hidden_dim] if isinstance(hidden_dim, int) else list(hidden_dim)
        self.num_constraints = int(num_constraints)
        self.device = torch.device(device)
        self.constraint_models = nn.ModuleList([MLP(obs_dim, act_dim, hidden) for _ in range(self.num_constraints)]).to(self.device)
        if isinstance(slack, (int, float)):
            slack = [slack] * self.num_constraints
        self.slack = torch.as_tensor(slack, dtype=torch.float32, device=self.device)
        self.optimizers = [torch.optim.Adam(m.parameters(), lr=lr) for m in self.constraint_models]

    def g(self, obs):
        """[B, C, A]: the learned sensitivities of every constraint to the action."""
        return torch.stack([m(obs) for m in self.constraint_models], dim=1)

    def compute_loss(self, batch):
        """Per-constraint L2 loss of the one-step linear model (safe_explorer_utils.py:89-108)."""
        pred = batch['c'] + (self.g(batch['obs']) * batch['act'][:, None, :]).sum(-1)
        return ((batch['c_next'] - pred) ** 2).mean(0)                    # [C]

    def update(self, batch):
        losses = self.compute_loss(batch)
        for opt in self.optimizers:
            opt.zero_grad()
        losses.sum().backward()                  # disjoint parameters: the same gradients as C separate backward passes
        for opt in self.optimizers:
            opt.step()
        return losses.detach()

    def get_safe_action(self, obs, act, c):
        """Eq. (5)-(6) of Dalal et al. 2018 for the constraint with the largest multiplier (safe_explorer_utils.py:120-176)."""
        g = self.g(obs)                                                    # [B, C, A]
        numer = (g * act[:, None, :]).sum(-1) + c + self.slack             # [B, C]
        denom = (g * g).sum(-1) + 1e-8
        mult = F.relu(numer / denom)
        max_mult, max_idx = mult.max(dim=-1)                               # (topk(., 1))
        max_g = g[torch.arange(g.shape[0], device=g.device), max_idx]      # [B, A]
        return act - max_mult[:, None] * max_g

    def state_dict(self):
        return {'constraint_models': self.constraint_models.state_dict(), 'optimizers': [o.state_dict() for o in self.optimizers]}

    def load_state_dict(self, sd):
        self.constraint_models.load_state_dict(sd['constraint_models'])
        for o, s in zip(self.optimizers, sd['optimizers']):
            o.load_state_dict(s)


class ConstraintBuffer:
    """Device ring of (obs, act, c, c_next) transitions."""

    def __init__(self, capacity, obs_dim, act_dim, num_constraints, device):
        f = dict(device=device, dtype=torch.float32)
        self.capacity = int(capacity)
        self.data = {'obs': torch.zeros(self.capacity, obs_dim, **f), 'act': torch.zeros(self.capacity, act_dim, **f),
                     'c': torch.zeros(self.capacity, num_constraints, **f), 'c_next': torch.zeros(self.capacity, num_constraints, **f)}
        self.pos, self.size = 0, 0

    def push(self, **items):
        n = items['obs'].shape[0]
        idx = (torch.arange(n, device=items['obs'].device) + self.pos) % self.capacity
        for k, v in items.items():
            self.data[k][idx] = v
        self.pos = (self.pos + n) % self.capacity
        self.size = min(self.size + n, self.capacity)

    def sampler(self, batch_size, generator=None):
        """Shuffled minibatches, drop last (ppo_utils.random_sample)."""
        perm = torch.randperm(self.size, device=self.data['obs'].device, generator=generator)
        for k in range(self.size // batch_size):
            idx = perm[k * batch_size:(k + 1) * batch_size]
            yield {name: t[idx] for name, t in self.data.items()}


class SafeExplorerPPO(PPO):
    def __init__(self, env, cfg: PPOConfig, seed=0, constraint_hidden_dim=64, constraint_lr=1e-3, constraint_slack=0.05,
                 constraint_batch_size=4096, constraint_buffer_size=1_000_000):
        self.C = env.spec.n_state_con_rows
        if self.C == 0:
            raise ValueError('Safe-Explorer needs an env with state constraints')
        cfg.extra = dict(cfg.extra, cuda_graphs=False)        # the safety layer sits inside the actor: collected eagerly
        super().__init__(env, cfg, seed)
        self.safety_layer = SafetyLayer(self.obs_dim, self.act_dim, self.C, constraint_hidden_dim, constraint_lr,
                                        constraint_slack, self.device)
        self.agent.ac.actor.action_modifier = self.safety_layer.get_safe_action
        self.constraint_batch_size = int(constraint_batch_size)
        self.constraint_buffer = ConstraintBuffer(constraint_buffer_size, self.obs_dim, self.act_dim, self.C, self.device)
        self.c_buf = torch.zeros(self.T, self.N, self.C, device=self.device)
        # per-step outputs incl. the env.state copy (needed for the constraint values of freshly reset envs)
        self._slots = [env.bind_outputs(obs=self.obs[t + 1], reward=self.rew[t], done=self.done[t], flags=self.flags[t],
                                        terminal_obs=self.term_obs[t], noisy_action=None, mse=None) for t in range(self.T)]
        self.obs[0].copy_(self.obs_normalizer(env.reset_tensors()))
        self.c = self._reset_c(env.out)

    def _reset_c(self, out, env=None):
        return (env or self.env).spec.state_constraint_values(out.state.t()).to(torch.float32)

    def _next_c(self, out, env=None):
        """c of the NEXT policy step and c_next of THIS transition (terminal values where the episode ended)."""
        c_step = out.c_values[:self.C].t().to(torch.float32)
        done = out.done.bool()
        c_now = torch.where(done[:, None], self._reset_c(out, env), c_step) if bool((env or self.env).auto_reset) else c_step
        return c_now, c_step

    # ---- pre-training of the constraint models (safe_ppo.py:196-296, :425-449)
    def collect_constraint_data(self, num_steps):
        env = self.env
        low = torch.as_tensor(env.spec.action_space.low, dtype=torch.float32, device=self.device)
        high = torch.as_tensor(env.spec.action_space.high, dtype=torch.float32, device=self.device)
        obs = self.obs_normalizer(env.reset_tensors()).clone()
        c = self._reset_c(env.out)
        steps = 0
        while steps < num_steps:
            act = low + (high - low) * torch.rand(self.N, self.act_dim, device=self.device)
            out = env.step_tensors(act)
            c_now, c_next = self._next_c(out)
            self.constraint_buffer.push(obs=obs, act=act, c=c, c_next=c_next)
            obs, c = self.obs_normalizer(out.obs).clone(), c_now
            steps += self.N * parallel.world_size()
        self.obs[0].copy_(obs)
        self.c = c

    def pretrain_step(self, steps_per_epoch, batch_size=None):
        """One epoch of upstream's pre-training (safe_ppo.py:281-297): fresh random-action transitions, one pass of shuffled
        minibatches over them, buffer cleared.  Returns the mean per-constraint losses."""
        self.obs_normalizer.unset_read_only()
        self.collect_constraint_data(steps_per_epoch)
        bs = min(int(batch_size or self.constraint_batch_size), self.constraint_buffer.size)
        acc, k = torch.zeros(self.C, device=self.device), 0
        for batch in self.constraint_buffer.sampler(bs):
            acc += self.safety_layer.update(batch)
            k += 1
        self.constraint_buffer.pos = self.constraint_buffer.size = 0
        return (acc / max(k, 1)).tolist()

    @torch.no_grad()
    def eval_constraint_models(self, num_steps, batch_size=None):
        """safe_ppo.py:451-466: mean per-constraint loss on freshly collected random-action data, statistics frozen, no update."""
        frozen = self.obs_normalizer.read_only
        self.obs_normalizer.set_read_only()
        self.collect_constraint_data(num_steps)
        self.obs_normalizer.read_only = frozen
        bs = min(int(batch_size or self.constraint_batch_size), self.constraint_buffer.size)
        acc, k = torch.zeros(self.C, device=self.device), 0
        for batch in self.constraint_buffer.sampler(bs):
            acc += self.safety_layer.compute_loss(batch)
            k += 1
        self.constraint_buffer.pos = self.constraint_buffer.size = 0
        return (acc / max(k, 1)).tolist()

    @torch.no_grad()
    def evaluate(self, env, episodes_per_env=1):
        """SafeExplorerPPO.run (safe_ppo.py:230-279) batched: the deterministic, safety-filtered policy on every env of `env` with
        the constraint values of the current state as its second input; per-env totals of the first `episodes_per_env` episodes."""
        N, dev = env.num_envs, env.device
        acc = {k: torch.zeros(N, device=dev) for k in ('count', 'ret', 'length', 'viol', 'mse')}
        nz = self.obs_normalizer
        frozen = nz.read_only
        nz.set_read_only()
        obs = nz(env.reset_tensors())
        c = self._reset_c(env.out, env)
        for _ in range(env.spec.max_episode_steps * episodes_per_env):
            out = env.step_tensors(self.agent.ac.act(obs, c))
            df = (out.done.bool() & (acc['count'] < episodes_per_env)).to(torch.float32)
            acc['ret'] += out.fin_return * df
            acc['length'] += out.fin_length * df
            acc['viol'] += out.fin_violation * df
            acc['mse'] += out.fin_mse * df
            acc['count'] += df
            c, _ = self._next_c(out, env)
            obs = nz(out.obs)
        nz.read_only = frozen
        return acc

    # ---- checkpoints with upstream's keys (safe_ppo.py:146-176): agent, safety_layer, normalisers
    def checkpoint_state(self, training=True):
        state = super().checkpoint_state(training)
        state['safety_layer'] = self.safety_layer.state_dict()
        if training:
            state['c'] = self.c.cpu()
            # upstream saves `total_steps`, which in the pre-training phase IS the epoch count (safe_ppo.py:146-160, :178-213);
            # here total_steps counts env steps of the PPO phase, so the pre-training progress travels under its own key
            state['pretrain_steps'] = int(getattr(self, 'pretrain_steps', 0))
        return state

    def load(self, path, training=True):
        state = torch.load(path, map_location='cpu', weights_only=False)
        super().load(path, training)
        self.load_safety_layer(state)
        if training and 'c' in state:
            self.c = state['c'].to(self.device)
        if training:
            self.pretrain_steps = int(state.get('pretrain_steps', 0))

    def load_safety_layer(self, state_or_path):
        """The `pretrained` hand-over of the second phase (safe_ppo.py:96-100): only the safety layer of a checkpoint."""
        import os
        if isinstance(state_or_path, (str, os.PathLike)):
            p = os.fspath(state_or_path)
            if os.path.isdir(p):
                p = os.path.join(p, 'model_latest.pt')
            state_or_path = torch.load(p, map_location='cpu', weights_only=False)
        sd = state_or_path['safety_layer']
        self.safety_layer.constraint_models.load_state_dict(sd['constraint_models'])
        for o, st in zip(self.safety_layer.optimizers, sd.get('optimizers', [])):
            o.load_state_dict(st)

    def pretrain(self, num_steps, epochs=5):
        self.collect_constraint_data(num_steps)
        hist = []
        for _ in range(epochs):
            acc, k = torch.zeros(self.C, device=self.device), 0
            for batch in self.constraint_buffer.sampler(min(self.constraint_batch_size, self.constraint_buffer.size)):
                acc += self.safety_layer.update(batch)
                k += 1
            hist.append((acc / max(k, 1)).tolist())
        return hist

    # ---- rollout with the constraint values as policy input (safe_ppo.py:299-345)
    def _collect_body(self):
        ac, env = self.agent.ac, self.env
        c = self.c
        for t in range(self.T):
            self.c_buf[t] = c
            act, v, logp = ac.step(self.obs[t], c)
            self.act[t], self.v[t], self.logp[t] = act, v, logp
            out, c_out = self._slots[t]
            env.step_tensors(self.act[t], out=out, c_out=c_out)
            if self._normalise:
                self.obs[t + 1].copy_(self.obs_normalizer(self.obs[t + 1]))
                self.rew[t].copy_(self.reward_normalizer(self.rew[t], self.done[t]))
            c, _ = self._next_c(out)
            d = self.done[t].to(torch.float32)
            self.ep_count += d.sum()
            self.ep_return_sum += (out.fin_return * d).sum()
            self.ep_length_sum += (out.fin_length * d).sum()
            self.ep_violation_sum += (out.fin_violation * d).sum()
        self.c = c

    def train_step(self):
        t0 = time.perf_counter()
        self.collect()
        ret, adv, moments = self._returns_body(dense=False)
        with torch.no_grad():
            parallel.all_reduce_sum_(moments)
            mean = moments[0] / moments[2]
            std = torch.sqrt(torch.clamp(moments[1] / moments[2] - mean * mean, min=0.0))
            adv = (adv - mean) / (std + 1e-6)
        M = self.T * self.N
        data = {'obs': self.obs[:self.T].reshape(M, self.obs_dim), 'act': self.act.reshape(M, self.act_dim),
                'logp': self.logp.reshape(M), 'adv': adv.reshape(M), 'ret': ret.reshape(M), 'v': self.v.reshape(M),
                'c': self.c_buf.reshape(M, self.C)}
        res = self.agent.update(data)
        self.obs[0].copy_(self.obs[self.T])
        res.update({'step': self.total_steps, 'elapsed_time': time.perf_counter() - t0})
        return res

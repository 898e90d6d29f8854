"""SAC on the HIP rollout engine.

Mirrors the reference's SAC (paths relative to /root/reference/safe_control_gym/controllers/sac):
  sac.py:269-335        train_step: one vectorised env step per call (uniform actions during warm-up), time-limit fix-up of
                        next_obs / mask (a truncated transition bootstraps from the TERMINAL observation with mask 1),
                        `train_interval` gradient updates every `train_interval` collected transitions
  sac_utils.py:110-170  losses: policy  (alpha * logp - min(q1,q2)).mean(),  twin-Q regression on
                        rew + gamma * mask * (min target-q - alpha * next_logp),  optional entropy tuning, Polyak update
  sac_utils.py:178-252  tanh-Gaussian actor (log_std clamp [-20, 2], logp -= 2 (log 2 - u - softplus(-2u))), rescaling
                        to the action space, twin MLPQFunction on [obs, act]
  sac_utils.py:301-413  SACBuffer: ring buffer (obs, act, rew, next_obs, mask), uniform sampling
  sac_utils.py:421-424  soft_update

MI355X-first differences: the replay ring lives in HBM (288 GB: 10^7 transitions of the 3-D quadrotor are 2.1 GB), the
step kernel's outputs are copied into it on device, sampling is `torch.randint` on device, and gradients travel in one
flat RCCL all-reduce per update when several ranks train (env + replay shards per rank).
"""
import math
import time
from copy import deepcopy
from dataclasses import dataclass, field

import torch
import torch.nn as nn
import torch.nn.functional as F

from safe_control_gym_amd import parallel
from safe_control_gym_amd.ppo import MLP

LOG2 = math.log(2.0)
LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


class MLPActor(nn.Module):
    def __init__(self, obs_dim, act_dim, hidden_dims, activation, low, high):
        super().__init__()
        self.net = MLP(obs_dim, hidden_dims[-1], hidden_dims[:-1], activation)
        self.mu_layer = nn.Linear(hidden_dims[-1], act_dim)
        self.log_std_layer = nn.Linear(hidden_dims[-1], act_dim)
        self.register_buffer('low', torch.as_tensor(low, dtype=torch.float32))
        self.register_buffer('high', torch.as_tensor(high, dtype=torch.float32))
        self.log_std_min, self.log_std_max = -20, 2

    def forward(self, obs, deterministic=False, with_logprob=True):
        h = self.net(obs)      # NB upstream MLP applies no activation after its last layer (neural_networks.py:50-54)
        mu = self.mu_layer(h)
        log_std = torch.clamp(self.log_std_layer(h), self.log_std_min, self.log_std_max)
        u = mu if deterministic else mu + log_std.exp() * torch.randn_like(mu)
        logp = None
        if with_logprob:
            z = (u - mu) * torch.exp(-log_std)
            logp = (-0.5 * z * z - log_std - LOG_SQRT_2PI).sum(-1, keepdim=True)
            logp = logp - (2 * (LOG2 - u - F.softplus(-2 * u))).sum(dim=1, keepdim=True)
        a = torch.tanh(u)
        return self.low + 0.5 * (a + 1.0) * (self.high - self.low), logp


class MLPQFunction(nn.Module):
    def __init__(self, obs_dim, act_dim, hidden_dims, activation):
        super().__init__()
        self.q_net = MLP(obs_dim + act_dim, 1, hidden_dims, activation)

    def forward(self, obs, act):
        return self.q_net(torch.cat([obs, act], dim=-1))


class MLPActorCritic(nn.Module):
    """state_dict layout of the reference (actor.net.fcs.*, actor.mu_layer, actor.log_std_layer, q1.q_net.fcs.*, q2...)."""

    def __init__(self, obs_dim, act_dim, low, high, hidden_dims=(64, 64), activation='relu'):
        super().__init__()
        self.actor = MLPActor(obs_dim, act_dim, list(hidden_dims), activation, low, high)
        self.q1 = MLPQFunction(obs_dim, act_dim, list(hidden_dims), activation)
        self.q2 = MLPQFunction(obs_dim, act_dim, list(hidden_dims), activation)

    @torch.no_grad()
    def act(self, obs, deterministic=False):
        return self.actor(obs, deterministic, False)[0]


@dataclass
class SACConfig:
    # names and defaults of controllers/sac/sac.yaml
    hidden_dim: int = 256
    activation: str = 'relu'
    gamma: float = 0.99
    tau: float = 0.005
    init_temperature: float = 0.2
    use_entropy_tuning: bool = False
    target_entropy: float = None
    train_interval: int = 100
    train_batch_size: int = 64
    actor_lr: float = 0.001
    critic_lr: float = 0.001
    entropy_lr: float = 0.001
    max_env_steps: int = 1000000
    warm_up_steps: int = 1000
    rollout_batch_size: int = 4
    max_buffer_size: int = 1000000
    extra: dict = field(default_factory=dict)

    @classmethod
    def from_dict(cls, d):
        known = {k: v for k, v in d.items() if k in cls.__dataclass_fields__}
        return cls(**known, extra={k: v for k, v in d.items() if k not in cls.__dataclass_fields__})


class DeviceReplay:
    """SACBuffer (sac_utils.py:301-413) as device tensors; a push appends a whole vectorised step.  The write position lives on the
    device as well (`pos_t`), so a push is a fixed sequence of static-shape device operations — capturable in a HIP graph together with
    the policy forward and the env step (SAC._collect_graph); `pos` / `size` are the host's mirror of it, advanced arithmetically."""

    def __init__(self, capacity, obs_dim, act_dim, device):
        f = dict(device=device, dtype=torch.float32)
        self.capacity = int(capacity)
        self.obs = torch.zeros(self.capacity, obs_dim, **f)
        self.next_obs = torch.zeros(self.capacity, obs_dim, **f)
        self.act = torch.zeros(self.capacity, act_dim, **f)
        self.rew = torch.zeros(self.capacity, 1, **f)
        self.mask = torch.ones(self.capacity, 1, **f)
        self.pos, self.size = 0, 0
        self.pos_t = torch.zeros((), dtype=torch.int64, device=device)          # `pos` on the device
        self.size_t = torch.zeros((), device=device)         # `size` as a device scalar: sampling inside a captured graph
        self.size_i32 = torch.zeros(1, dtype=torch.int32, device=device)       # (the fused update samples in its own kernel)
        self._ar = None

    def push_device(self, obs, act, rew, next_obs, mask):
        """The device side of a push: rows pos .. pos + n - 1 (mod capacity) of the ring, then pos / size advance — no host value."""
        n = obs.shape[0]
        if n > self.capacity:
            raise ValueError('replay capacity smaller than one vectorised step')
        if self._ar is None or self._ar.shape[0] != n:
            self._ar = torch.arange(n, device=self.obs.device)
        idx = (self._ar + self.pos_t) % self.capacity
        self.obs.index_copy_(0, idx, obs)
        self.act.index_copy_(0, idx, act)
        self.next_obs.index_copy_(0, idx, next_obs)
        self.rew.index_copy_(0, idx, rew.reshape(n, 1))
        self.mask.index_copy_(0, idx, mask.reshape(n, 1))
        self.pos_t.add_(n).remainder_(self.capacity)
        self.size_t.add_(float(n)).clamp_(max=float(self.capacity))
        self.size_i32.add_(n).clamp_(max=self.capacity)

    def advance_host(self, n):
        self.pos = (self.pos + n) % self.capacity
        self.size = min(self.size + n, self.capacity)

    def push(self, obs, act, rew, next_obs, mask):
        self.push_device(obs, act, rew, next_obs, mask)
        self.advance_host(obs.shape[0])

    def state_dict(self):
        """SACBuffer.state_dict (sac_utils.py:330-338): the filled part of the ring + the write position."""
        n = self.size
        return {'obs': self.obs[:n].cpu(), 'act': self.act[:n].cpu(), 'rew': self.rew[:n].cpu(), 'next_obs': self.next_obs[:n].cpu(),
                'mask': self.mask[:n].cpu(), 'pos': self.pos, 'size': self.size}

    def load_state_dict(self, sd):
        n = int(sd['size'])
        if n > self.capacity:
            raise ValueError(f'checkpointed replay holds {n} transitions, capacity is {self.capacity}')
        for k in ('obs', 'act', 'rew', 'next_obs', 'mask'):
            getattr(self, k)[:n].copy_(sd[k].to(self.obs.device))
        self.pos, self.size = int(sd['pos']) % self.capacity, n
        self.pos_t.fill_(self.pos)
        self.size_t.fill_(float(n))
        self.size_i32.fill_(n)

    def sample_static(self, batch_size):
        """Uniform sample with static shapes and no host value: usable under HIP-graph capture."""
        idx = (torch.rand(batch_size, device=self.obs.device) * self.size_t).long().clamp_(min=0)
        idx = torch.minimum(idx, (self.size_t - 1).clamp(min=0).long())
        return {'obs': self.obs[idx], 'act': self.act[idx], 'rew': self.rew[idx], 'next_obs': self.next_obs[idx],
                'mask': self.mask[idx]}

    def sample(self, batch_size, generator=None):
        idx = torch.randint(0, self.size, (batch_size,), device=self.obs.device, generator=generator)
        return {'obs': self.obs[idx], 'act': self.act[idx], 'rew': self.rew[idx], 'next_obs': self.next_obs[idx],
                'mask': self.mask[idx]}


class SACAgent:
    def __init__(self, obs_dim, act_dim, low, high, cfg: SACConfig, device):
        self.cfg = cfg
        self.ac = MLPActorCritic(obs_dim, act_dim, low, high, [cfg.hidden_dim] * 2, cfg.activation).to(device)
        parallel.broadcast_parameters([self.ac])
        self.ac_targ = deepcopy(self.ac)
        for p in self.ac_targ.parameters():
            p.requires_grad = False
        self.log_alpha = torch.tensor(math.log(cfg.init_temperature), device=device, requires_grad=cfg.use_entropy_tuning)
        self.target_entropy = -float(act_dim) if cfg.target_entropy is None else cfg.target_entropy
        # One HIP graph per gradient step (sample + three Adam steps + Polyak) on a single GPU: the update is ~100 tiny
        # kernels on 128-wide MLPs, i.e. launch-bound.  With several ranks the all-reduces sit between backward and step,
        # and the update stays eager.
        self.use_graphs = (torch.device(device).type == 'cuda' and bool(cfg.extra.get('cuda_graphs', True))
                           and parallel.world_size() == 1)
        # Fused gradient step (csrc/scg_sac.hip, include/scg_sac.h): the whole SACAgent.update — sampling, the five network
        # passes on the matrix cores, both Adam steps, temperature, Polyak — as 9 launches on flat parameter vectors instead
        # of ~100 PyTorch kernels.  Chosen here, visibly: single-rank GPU runs of shapes the library serves.
        from safe_control_gym_amd import _sac
        self.obs_dim, self.act_dim = obs_dim, act_dim
        self.use_fused = (torch.device(device).type == 'cuda' and bool(cfg.extra.get('cuda_graphs', True))
                          and bool(cfg.extra.get('fused_update', True))
                          and _sac.supported(obs_dim, cfg.hidden_dim, act_dim, cfg.activation))
        self._flat = self._flatten(low, high) if self.use_fused else None
        self._fused = None
        self.dp_path, self.dp_capture_error = None, None       # how the last data-parallel update ran (_update_fused)
        if self.use_fused:          # one-time kernel attributes NOW (not a stream operation: must not fall into a later graph capture)
            with torch.cuda.device(self.log_alpha.device):
                D = _sac.lib(obs_dim, cfg.hidden_dim, act_dim, cfg.activation)
                _sac.check(D, D.scg_sac_prepare())
        kw = {'capturable': True} if self.use_graphs else {}
        self.actor_opt = torch.optim.Adam(self.ac.actor.parameters(), cfg.actor_lr, **kw)
        self.critic_opt = torch.optim.Adam(list(self.ac.q1.parameters()) + list(self.ac.q2.parameters()), cfg.critic_lr, **kw)
        self.alpha_opt = torch.optim.Adam([self.log_alpha], cfg.entropy_lr, **kw)
        self._ab = self._cb = None
        self._graph = None

    @property
    def alpha(self):
        return self.log_alpha.exp()

    # ---- fused update ------------------------------------------------------------------------------------------
    def _flatten(self, low, high):
        """All trainable tensors as views of ONE flat vector [actor | q1 | q2 | log_alpha] (+ a target vector in the same
        order, gradient scratch, Adam moments); the actor's two heads are stored as one [2 act_dim][H] matrix / [2 act_dim]
        bias (scg_sac.h).  The torch modules keep working on the views (acting, evaluation, checkpoints)."""
        from safe_control_gym_amd._learn import MlpLayout
        a, dev = self.ac.actor, self.log_alpha.device

        def order(ac):
            act = ac.actor
            return ([act.net.fcs[0].weight, act.net.fcs[0].bias, act.net.fcs[1].weight, act.net.fcs[1].bias, act.mu_layer.weight,
                     act.log_std_layer.weight, act.mu_layer.bias, act.log_std_layer.bias],
                    [p for f in ac.q1.q_net.fcs for p in (f.weight, f.bias)], [p for f in ac.q2.q_net.fcs for p in (f.weight, f.bias)])
        if len(a.net.fcs) != 2 or len(self.ac.q1.q_net.fcs) != 3:
            raise ValueError('the fused SAC update serves two hidden layers')
        groups = order(self.ac)
        params = [p for g in groups for p in g]
        n = sum(p.numel() for p in params)
        flat = torch.cat([p.data.reshape(-1) for p in params] + [self.log_alpha.data.reshape(1)]).contiguous()
        offs, off = [], 0
        for p in params:
            offs.append(off)
            p.data = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.log_alpha.data = flat[n:n + 1].view(())
        targ = torch.empty(n, device=dev)
        off = 0
        for p in [q for g in order(self.ac_targ) for q in g]:
            targ[off:off + p.numel()].copy_(p.data.reshape(-1))
            p.data = targ[off:off + p.numel()].view_as(p)
            off += p.numel()
        lay = lambda k: MlpLayout(offs[k], offs[k + 1], offs[k + 2], offs[k + 3], offs[k + 4], offs[k + 5])      # noqa: E731
        actor_lay = MlpLayout(offs[0], offs[1], offs[2], offs[3], offs[4], offs[6])       # W3 = [mu; log_std] rows, b3 = [mu; log_std]
        n_actor = sum(p.numel() for p in groups[0])
        return {'p': flat, 'targ': targ, 'g': torch.zeros(n + 1, device=dev), 'm': torch.zeros(n + 1, device=dev),
                'v': torch.zeros(n + 1, device=dev), 'steps': torch.zeros(3, device=dev), 'n': n, 'n_actor': n_actor,
                'actor': actor_lay, 'q1': lay(8), 'q2': lay(14), 'low': [float(x) for x in low.reshape(-1)],
                'high': [float(x) for x in high.reshape(-1)], 'counter': torch.zeros(1, dtype=torch.int32, device=dev),
                'seed': int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF}

    def _fused_args(self, buffer, batch_size, idx=None, eps=None, eps_next=None):
        import ctypes as C
        from safe_control_gym_amd import _sac
        fl, cfg = self._flat, self.cfg
        D = _sac.lib(self.obs_dim, cfg.hidden_dim, self.act_dim, cfg.activation)
        dev = fl['p'].device
        with torch.cuda.device(dev):            # kernel attributes are per device (csrc/scg_once.h)
            _sac.check(D, D.scg_sac_prepare())
        ws = torch.empty(D.scg_sac_workspace_bytes(int(batch_size)), dtype=torch.uint8, device=dev)
        stats, acc = torch.zeros(4, device=dev), torch.zeros(4, device=dev)
        p = lambda t: t.data_ptr() if t is not None else None          # noqa: E731
        a = _sac.SacArgs(d_params=p(fl['p']), d_target=p(fl['targ']), d_grad=p(fl['g']), d_m=p(fl['m']), d_v=p(fl['v']), d_steps=p(fl['steps']),
                         actor=fl['actor'], q1=fl['q1'], q2=fl['q2'], n_actor=fl['n_actor'], n_params=fl['n'], d_obs=p(buffer.obs),
                         d_act=p(buffer.act), d_rew=p(buffer.rew), d_next_obs=p(buffer.next_obs), d_mask=p(buffer.mask),
                         d_ring_size=p(buffer.size_i32), batch=int(batch_size), gamma=float(cfg.gamma), tau=float(cfg.tau),
                         actor_lr=float(cfg.actor_lr), critic_lr=float(cfg.critic_lr), entropy_lr=float(cfg.entropy_lr),
                         use_entropy_tuning=int(bool(cfg.use_entropy_tuning)), target_entropy=float(self.target_entropy),
                         seed=fl['seed'], d_counter=p(fl['counter']), d_idx_in=p(idx), d_eps_in=p(eps), d_eps_next_in=p(eps_next),
                         d_workspace=p(ws), d_stats=p(stats), d_stats_acc=p(acc))
        for j in range(self.act_dim):
            a.act_low[j], a.act_high[j] = fl['low'][j], fl['high'][j]
        return {'D': D, 'args': a, 'ws': ws, 'stats': stats, 'acc': acc, 'C': C, 'keep': (idx, eps, eps_next, buffer)}

    def _fused_step(self, F, phases=0):
        """Enqueue ONE gradient step (or the given part of it, scg_sac.h: SCG_SAC_*) on the current stream."""
        from safe_control_gym_amd import _sac
        st = F['C'].c_void_p(torch.cuda.current_stream(self._flat['p'].device).cuda_stream)
        F['args'].phases = int(phases)
        _sac.check(F['D'], F['D'].scg_sac_update(F['C'].byref(F['args']), st))

    def _fused_step_dp(self, F):
        """One data-parallel gradient step: every rank samples its own minibatch; the actor's (+ temperature's) gradient and
        the critics' gradient are averaged over the ranks where sac_utils.py's update would call backward() — two all-reduces of
        the flat gradient vector per step (RCCL; latency-bound at 246 KB), everything else stays the fused kernels."""
        from safe_control_gym_amd import _sac
        g, world = self._flat['g'], max(parallel.world_size(), 1)
        self._fused_step(F, _sac.ACTOR_GRAD)
        parallel.all_reduce_sum_(g)
        g.div_(world)
        self._fused_step(F, _sac.CRITIC_GRAD)
        parallel.all_reduce_sum_(g)
        g.div_(world)
        self._fused_step(F, _sac.FINISH)

    def _update_fused(self, buffer, batch_size, n_updates, lazy=False):
        """lazy=True: enqueue only — the loss sums stay on the device ('stats_dev': policy, critic, entropy loss summed over the
        n_updates steps; valid until the next update's kernels run), nothing is read back and the call does not wait for the GPU."""
        if batch_size % 32:
            raise ValueError('the fused SAC update needs train_batch_size to be a multiple of 32')
        c = self.cfg                # (every value the argument block — and with it the captured graphs — carries BY VALUE: a changed cfg rebuilds)
        key = (id(buffer), batch_size, float(c.gamma), float(c.tau), float(c.actor_lr), float(c.critic_lr), float(c.entropy_lr),
               bool(c.use_entropy_tuning), float(c.target_entropy) if getattr(c, 'target_entropy', None) is not None else None)
        if self._fused is None or self._fused['key'] != key:
            self._fused = dict(self._fused_args(buffer, batch_size), key=key, graphs={})
        F = self._fused
        F['acc'].zero_()
        dev = self._flat['p'].device
        if parallel.world_size() > 1 or self.cfg.extra.get('force_data_parallel'):
            # Collectives between the parts of a step.  Over RCCL the n_updates steps (two all-reduces each) are ONE HIP-graph replay —
            # the first call runs eagerly (it warms the communicator up), the second captures; over gloo, or if the capture fails, the
            # steps are enqueued one by one (`dp_path` says which ran).
            with torch.cuda.device(dev):
                key = ('dp', n_updates)
                state = F['graphs'].get(key)
                if parallel.collectives_capturable() and self.cfg.extra.get('graph_collectives', True) and state != 'eager':
                    if state is None:
                        F['graphs'][key] = 'warm'
                    elif state == 'warm':
                        try:
                            torch.cuda.synchronize(dev)
                            g = torch.cuda.CUDAGraph()
                            # (thread_local: ProcessGroupNCCL's watchdog thread must not invalidate this thread's capture)
                            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                                for _ in range(n_updates):
                                    self._fused_step_dp(F)
                            F['graphs'][key] = state = g
                        except Exception as exc:                    # noqa: BLE001
                            F['graphs'][key] = state = 'eager'
                            self.dp_capture_error = repr(exc)[:200]
                            import warnings
                            warnings.warn(f'SAC data-parallel update: HIP-graph capture failed, staying on the eager loop ({self.dp_capture_error})')
                    if isinstance(state, torch.cuda.CUDAGraph):
                        state.replay()
                        self.dp_path = f'one graph replay per vector step ({n_updates} gradient steps, {2 * n_updates} all-reduces)'
                        return self._fused_stats(F, n_updates, lazy)
                self.dp_path = 'eager'
                for _ in range(n_updates):
                    self._fused_step_dp(F)
            return self._fused_stats(F, n_updates, lazy)
        g = F['graphs'].get(n_updates)
        if g is None:                               # n_updates steps as one HIP graph (7 launches each + 1: host-launch bound otherwise)
            from safe_control_gym_amd import _sac
            with torch.cuda.device(dev):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    # scg_sac_update_n: step k + 1's first launch rides in step k's target-action launch (bit-identical to n calls)
                    F['args'].phases = 0
                    st = F['C'].c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                    _sac.check(F['D'], F['D'].scg_sac_update_n(F['C'].byref(F['args']), int(n_updates), st))
            F['graphs'][n_updates] = g
        g.replay()
        return self._fused_stats(F, n_updates, lazy)

    @staticmethod
    def _fused_stats(F, n_updates, lazy):
        if lazy:
            return {'stats_dev': F['acc'], 'stats_updates': n_updates}
        st = (F['acc'] / n_updates).tolist()
        return {'policy_loss': st[0], 'critic_loss': st[1], 'entropy_loss': st[2]}

    # ---- the two Adam layouts: torch.optim per-parameter state <-> the fused update's flat m / v / steps
    def _opt_groups(self):
        return ((self.actor_opt, 0), (self.critic_opt, 1), (self.alpha_opt, 2))

    def _flat_offset(self, p):
        fl = self._flat['p']
        return (p.data_ptr() - fl.data_ptr()) // fl.element_size()

    def _adam_flat_to_torch(self):
        """Write the flat moments into the torch optimisers' state (what their state_dict() then carries): a checkpoint of a fused
        agent resumes in an agent that steps torch.optim (CPU, unsupported shape, fused_update off) and in the reference's SACAgent."""
        fl = self._flat
        steps = fl['steps'].tolist()
        for opt, which in self._opt_groups():
            if steps[which] <= 0:
                continue
            for p in opt.param_groups[0]['params']:
                o, k = self._flat_offset(p), p.numel()
                st = opt.state[p]
                st['exp_avg'] = fl['m'][o:o + k].view_as(p).clone()
                st['exp_avg_sq'] = fl['v'][o:o + k].view_as(p).clone()
                st['step'] = torch.tensor(float(steps[which]), device=p.device if opt.defaults.get('capturable') else 'cpu')

    def _adam_torch_to_flat(self):
        """The converse: per-parameter torch.optim state (the reference's training checkpoints, or one saved with fused_update off)
        scattered into the flat moments the fused kernels step."""
        fl = self._flat
        for opt, which in self._opt_groups():
            n_steps = 0.0
            for p in opt.param_groups[0]['params']:
                st = opt.state.get(p)
                o, k = self._flat_offset(p), p.numel()
                if not st:                      # no state for this parameter (zero-step checkpoint): a load is a full overwrite
                    fl['m'][o:o + k].zero_()
                    fl['v'][o:o + k].zero_()
                    continue
                fl['m'][o:o + k].copy_(st['exp_avg'].reshape(-1))
                fl['v'][o:o + k].copy_(st['exp_avg_sq'].reshape(-1))
                n_steps = float(st['step'])
            fl['steps'][which] = n_steps

    # ---- deterministic actions for evaluation
    @torch.no_grad()
    def act_deterministic(self, obs):
        """`ac.act(obs, deterministic=True)` (sac_utils.py:209-215: tanh of the mean, rescaled to the action space).  On the fused path ONE
        launch of the library's batched actor on the flat parameter vector (scg_sac_act, exact-f32 MFMA) instead of ~9 PyTorch kernels:
        the evaluation loop is 250 sequential (policy, env step) pairs on a 256-env batch, i.e. launch-bound."""
        if not self.use_fused or obs.dtype != torch.float32 or obs.dim() != 2:
            return self.ac.act(obs, deterministic=True)
        import ctypes as C
        from safe_control_gym_amd import _sac
        fl = self._flat
        D = _sac.lib(self.obs_dim, self.cfg.hidden_dim, self.act_dim, self.cfg.activation)
        if '_act_bounds' not in fl:
            fl['_act_bounds'] = ((C.c_float * 4)(*(fl['low'] + [0.0] * (4 - self.act_dim))), (C.c_float * 4)(*(fl['high'] + [0.0] * (4 - self.act_dim))))
        lo, hi = fl['_act_bounds']
        x = obs.contiguous()
        out = torch.empty(x.shape[0], self.act_dim, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _sac.check(D, D.scg_sac_act(fl['p'].data_ptr(), C.byref(fl['actor']), lo, hi, x.data_ptr(), x.shape[0], out.data_ptr(),
                                        C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        return out

    def deterministic_policy(self):
        """An object with `.act(obs)` for ppo.evaluate / the controllers (one per agent: evaluate caches its captured graph per policy object)."""
        if getattr(self, '_det_policy', None) is None:
            agent = self

            class _Det:
                ac = agent.ac

                @staticmethod
                def act(obs):
                    return agent.act_deterministic(obs)
            self._det_policy = _Det()
        return self._det_policy

    # same keys as the reference's SACAgent (sac_utils.py:85-108), so its checkpoints load directly; the action bounds are
    # buffers here (not in upstream's state dict), hence strict=False
    def state_dict(self):
        if self._flat is not None:
            self._adam_flat_to_torch()
        sd = {'ac': self.ac.state_dict(), 'log_alpha': self.log_alpha.detach().clone(), 'ac_targ': self.ac_targ.state_dict(),
              'actor_opt': self.actor_opt.state_dict(), 'critic_opt': self.critic_opt.state_dict(),
              'alpha_opt': self.alpha_opt.state_dict()}
        if self._flat is not None:          # fused mode: the sampling counter of the update kernels (the moments travel in torch's layout above)
            sd['flat_adam'] = {'counter': self._flat['counter'].clone()}
        return sd

    def load_state_dict(self, sd, with_optimizers=True):
        self.ac.load_state_dict(sd['ac'], strict=False)
        if 'ac_targ' in sd:
            self.ac_targ.load_state_dict(sd['ac_targ'], strict=False)
        if 'log_alpha' in sd:
            with torch.no_grad():
                self.log_alpha.copy_(torch.as_tensor(sd['log_alpha'], dtype=self.log_alpha.dtype).reshape(()))
        if with_optimizers and 'actor_opt' in sd:
            # (also with HIP graphs on: capturable Adam keeps its state in device tensors, which load_state_dict replaces —
            #  the captured update graph aliases the old ones and is re-captured on the next update)
            self.actor_opt.load_state_dict(sd['actor_opt'])
            self.critic_opt.load_state_dict(sd['critic_opt'])
            self.alpha_opt.load_state_dict(sd['alpha_opt'])
            for opt, _ in self._opt_groups():               # (a checkpoint mapped onto the GPU: eager Adam keeps its step counts on the host)
                if not opt.defaults.get('capturable'):
                    for st in opt.state.values():
                        if torch.is_tensor(st.get('step')):
                            st['step'] = st['step'].cpu()
            self._graph = None
            if self._flat is not None:      # in place: captured graphs alias the flat buffers
                self._adam_torch_to_flat()
                for k, t in sd.get('flat_adam', {}).items():        # (round-3 checkpoints carry m / v / steps here as well: same values)
                    self._flat[k].copy_(t.to(self._flat[k].device))

    def policy_loss(self, batch):
        obs = batch['obs']
        act, logp = self.ac.actor(obs)
        q = torch.min(self.ac.q1(obs, act), self.ac.q2(obs, act))
        policy_loss = (self.alpha.detach() * logp - q).mean()
        entropy_loss = torch.zeros((), device=obs.device)
        if self.cfg.use_entropy_tuning:
            entropy_loss = -(self.log_alpha * (logp + self.target_entropy).detach()).mean()
        return policy_loss, entropy_loss

    def q_loss(self, batch):
        obs, act, rew, next_obs, mask = batch['obs'], batch['act'], batch['rew'], batch['next_obs'], batch['mask']
        q1, q2 = self.ac.q1(obs, act), self.ac.q2(obs, act)
        with torch.no_grad():
            next_act, next_logp = self.ac.actor(next_obs)
            nq = torch.min(self.ac_targ.q1(next_obs, next_act), self.ac_targ.q2(next_obs, next_act))
            q_targ = rew + self.cfg.gamma * mask * (nq - self.alpha * next_logp)
        return (q1 - q_targ).pow(2).mean() + (q2 - q_targ).pow(2).mean()

    def _reduce(self, params, attr):
        if parallel.world_size() > 1:
            b = getattr(self, attr)
            if b is None:
                b = parallel.FlatBucket(params)
                setattr(self, attr, b)
            b.pack()
            b.all_reduce_mean()
            b.unpack()

    def update(self, batch):
        """sac_utils.py:143-170: actor step, optional temperature step, critic step, Polyak update."""
        policy_loss, entropy_loss = self.policy_loss(batch)
        self.actor_opt.zero_grad(set_to_none=False)
        policy_loss.backward()
        self._reduce(list(self.ac.actor.parameters()), '_ab')
        self.actor_opt.step()
        if self.cfg.use_entropy_tuning:
            self.alpha_opt.zero_grad()
            entropy_loss.backward()
            if parallel.world_size() > 1:
                parallel.all_reduce_sum_(self.log_alpha.grad).div_(parallel.world_size())
            self.alpha_opt.step()
        critic_loss = self.q_loss(batch)
        self.critic_opt.zero_grad(set_to_none=False)
        critic_loss.backward()
        self._reduce(list(self.ac.q1.parameters()) + list(self.ac.q2.parameters()), '_cb')
        self.critic_opt.step()
        with torch.no_grad():       # soft_update (sac_utils.py:421-424)
            for p, pt in zip(self.ac.parameters(), self.ac_targ.parameters()):
                pt.mul_(1.0 - self.cfg.tau).add_(p, alpha=self.cfg.tau)
        return {'policy_loss': policy_loss.detach(), 'critic_loss': critic_loss.detach(), 'entropy_loss': entropy_loss.detach()}


    def update_from_buffer(self, buffer, batch_size, n_updates, lazy=False):
        """n_updates gradient steps on fresh uniform samples; replays one captured graph per step when enabled.
        lazy (fused path): see _update_fused."""
        if self.use_fused:
            return self._update_fused(buffer, batch_size, n_updates, lazy=lazy)
        if not self.use_graphs:
            acc = None
            for _ in range(n_updates):
                res = self.update(buffer.sample(batch_size))
                acc = res if acc is None else {k: acc[k] + v for k, v in res.items()}
            return {k: float(v) / n_updates for k, v in acc.items()}
        if self._graph is None or self._graph['key'] != (id(buffer), batch_size):
            dev = buffer.obs.device
            stats = torch.zeros(3, device=dev)

            def one():
                res = self.update(buffer.sample_static(batch_size))
                stats.add_(torch.stack([res['policy_loss'], res['critic_loss'], res['entropy_loss']]))

            s = torch.cuda.Stream(dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                for _ in range(3):                      # warm-up = three real gradient steps
                    one()
            torch.cuda.current_stream(dev).wait_stream(s)
            for opt in (self.actor_opt, self.critic_opt, self.alpha_opt):
                opt.zero_grad(set_to_none=True)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                one()
            self._graph = {'key': (id(buffer), batch_size), 'g': g, 'stats': stats}
        G = self._graph
        G['stats'].zero_()
        for _ in range(n_updates):
            G['g'].replay()
        st = (G['stats'] / n_updates).tolist()
        return {'policy_loss': st[0], 'critic_loss': st[1], 'entropy_loss': st[2]}


class SAC:
    """SAC.train_step / learn on a HipVecEnv (sac.py:162-335)."""

    def __init__(self, env, cfg: SACConfig, seed=0):
        self.env, self.cfg = env, cfg
        self.device = env.device
        if env.dtype != torch.float32:
            raise ValueError('the SAC collector runs on float32 environments')
        # sac.yaml:6-9 / sac.py:75-81: running normalisers around env.step (defaults off), device-resident (normalization.py)
        from safe_control_gym_amd.normalization import BaseNormalizer, MeanStdNormalizer, RewardStdNormalizer
        x = cfg.extra
        self.obs_normalizer = (MeanStdNormalizer((env.spec.obs_dim,), self.device, clip=x.get('clip_obs', 10.0)) if x.get('norm_obs')
                               else BaseNormalizer())
        self.reward_normalizer = (RewardStdNormalizer(cfg.gamma, self.device, clip=x.get('clip_reward', 10.0)) if x.get('norm_reward')
                                  else BaseNormalizer())
        self._normalise = bool(x.get('norm_obs') or x.get('norm_reward'))
        if x.get('norm_reward'):        # eager: graphs captured later (and checkpoint loads, in place) share this storage
            self.reward_normalizer.ret = torch.zeros(env.num_envs, dtype=torch.float64, device=self.device)
        spec = env.spec
        self.N, self.obs_dim, self.act_dim = env.num_envs, spec.obs_dim, spec.nu
        rank = torch.distributed.get_rank() if parallel.world_size() > 1 else 0
        torch.manual_seed(seed + 7919 * rank)
        self.low = torch.as_tensor(spec.action_space.low, dtype=torch.float32, device=self.device)
        self.high = torch.as_tensor(spec.action_space.high, dtype=torch.float32, device=self.device)
        self.agent = SACAgent(self.obs_dim, self.act_dim, self.low, self.high, cfg, self.device)
        self.buffer = DeviceReplay(cfg.max_buffer_size, self.obs_dim, self.act_dim, self.device)
        self.obs = self.obs_normalizer(env.reset_tensors()).clone()         # sac.py:103-104; persistent storage (captured graphs alias it)
        self.total_steps = 0
        self._since_update = 0
        self._graph_collect = self.device.type == 'cuda' and bool(cfg.extra.get('graph_collect', cfg.extra.get('cuda_graphs', True)))
        self._collect_graphs = {}
        # fused collector (csrc/scg_sac.hip: scg_sac_sample, scg_sac_push): the policy's sampled action in ONE launch on the flat
        # parameter vector (in-kernel Philox noise) and the time-limit fix-up + five ring writes + position bookkeeping in two, in place
        # of ~40 PyTorch kernels per vector step.  Needs the fused agent and no running normalisers; extra['fused_collect'] = False
        # keeps the PyTorch collector (A/B, tests) — chosen here, visibly.
        self._fused_collect = bool(self.agent.use_fused and not self._normalise and cfg.extra.get('fused_collect', True))
        if self._fused_collect:
            self._act = torch.zeros(self.N, self.act_dim, device=self.device)
            self._collect_counter = torch.zeros(1, dtype=torch.int32, device=self.device)       # counter word of the action noise

    # ---- one vectorised env step into the replay ring (sac.py:273-311)
    @torch.no_grad()
    def _collect_body(self, warm):
        """Static-shape device operations only: action (uniform during warm-up, else a sample of the policy), env step kernel,
        time-limit fix-up, normalisers, ring push, the next observation into the persistent `self.obs`."""
        env = self.env
        if self._fused_collect:
            return self._collect_body_fused(warm)
        if warm:                                        # action_space.sample() per env (sac.py:276-277)
            act = self.low + (self.high - self.low) * torch.rand(self.N, self.act_dim, device=self.device)
        else:
            act = self.agent.ac.act(self.obs)
        out = env.step_tensors(act)
        done = out.done.bool()
        trunc = (out.flags & 1).bool() & done
        if self._normalise:
            # sac.py:281-297, in upstream's order: next_obs, then the reward (running returns, its index-array reset), then the
            # terminal observations of the TRUNCATED envs only — each call updates the running statistics
            self.obs_normalizer.unset_read_only()
            obs_n = self.obs_normalizer(out.obs)
            rew = self.reward_normalizer(out.reward, done)
            term_n = self.obs_normalizer(out.terminal_obs, mask=trunc)
        else:
            obs_n, rew, term_n = out.obs, out.reward, out.terminal_obs
        # time truncation is not termination: the stored next state is the terminal observation, mask 1 (sac.py:287-305)
        next_obs = torch.where(trunc[:, None], term_n, obs_n)
        mask = torch.where(trunc, torch.ones_like(out.reward), 1.0 - done.to(torch.float32))
        self.buffer.push_device(self.obs, act, rew, next_obs, mask)
        self.obs.copy_(obs_n)

    def _collect_body_fused(self, warm):
        """The same vector step as three library calls / four launches: scg_sac_sample (uniform warm-up action or a sample of the policy),
        the env step kernel, scg_sac_push (fix-up, ring rows, current observation, position / size / noise counter)."""
        import ctypes as C
        from safe_control_gym_amd import _sac
        ag, buf = self.agent, self.buffer
        fl = ag._flat
        D = _sac.lib(self.obs_dim, self.cfg.hidden_dim, self.act_dim, self.cfg.activation)
        if '_act_bounds' not in fl:
            fl['_act_bounds'] = ((C.c_float * 4)(*(fl['low'] + [0.0] * (4 - self.act_dim))), (C.c_float * 4)(*(fl['high'] + [0.0] * (4 - self.act_dim))))
        lo, hi = fl['_act_bounds']
        p = lambda t: t.data_ptr()                      # noqa: E731
        ring = getattr(self, '_ring', None)
        if ring is None or self._ring_of is not buf:
            ring = self._ring = _sac.SacRing(d_obs=p(buf.obs), d_act=p(buf.act), d_rew=p(buf.rew), d_next_obs=p(buf.next_obs), d_mask=p(buf.mask),
                                             capacity=buf.capacity, d_pos=p(buf.pos_t), d_size_f=p(buf.size_t), d_size_i32=p(buf.size_i32),
                                             d_counter=p(self._collect_counter))
            self._ring_of = buf
        if self.N > buf.capacity:
            raise ValueError('replay capacity smaller than one vectorised step')
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _sac.check(D, D.scg_sac_sample(p(fl['p']), C.byref(fl['actor']), lo, hi, p(self.obs), self.N, fl['seed'], p(self._collect_counter),
                                           int(bool(warm)), None, p(self._act), st))
            out = self.env.step_tensors(self._act)
            _sac.check(D, D.scg_sac_push(C.byref(ring), p(self.obs), p(self._act), p(out.reward), p(out.obs), p(out.terminal_obs), p(out.done),
                                         p(out.flags), self.N, st))

    def _collect(self, warm):
        """Eager, or (GPU, cfg.extra['cuda_graphs'] not off) ONE HIP-graph replay per vector step: the eager collector is ~45 small
        launches per step — policy MLP, sampling, env kernel, fix-up, five ring writes — i.e. host-launch bound (0.7 ms of a 2.7 ms
        vector step at 2048 envs x 16 gradient steps).  One graph per phase (warm-up / policy), captured after two eager steps of
        that phase and re-captured after env.seed() (the Philox key is a kernel argument)."""
        if not self._graph_collect:
            self._collect_body(warm)
            return
        key = (bool(warm), getattr(self.env, 'seed_epoch', 0), id(self.buffer))
        st = self._collect_graphs.get(key)
        if st is None:
            st = self._collect_graphs[key] = {'eager': 0, 'g': None}
        if st['g'] is None:
            if st['eager'] < 2:
                st['eager'] += 1
                self._collect_body(warm)
                return
            torch.cuda.current_stream(self.device).synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._collect_body(warm)
            st['g'] = g
        st['g'].replay()

    def train_step(self, lazy=False):
        """One vectorised env step (+ the gradient steps it owes).  lazy=True: the fused update's statistics are not read back
        ('stats_dev' instead of floats) — with the graph-replayed collector the call then never waits for the GPU."""
        cfg = self.cfg
        t0 = time.perf_counter()
        self._collect(self.total_steps < cfg.warm_up_steps)
        self.buffer.advance_host(self.N)
        world = parallel.world_size()
        self.total_steps += self.N * world
        self._since_update += self.N * world
        results = {}
        if self.total_steps > cfg.warm_up_steps and self._since_update >= cfg.train_interval:
            # the reference locks the ratio of gradient steps to env steps to 1 (sac.py:323-331); with N envs per
            # vectorised step that is N updates per step — `updates_per_step` caps it (documented deviation knob)
            n_updates = int(cfg.extra.get('updates_per_step', self._since_update))
            self._since_update = 0
            results = self.agent.update_from_buffer(self.buffer, cfg.train_batch_size, n_updates, lazy=lazy)
            results['updates'] = n_updates
        results.update({'step': self.total_steps, 'elapsed_time': time.perf_counter() - t0})
        return results

    # ---- checkpoint / resume (sac.py:119-160: same keys)
    def save(self, path, training=True, save_buffer=False):
        import os
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        state = {'agent': self.agent.state_dict(), 'obs_normalizer': self.obs_normalizer.state_dict(),
                 'reward_normalizer': self.reward_normalizer.state_dict()}            # sac.py:124-128
        for name, nz in (('obs', self.obs_normalizer), ('reward', self.reward_normalizer)):
            if hasattr(nz, 'rms'):              # (upstream's state dict drops the count: a resumed run would re-weight new batches)
                state[f'{name}_normalizer_count'] = float(nz.rms.count)
        if getattr(self.reward_normalizer, 'ret', None) is not None:
            state['reward_normalizer_ret'] = self.reward_normalizer.ret.cpu()
        if training:
            if self._fused_collect:             # counter word of the fused collector's action noise (scg_sac_sample)
                state['collect_counter'] = int(self._collect_counter.item())
            state.update({'total_steps': self.total_steps, 'since_update': self._since_update, 'obs': self.obs.cpu(),
                          'random_state': {'torch': torch.get_rng_state(),
                                           'torch_cuda': torch.cuda.get_rng_state(self.device) if self.device.type == 'cuda' else None},
                          'env_random_state': self.env.get_env_random_state()})
            if save_buffer:                     # (upstream's default is off, sac.py:119: the ring is large; on for an exact experiment restore)
                state['buffer'] = self.buffer.state_dict()
        torch.save(state, path)

    def load(self, path, training=True):
        state = torch.load(path, map_location=self.device, weights_only=False)
        self.agent.load_state_dict(state['agent'], with_optimizers=training)
        for name, nz in (('obs', self.obs_normalizer), ('reward', self.reward_normalizer)):         # sac.py:147-149
            if state.get(f'{name}_normalizer') and hasattr(nz, 'rms'):
                nz.load_state_dict(state[f'{name}_normalizer'])
                if f'{name}_normalizer_count' in state:
                    nz.rms.count.fill_(state[f'{name}_normalizer_count'])
        if 'reward_normalizer_ret' in state and hasattr(self.reward_normalizer, 'rms'):
            ret = state['reward_normalizer_ret'].to(self.device)
            if self.reward_normalizer.ret is not None and self.reward_normalizer.ret.shape == ret.shape:
                self.reward_normalizer.ret.copy_(ret)               # in place: the collector's graphs alias this tensor
            else:
                self.reward_normalizer.ret = ret
                self._collect_graphs = {}
        if training and 'total_steps' in state:
            self.total_steps = int(state['total_steps'])
            self._since_update = int(state.get('since_update', 0))
            if 'obs' in state:
                self.obs.copy_(state['obs'].to(self.device))              # in place: the collector's graphs alias this tensor
            if self._fused_collect and 'collect_counter' in state:
                self._collect_counter.fill_(int(state['collect_counter']))
            if 'env_random_state' in state:
                self.env.set_env_random_state(state['env_random_state'])
            rs = state.get('random_state')
            if rs:
                torch.set_rng_state(rs['torch'].cpu())
                if rs.get('torch_cuda') is not None and self.device.type == 'cuda':
                    torch.cuda.set_rng_state(rs['torch_cuda'].cpu(), self.device)
            if 'buffer' in state:
                self.buffer.load_state_dict(state['buffer'])
            elif self.total_steps > self.cfg.warm_up_steps:
                import warnings
                warnings.warn('SAC checkpoint without replay buffer (save_buffer=False): training resumes past warm-up with an '
                              'EMPTY buffer — the first updates sample only the transitions collected since the resume')
        return state

    def learn(self, max_env_steps=None, log=None):
        max_env_steps = max_env_steps or self.cfg.max_env_steps
        hist = []
        while self.total_steps < max_env_steps:
            res = self.train_step()
            hist.append(res)
            if log:
                log(res)
        return hist

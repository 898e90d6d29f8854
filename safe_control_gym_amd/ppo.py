"""PPO on the HIP rollout engine.

Mirrors the reference's PPO (paths relative to /root/reference/safe_control_gym):
  controllers/ppo/ppo.py:259-303        train_step: rollout collector, truncated-episode bootstrap, advantage normalisation
  controllers/ppo/ppo_utils.py:18-146   PPOAgent: two Adam optimisers, clipped surrogate, entropy bonus, approx-KL gate
                                        (actor step skipped when approx_kl > 1.5 target_kl, epoch loop continues, no
                                        gradient clipping), critic step always
  controllers/ppo/ppo_utils.py:149-238  MLPActor (state-independent logstd = -0.5), MLPCritic, MLPActorCritic.step/act
  controllers/ppo/ppo_utils.py:358-400  random_sample (permutation, drop last), compute_returns_and_advantages
  math_and_models/neural_networks.py:18-54, distributions.py:9-33
  controllers/ppo/ppo.yaml              hyper-parameter names

What is different by construction (MI355X-first): the env batch lives on the GPU (HipVecEnv), the step kernel writes
observations / rewards / done flags / terminal observations straight into the [T, N, .] rollout tensors, nothing crosses
PCIe during a rollout, GAE runs as `scg_gae`, minibatches are gathered on device, and with several ranks (one process per
GPU, env shards) ONE flat RCCL all-reduce per minibatch carries both networks' gradients plus the approx-KL so that
every rank takes the same actor-gate branch.
"""
import math
import time
from dataclasses import dataclass, field

import torch
import torch.nn as nn
import torch.nn.functional as F

from safe_control_gym_amd import parallel


# ------------------------------------------------------------------ networks
class MLP(nn.Module):
    """Linear stack with a named torch.nn.functional activation (neural_networks.py:18-54); PyTorch default init."""

    def __init__(self, input_dim, output_dim, hidden_dims=(), act='relu', output_act=None):
        super().__init__()
        dims = [input_dim] + list(hidden_dims) + [output_dim]
        self.fcs = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
        self.act = getattr(F, act) if act else (lambda x: x)
        self.output_act = getattr(F, output_act) if output_act else (lambda x: x)

    def forward(self, x):
        for fc in self.fcs[:-1]:
            x = self.act(fc(x))
        return self.output_act(self.fcs[-1](x))


class MLPActor(nn.Module):
    def __init__(self, obs_dim, act_dim, hidden_dims, activation):
        super().__init__()
        self.pi_net = MLP(obs_dim, act_dim, hidden_dims, activation)
        self.logstd = nn.Parameter(-0.5 * torch.ones(act_dim))       # ppo_utils.py:166

    def forward(self, obs, c=None):
        """(mean, logstd).  `action_modifier(obs, mean, c)`, when set, filters the mean (Safe-Explorer's safety layer,
        safe_explorer/safe_ppo_utils.py:88-110); it is not a sub-module, so its parameters are not the actor's."""
        mean = self.pi_net(obs)
        if self.action_modifier is not None:
            mean = self.action_modifier(obs, mean, c)
        return mean, self.logstd

    action_modifier = None


class MLPCritic(nn.Module):
    def __init__(self, obs_dim, hidden_dims, activation):
        super().__init__()
        self.v_net = MLP(obs_dim, 1, hidden_dims, activation)

    def forward(self, obs):
        return self.v_net(obs)


LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


def normal_log_prob(mean, logstd, act):
    """Normal(mean, exp(logstd)).log_prob(act).sum(-1) (distributions.py:12-22)."""
    z = (act - mean) * torch.exp(-logstd)
    return (-0.5 * z * z - logstd - LOG_SQRT_2PI).sum(-1)


def normal_entropy(logstd):
    """Entropy summed over action dims (distributions.py:24-30); independent of the state."""
    return (0.5 + LOG_SQRT_2PI + logstd).sum(-1)


class MLPActorCritic(nn.Module):
    """Same state_dict layout as the reference (actor.pi_net.fcs.N.*, actor.logstd, critic.v_net.fcs.N.*), so the
    shipped checkpoints load directly."""

    def __init__(self, obs_dim, act_dim, hidden_dims=(64, 64), activation='tanh'):
        super().__init__()
        self.actor = MLPActor(obs_dim, act_dim, list(hidden_dims), activation)
        self.critic = MLPCritic(obs_dim, list(hidden_dims), activation)

    @torch.no_grad()
    def step(self, obs, c=None):
        """Sampled action, value, log-prob (ppo_utils.py:224-231), all on device."""
        mean, logstd = self.actor(obs, c)
        act = mean + torch.exp(logstd) * torch.randn_like(mean)
        return act, self.critic(obs).squeeze(-1), normal_log_prob(mean, logstd, act)

    @torch.no_grad()
    def act(self, obs, c=None):
        """Deterministic action = distribution mode (ppo_utils.py:233-238)."""
        return self.actor(obs, c)[0]


# ------------------------------------------------------------------ hyper-parameters
@dataclass
class PPOConfig:
    # names and defaults of controllers/ppo/ppo.yaml
    hidden_dim: int = 64
    activation: str = 'tanh'
    norm_obs: bool = False
    norm_reward: bool = False
    clip_obs: float = 10.0
    clip_reward: float = 10.0
    gamma: float = 0.99
    use_gae: bool = False
    gae_lambda: float = 0.95
    use_clipped_value: bool = False
    clip_param: float = 0.2
    target_kl: float = 0.01
    entropy_coef: float = 0.01
    opt_epochs: int = 10
    mini_batch_size: int = 64
    actor_lr: float = 0.0003
    critic_lr: float = 0.001
    max_grad_norm: float = 0.5          # configured but never applied upstream (ppo_utils.py:113-146)
    max_env_steps: int = 1000000
    rollout_batch_size: int = 4         # = number of envs (per rank)
    rollout_steps: int = 100
    eval_batch_size: int = 10
    extra: dict = field(default_factory=dict)

    @classmethod
    def from_dict(cls, d):
        known = {k: v for k, v in d.items() if k in cls.__dataclass_fields__}
        return cls(**known, extra={k: v for k, v in d.items() if k not in cls.__dataclass_fields__})


# ------------------------------------------------------------------ losses / update
def policy_loss_terms(ac, batch, clip_param):
    """ppo_utils.py:82-96."""
    mean, logstd = ac.actor(batch['obs'], batch.get('c'))
    logp = normal_log_prob(mean, logstd, batch['act'])
    ratio = torch.exp(logp - batch['logp'])
    adv = batch['adv']
    clip_adv = torch.clamp(ratio, 1 - clip_param, 1 + clip_param) * adv
    policy_loss = -torch.min(ratio * adv, clip_adv).mean()
    entropy_loss = -normal_entropy(logstd).expand(logp.shape[0]).mean()
    approx_kl = (batch['logp'] - logp).mean()
    return policy_loss, entropy_loss, approx_kl


def value_loss_term(ac, batch, clip_param, use_clipped_value):
    """ppo_utils.py:98-111."""
    v_cur = ac.critic(batch['obs']).squeeze(-1)
    ret = batch['ret']
    if use_clipped_value:
        v_old = batch['v']
        v_clipped = v_old + (v_cur - v_old).clamp(-clip_param, clip_param)
        return 0.5 * torch.max((v_cur - ret).pow(2), (v_clipped - ret).pow(2)).mean()
    return 0.5 * (v_cur - ret).pow(2).mean()


class PPOAgent:
    def __init__(self, obs_dim, act_dim, cfg: PPOConfig, device):
        self.cfg = cfg
        self.device = torch.device(device)
        self.ac = MLPActorCritic(obs_dim, act_dim, [cfg.hidden_dim] * 2, cfg.activation).to(device)
        self.actor_opt = torch.optim.Adam(self.ac.actor.parameters(), cfg.actor_lr)
        self.critic_opt = torch.optim.Adam(self.ac.critic.parameters(), cfg.critic_lr)
        parallel.broadcast_parameters([self.ac])
        self._bucket = None
        # HIP-graph replay of the minibatch step (launch-bound: ~100 tiny kernels on a 6->64->64->2 MLP); the
        # eager path below stays the reference implementation and is what the CPU tests exercise
        self.use_graphs = self.device.type == 'cuda' and bool(cfg.extra.get('cuda_graphs', True))
        self._g = None
        # flatten NOW: every graph captured later (rollout and update) must see the final parameter storage
        self._flat = self._flatten() if self.use_graphs else None
        # fused MFMA learner (csrc/scg_learn.hip): forward + backward of both networks + approx-KL in ONE launch per
        # minibatch, a deterministic reduction and a gated Adam kernel, in place of ~100 PyTorch kernels; exact float32.
        # Shapes it does not serve (hidden not a multiple of 32 / > 128, 3 or > 4 actions, a safety-layer action
        # modifier) keep the PyTorch update below — chosen here, visibly, not silently at run time.
        from safe_control_gym_amd import _learn
        self.obs_dim, self.act_dim = obs_dim, act_dim
        self._perm_key, self._perm_count = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF, 0
        self._perm_dev = None               # device mirror {base key, epochs drawn} (scg_random_permutation_keyed), made on first use
        self.use_fused = (self.use_graphs and bool(cfg.extra.get('fused_update', True))
                          and _learn.supported(obs_dim, cfg.hidden_dim, act_dim, cfg.activation))
        self._fused = None
        self.dp_path, self.dp_capture_error = None, None                   # how the last data-parallel epoch ran (_dp_epoch)
        self._fused_step_ok = bool(cfg.extra.get('fused_step', True))      # (False: scg_ppo_grad + scg_adam_gated also on one rank — A/B, tests)

    # ---- graphed update -------------------------------------------------------------------------------------
    def _flatten(self):
        """Parameters and gradients of both networks as views of two flat buffers (+1 gradient slot for approx_kl, so
        that one all-reduce carries everything), Adam moments next to them."""
        actor, critic = list(self.ac.actor.parameters()), list(self.ac.critic.parameters())
        params = actor + critic
        n_a = sum(p.numel() for p in actor)
        n = sum(p.numel() for p in params)
        flat_p = torch.cat([p.data.reshape(-1) for p in params]).contiguous()
        flat_g = torch.zeros(n + 1, device=self.device)
        off = 0
        for p in params:
            k = p.numel()
            p.data = flat_p[off:off + k].view_as(p)
            p.grad = flat_g[off:off + k].view_as(p)
            off += k
        lr = torch.cat([torch.full((n_a,), float(self.cfg.actor_lr)), torch.full((n - n_a,), float(self.cfg.critic_lr))]).to(self.device)
        is_critic = torch.cat([torch.zeros(n_a, dtype=torch.bool), torch.ones(n - n_a, dtype=torch.bool)]).to(self.device)
        return {'p': flat_p, 'g': flat_g, 'm': torch.zeros(n, device=self.device), 'v': torch.zeros(n, device=self.device),
                'lr': lr, 'is_critic': is_critic, 'n_a': n_a, 'n': n,
                'steps': torch.zeros(2, device=self.device)}          # Adam step counts: actor, critic

    # ---- fused update (scg_ppo_grad / scg_adam_gated) -------------------------------------------------------
    def _layouts(self):
        """Offsets of every tensor inside the flat parameter vector built by _flatten (actor parameters, then critic's)."""
        from safe_control_gym_amd._learn import MlpLayout
        off, table = 0, {}
        for prefix, mod in (('actor', self.ac.actor), ('critic', self.ac.critic)):
            for name, prm in mod.named_parameters():
                table[f'{prefix}.{name}'] = off
                off += prm.numel()
        def lay(prefix, net):
            return MlpLayout(*[table[f'{prefix}.{net}.fcs.{i}.{k}'] for i in range(3) for k in ('weight', 'bias')])
        return lay('actor', 'pi_net'), lay('critic', 'v_net'), table['actor.logstd'], off

    def _build_fused(self, data, mb):
        import ctypes as C
        from safe_control_gym_amd import _learn
        cfg, dev = self.cfg, self.device
        D = _learn.lib(self.obs_dim, cfg.hidden_dim, self.act_dim, cfg.activation)
        F = {'lib': D, 'data': data}
        a_lay, c_lay, ls_off, n = self._layouts()
        assert n == self._flat['n']
        # workgroups per network: one per CU for the pair, each of 4 waves walking 32-row tiles.  Of {CU/2, CU/2 - 1} take
        # the one with fewer tiles on the busiest wave, and on a tie the smaller: a launch that leaves two CUs free runs next
        # to the one-workgroup evaluation kernel of AsyncEvaluator instead of queueing a workgroup behind it (the gradient
        # workgroup needs a CU's whole LDS) — mini_batch_size = 127 x 512 = 65 024 makes that split exact on 256 CUs.
        half = max(1, torch.cuda.get_device_properties(dev).multi_processor_count // 2)
        tiles = mb // 32
        busiest = lambda w: -(-tiles // (4 * w))               # noqa: E731
        n_wg = half - 1 if (half > 1 and busiest(half - 1) <= busiest(half)) else half
        n_wg = min(n_wg, max(1, mb // 128))                     # every wave of every workgroup gets at least one tile
        F['ws'] = torch.empty(D.scg_ppo_grad_workspace_bytes(n_wg), dtype=torch.uint8, device=dev)
        F['stats'] = torch.zeros(4, device=dev)
        F['stats_acc'] = torch.zeros(5, device=dev)
        F['adam_sync'] = torch.zeros(1, dtype=torch.int32, device=dev)       # scg_adam_gated's block counter
        F['idx'] = torch.zeros(mb, dtype=torch.int32, device=dev)
        p = lambda t: t.data_ptr()                              # noqa: E731
        F['args'] = _learn.PpoGradArgs(
            d_params=p(self._flat['p']), actor=a_lay, critic=c_lay, logstd_off=ls_off, n_params=n,
            d_obs=p(data['obs']), d_act=p(data['act']), d_logp_old=p(data['logp']), d_adv=p(data['adv']), d_ret=p(data['ret']),
            d_v_old=p(data['v']), d_idx=p(F['idx']), batch=mb, clip_param=float(cfg.clip_param),
            entropy_coef=float(cfg.entropy_coef), use_clipped_value=int(bool(cfg.use_clipped_value)), n_workgroups=n_wg,
            d_workspace=p(F['ws']), d_grad=p(self._flat['g']), d_stats=p(F['stats']))
        F['C'] = C
        return F

    def _fused_grad(self, F):
        """Gradients of the minibatch F['idx'] into the flat gradient buffer (+ approx_kl in its last slot)."""
        from safe_control_gym_amd import _learn
        st = F['C'].c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _learn.check(F['lib'], F['lib'].scg_ppo_grad(F['C'].byref(F['args']), st))

    def _fused_adam(self, F, grad_scale=1.0):
        """The two gated Adam steps on the flat buffers; `grad_scale` multiplies every read of the gradient buffer (1 / world after
        the SUM all-reduce of a data-parallel step: no division launch in between)."""
        from safe_control_gym_amd import _learn
        fl, cfg, C = self._flat, self.cfg, F['C']
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _learn.check(F['lib'], F['lib'].scg_adam_gated_scaled(
            fl['p'].data_ptr(), fl['g'].data_ptr(), fl['m'].data_ptr(), fl['v'].data_ptr(), fl['n'], fl['n_a'],
            float(cfg.actor_lr), float(cfg.critic_lr), fl['steps'].data_ptr(), float(cfg.target_kl),
            F['stats_acc'].data_ptr(), F['stats'].data_ptr(), F['adam_sync'].data_ptr(), float(grad_scale), st))

    def _dp_minibatches(self, F, perm, n_mb, world):
        """Data-parallel minibatch loop of one epoch: gradient kernel -> ONE flat SUM all-reduce (both networks' gradients + the
        approx-KL slot) -> gated Adam reading the sum x 1 / world."""
        for j in range(n_mb):
            F['args'].d_idx = perm[j].data_ptr()
            self._fused_grad(F)
            parallel.all_reduce_sum_(self._flat['g'])
            self._fused_adam(F, 1.0 / world)

    def _dp_epoch(self, F, perm, bank, n_mb, world):
        """One epoch of data-parallel optimiser steps.  Over RCCL (backend "nccl") the whole loop — n_mb x (gradient kernel, reduction,
        all-reduce, Adam) — is ONE HIP-graph replay per epoch: the collective is captured with the kernels around it, so an iteration
        costs the host `opt_epochs` replays instead of n_mb x opt_epochs enqueue round trips (64 per iteration in bench.py's
        configuration).  The first epoch that sees a (permutation bank, n_mb) pair runs eagerly (it warms the communicator up for this
        message size), the second captures; a capture that fails falls back to the eager loop for good (recorded in `dp_path`)."""
        graphs = F.setdefault('dp_graphs', {})
        cfg = self.cfg              # (every value the captured launches carry BY VALUE is part of the key: a changed cfg re-captures)
        key = (bank, n_mb, world, float(cfg.actor_lr), float(cfg.critic_lr), float(cfg.target_kl), float(cfg.clip_param),
               float(cfg.entropy_coef), bool(cfg.use_clipped_value), bool(self._fused_step_ok))
        state = graphs.get(key)
        want = parallel.collectives_capturable() and self.cfg.extra.get('graph_collectives', True)
        if not want or state == 'eager':
            self.dp_path = 'eager'
            return self._dp_minibatches(F, perm, n_mb, world)
        if state is None:
            graphs[key] = 'warm'
            self.dp_path = 'eager (first epoch on this buffer: warm-up)'
            return self._dp_minibatches(F, perm, n_mb, world)
        if state == 'warm':
            try:
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                # thread_local: with several ranks ProcessGroupNCCL's watchdog THREAD queries events while this thread captures; under
                # the default (global) mode such a call from another thread invalidates the capture
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    self._dp_minibatches(F, perm, n_mb, world)
                graphs[key] = state = g
            except Exception as exc:                                  # noqa: BLE001  (no capture support in this build: stay eager)
                graphs[key] = 'eager'
                self.dp_capture_error = repr(exc)[:200]
                self.dp_path = 'eager (capture failed)'
                import warnings
                warnings.warn(f'PPO data-parallel epoch: HIP-graph capture failed, staying on the eager loop ({self.dp_capture_error})')
                return self._dp_minibatches(F, perm, n_mb, world)
        state.replay()
        self.dp_path = f'one graph replay per epoch ({n_mb} x (grad, reduce, all-reduce, adam))'

    def _fused_step(self, F):
        """One optimiser step on ONE rank in two launches (scg_ppo_step): gradient kernel, then reduction + gated Adam in one kernel.
        The step counts are double-buffered — bank 0 is `_flat['steps']` (what checkpoints and the other update paths read), bank 1 a
        scratch pair; `_update_fused` leaves the current counts in bank 0."""
        from safe_control_gym_amd import _learn
        fl, cfg, C = self._flat, self.cfg, F['C']
        if 'steps_b' not in fl:
            fl['steps_b'] = torch.zeros(2, device=self.device)
            fl['bank'] = 0
        banks = (fl['steps'], fl['steps_b'])
        src, dst = banks[fl['bank']], banks[1 - fl['bank']]
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _learn.check(F['lib'], F['lib'].scg_ppo_step(C.byref(F['args']), fl['m'].data_ptr(), fl['v'].data_ptr(), float(cfg.actor_lr),
                                                     float(cfg.critic_lr), src.data_ptr(), dst.data_ptr(), float(cfg.target_kl),
                                                     F['stats_acc'].data_ptr(), st))
        fl['bank'] = 1 - fl['bank']

    def _capped(self, n_mb):
        """Minibatches walked per epoch: all of the shuffled epoch, or the first extra['minibatches_per_epoch'] of them (every update path)."""
        cap = self.cfg.extra.get('minibatches_per_epoch')
        return max(1, min(n_mb, int(cap))) if cap else n_mb

    def _perm_state(self):
        """{base key, epochs drawn so far} in device memory: what scg_random_permutation_keyed derives each epoch's key from, so that
        an update captured in a HIP graph shuffles afresh at every replay.  Mirrors (self._perm_key, self._perm_count)."""
        vals = [self._perm_key - (1 << 64) if self._perm_key >= (1 << 63) else self._perm_key, self._perm_count]
        if self._perm_dev is None:
            self._perm_dev = torch.tensor(vals, dtype=torch.int64, device=self.device)
        return self._perm_dev

    def _sync_perm_state(self):
        """(after load_state_dict: in place — captured graphs alias the tensor)"""
        if self._perm_dev is not None:
            vals = [self._perm_key - (1 << 64) if self._perm_key >= (1 << 63) else self._perm_key, self._perm_count]
            self._perm_dev.copy_(torch.tensor(vals, dtype=torch.int64))

    def _update_fused(self, data, generator=None, lazy=False):
        """lazy=True: enqueue only — nothing is read back; the statistics stay in the device tensor returned as 'stats_dev'
        (sums over the update's minibatches of policy / value / entropy loss, approx-KL, actor steps taken; valid until the next
        update's kernels run).  Every launch of this function is capturable in a HIP graph (PPO._iteration_graph)."""
        cfg = self.cfg
        M = data['obs'].shape[0]
        mb = min(cfg.mini_batch_size, M)
        assert mb >= 32 and mb % 32 == 0, 'update() routes other minibatch sizes to the PyTorch path'     # 32-sample tiles
        n_mb = M // mb
        assert n_mb != 0, 'num_mini_batch is 0'
        # extra['minibatches_per_epoch']: a PARTIAL epoch — the first k minibatches of the epoch's permutation (a uniform subsample of
        # the rollout; upstream always walks the whole permutation, ppo_utils.py:358-371).  With 65 536 envs x 32 steps an iteration
        # holds 2 M samples: the number of optimiser steps per iteration, not the number of samples seen, is what the KL-limited
        # policy iteration needs, and the learner is the whole cost of the iteration (bench.py's PPO leg says which it uses).
        n_mb = self._capped(n_mb)
        for k, v in data.items():
            assert v.dtype == torch.float32 and v.is_contiguous(), k
        fkey = (M, mb, float(cfg.clip_param), float(cfg.entropy_coef), bool(cfg.use_clipped_value))   # (what the argument block carries by value)
        if self._fused is None or self._fused['key'] != fkey:
            static = {k: (v if k in ('obs', 'act', 'logp', 'v') else torch.empty_like(v)) for k, v in data.items()}
            self._fused = self._build_fused(static, mb)
            self._fused['key'] = fkey
        F = self._fused
        from safe_control_gym_amd import _learn
        if 'perm' not in F or F['perm'].shape[1] != n_mb * mb:
            F['perm'] = torch.empty(2, n_mb * mb, dtype=torch.int32, device=self.device)
            F.pop('dp_graphs', None)                                # (captured epochs hold the old buffer's addresses)
        for k, v in data.items():
            if F['data'][k].data_ptr() != v.data_ptr():
                F['data'][k].copy_(v)
        F['stats_acc'].zero_()
        world = parallel.world_size()
        key_state = self._perm_state()
        with torch.cuda.device(self.device):
            for ep in range(cfg.opt_epochs):
                if generator is not None:       # (tests: torch's own shuffle, reproducible against the PyTorch update)
                    perm = torch.randperm(M, device=self.device, generator=generator)[:n_mb * mb].to(torch.int32).view(n_mb, mb)
                else:
                    # one launch: keyed Feistel permutation of range(M) (csrc/scg_learn.hip), the key derived ON THE DEVICE from
                    # {base key, epochs drawn} + this epoch's offset.  Two index buffers, alternating by epoch (bank = ep % 2 — a
                    # property of the launch sequence, not of a host counter: the sequence may be a captured graph).
                    perm = F['perm'][ep % 2]
                    _learn.check(F['lib'], F['lib'].scg_random_permutation_keyed(perm.data_ptr(), M, n_mb * mb, key_state.data_ptr(), ep,
                                                                                 F['C'].c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
                    perm = perm.view(n_mb, mb)
                if world > 1 or cfg.extra.get('force_data_parallel'):   # gradients of both networks + approx_kl: one collective per step
                    # (force_data_parallel: the data-parallel path on ONE rank — tests and the 1-GPU measurement of its fixed cost)
                    if generator is not None:
                        self._dp_minibatches(F, perm, n_mb, world)
                    else:
                        self._dp_epoch(F, perm, ep % 2, n_mb, world)
                else:
                    for j in range(n_mb):
                        F['args'].d_idx = perm[j].data_ptr()
                        if self._fused_step_ok:
                            self._fused_step(F)                     # gradient kernel + (reduction and gated Adam in one launch)
                        else:
                            self._fused_grad(F)
                            self._fused_adam(F)
                F['keep'] = perm                                    # (the index rows must outlive the queued launches)
            if generator is None:                                   # the epochs drawn, on the stream (and its host mirror)
                key_state[1:2].add_(cfg.opt_epochs)
                self._perm_count += cfg.opt_epochs
            if self._flat.get('bank'):                              # an odd number of fused steps: the counts are in the scratch bank
                self._flat['steps'].copy_(self._flat['steps_b'])
                self._flat['bank'] = 0
        if lazy:
            return {'stats_dev': F['stats_acc'], 'minibatches': cfg.opt_epochs * n_mb}
        return self.stats_of(F['stats_acc'], cfg.opt_epochs * n_mb)

    @staticmethod
    def stats_of(stats_dev, minibatches):
        """The reference's averaged loss statistics (ppo_utils.py:140-146) from an update's device sums — the one host read."""
        st = (stats_dev / minibatches).tolist()
        return {'policy_loss': st[0], 'value_loss': st[1], 'entropy_loss': st[2], 'approx_kl': st[3],
                'actor_steps': int(round(st[4] * minibatches)), 'minibatches': minibatches}

    def _build_graphs(self, data, mb):
        cfg = self.cfg
        G = dict(self._flat)
        G['idx'] = torch.zeros(mb, dtype=torch.long, device=self.device)
        G['data'] = data
        G['stats'] = torch.zeros(5, device=self.device)              # policy, value, entropy, kl sums; actor steps
        b1, b2, eps = 0.9, 0.999, 1e-8                               # torch.optim.Adam defaults, as upstream

        def fwd_bwd():
            batch = {k: v.index_select(0, G['idx']) for k, v in data.items()}
            policy_loss, entropy_loss, approx_kl = policy_loss_terms(self.ac, batch, cfg.clip_param)
            value_loss = value_loss_term(self.ac, batch, cfg.clip_param, cfg.use_clipped_value)
            G['g'].zero_()
            (policy_loss + cfg.entropy_coef * entropy_loss).backward()
            value_loss.backward()
            G['g'][-1] = approx_kl.detach()
            G['losses'] = torch.stack([policy_loss.detach(), value_loss.detach(), entropy_loss.detach()])

        def adam():
            kl = G['g'][-1]
            gate = (kl <= 1.5 * cfg.target_kl) if cfg.target_kl > 0 else torch.ones((), dtype=torch.bool, device=self.device)
            G['steps'] += torch.stack([gate.to(torch.float32), torch.ones((), device=self.device)])
            take = G['is_critic'] | gate                               # per element: does its optimiser step now?
            g = G['g'][:-1]
            m = torch.where(take, b1 * G['m'] + (1 - b1) * g, G['m'])
            v = torch.where(take, b2 * G['v'] + (1 - b2) * g * g, G['v'])
            t = torch.where(G['is_critic'], G['steps'][1], G['steps'][0]).clamp(min=1.0)
            bc1 = 1 - torch.pow(b1, t)
            bc2 = 1 - torch.pow(b2, t)
            upd = G['lr'] / bc1 * m / (v.sqrt() / bc2.sqrt() + eps)
            G['p'].sub_(torch.where(take, upd, torch.zeros_like(upd)))
            G['m'].copy_(m)
            G['v'].copy_(v)
            G['stats'] += torch.cat([G['losses'], kl.reshape(1), gate.to(torch.float32).reshape(1)])

        s = torch.cuda.Stream(self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        keep = {k: G[k].clone() for k in ('p', 'm', 'v', 'steps')}
        with torch.cuda.stream(s):
            for _ in range(3):                                           # warm-up (allocator, autograd), state restored below
                fwd_bwd()
                adam()
        torch.cuda.current_stream(self.device).wait_stream(s)
        for k, t in keep.items():
            G[k].copy_(t)
        G['stats'].zero_()
        G['fb'], G['ad'] = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(G['fb']):
            fwd_bwd()
        with torch.cuda.graph(G['ad'], pool=G['fb'].pool()):
            adam()
        for k, t in keep.items():
            G[k].copy_(t)
        G['stats'].zero_()
        return G

    def _update_graphed(self, data, generator=None):
        cfg = self.cfg
        M = data['obs'].shape[0]
        mb = min(cfg.mini_batch_size, M)
        n_mb = M // mb
        n_mb = self._capped(n_mb)
        assert n_mb != 0, 'num_mini_batch is 0'
        if self._g is None or self._g['key'] != (M, mb):
            # static copies of the per-iteration inputs (the rollout tensors keep their addresses already)
            static = {k: (v if k in ('obs', 'act', 'logp', 'v') else torch.empty_like(v)) for k, v in data.items()}
            self._g = self._build_graphs(static, mb)
            self._g['key'] = (M, mb)
        G = self._g
        for k, v in data.items():
            if G['data'][k].data_ptr() != v.data_ptr():
                G['data'][k].copy_(v)
        G['stats'].zero_()
        world = parallel.world_size()
        for _ in range(cfg.opt_epochs):
            perm = torch.randperm(M, device=self.device, generator=generator)[:n_mb * mb].view(n_mb, mb)
            for j in range(n_mb):
                G['idx'].copy_(perm[j])
                G['fb'].replay()
                if world > 1:                                            # gradients of both networks + approx_kl, one collective
                    parallel.all_reduce_sum_(G['g'])
                    G['g'].div_(world)
                G['ad'].replay()
        st = (G['stats'] / (cfg.opt_epochs * n_mb)).tolist()
        return {'policy_loss': st[0], 'value_loss': st[1], 'entropy_loss': st[2], 'approx_kl': st[3],
                'actor_steps': int(round(st[4] * cfg.opt_epochs * n_mb)), 'minibatches': cfg.opt_epochs * n_mb}

    def state_dict(self):
        sd = {'ac': self.ac.state_dict(), 'actor_opt': self.actor_opt.state_dict(), 'critic_opt': self.critic_opt.state_dict()}
        if self._flat is not None:      # graph / fused modes: the Adam moments live in the flat buffers (torch optimisers never step)
            sd['flat_adam'] = {k: self._flat[k].clone() for k in ('m', 'v', 'steps')}
        sd['perm_state'] = (self._perm_key, self._perm_count)       # the fused update's keyed minibatch permutations
        return sd

    def load_state_dict(self, sd):
        self.ac.load_state_dict(sd['ac'])
        if 'actor_opt' in sd:
            self.actor_opt.load_state_dict(sd['actor_opt'])
            self.critic_opt.load_state_dict(sd['critic_opt'])
        if 'perm_state' in sd:
            self._perm_key, self._perm_count = int(sd['perm_state'][0]), int(sd['perm_state'][1])
            self._sync_perm_state()
        if 'flat_adam' in sd and self._flat is not None:
            for k, t in sd['flat_adam'].items():
                self._flat[k].copy_(t.to(self.device))      # in place: captured graphs alias these buffers
        elif 'flat_adam' in sd and int(sd['flat_adam']['steps'].max()) > 0:
            raise ValueError('checkpoint carries the flat Adam moments of a graph / fused-mode agent; this agent steps torch.optim '
                             "optimisers (cuda_graphs off) and would silently restart them from zero — load it with cuda_graphs on")
        elif self._flat is not None and 'actor_opt' in sd and sd['actor_opt'].get('state'):
            raise ValueError('checkpoint carries torch.optim state only (saved with cuda_graphs off); load it with '
                             "extra={'cuda_graphs': False} or re-save — the flat Adam moments would silently restart from zero")

    def fused_update_serves(self, n_rows):
        """Whether update() takes the MFMA learner for a rollout of n_rows samples (32-row tiles: minibatch sizes it cannot serve
        EXACTLY take the graphed PyTorch update — same result as the reference's minibatching — instead of being rounded down)."""
        mb0 = min(self.cfg.mini_batch_size, n_rows)
        return bool(self.use_fused and self.ac.actor.action_modifier is None and mb0 >= 32 and mb0 % 32 == 0)

    def update(self, data, generator=None, perms=None, lazy=False):
        """`data`: dict of flat [M, .] tensors (obs, act, logp, adv, ret, v).  Epochs x shuffled minibatches, drop last
        (ppo_utils.py:113-146, :358-371).  Returns the reference's averaged loss statistics.
        perms (tests): one index permutation of range(M) per epoch instead of torch.randperm (eager path).
        lazy (fused path only): enqueue without reading anything back — see _update_fused."""
        if perms is not None:
            assert not self.use_graphs, 'explicit permutations are an eager-path (test) facility'
        if self.fused_update_serves(data['obs'].shape[0]):
            return self._update_fused(data, generator, lazy=lazy)
        if self.use_graphs:
            return self._update_graphed(data, generator)
        cfg = self.cfg
        M = data['obs'].shape[0]
        mb = min(cfg.mini_batch_size, M)
        n_mb = M // mb
        n_mb = self._capped(n_mb)
        assert n_mb != 0, 'num_mini_batch is 0'
        if self._bucket is None:
            params = list(self.ac.actor.parameters()) + list(self.ac.critic.parameters())
            self._bucket = parallel.FlatBucket(params, n_scalars=1)
        stats = torch.zeros(4, device=data['obs'].device)
        n_actor_steps = 0
        for ep in range(cfg.opt_epochs):
            perm = (torch.as_tensor(perms[ep], device=data['obs'].device) if perms is not None
                    else torch.randperm(M, device=data['obs'].device, generator=generator))[:n_mb * mb].view(n_mb, mb)
            for idx in perm:
                batch = {k: v[idx] for k, v in data.items()}
                policy_loss, entropy_loss, approx_kl = policy_loss_terms(self.ac, batch, cfg.clip_param)
                value_loss = value_loss_term(self.ac, batch, cfg.clip_param, cfg.use_clipped_value)
                self.actor_opt.zero_grad(set_to_none=False)
                self.critic_opt.zero_grad(set_to_none=False)
                (policy_loss + cfg.entropy_coef * entropy_loss).backward()
                value_loss.backward()
                if parallel.world_size() > 1:
                    # one flat all-reduce: both networks' gradients + approx_kl, so every rank gates identically
                    self._bucket.pack([approx_kl.detach()])
                    self._bucket.all_reduce_mean()
                    kl = self._bucket.unpack()[0]
                else:
                    kl = approx_kl.detach()
                # update only when no KL constraint or the constraint is satisfied (ppo_utils.py:127-131)
                if cfg.target_kl <= 0 or float(kl) <= 1.5 * cfg.target_kl:
                    self.actor_opt.step()
                    n_actor_steps += 1
                self.critic_opt.step()
                stats += torch.stack([policy_loss.detach(), value_loss.detach(), entropy_loss.detach(), kl])
        stats = (stats / (cfg.opt_epochs * n_mb)).tolist()
        return {'policy_loss': stats[0], 'value_loss': stats[1], 'entropy_loss': stats[2], 'approx_kl': stats[3],
                'actor_steps': n_actor_steps, 'minibatches': cfg.opt_epochs * n_mb}


# ------------------------------------------------------------------ collector + trainer
class PPO:
    """PPO.train_step / learn / run on a HipVecEnv (ppo.py:150-303)."""

    def __init__(self, env, cfg: PPOConfig, seed=0):
        from safe_control_gym_amd.rollout import gae_returns
        self._gae = gae_returns
        self.env, self.cfg = env, cfg
        self.device, self.dtype = env.device, env.dtype
        if self.dtype != torch.float32:
            raise ValueError('the PPO collector runs on float32 environments')
        self.N, self.T = env.num_envs, cfg.rollout_steps
        spec = env.spec
        self.obs_dim, self.act_dim = spec.obs_dim, spec.nu
        rank = torch.distributed.get_rank() if parallel.world_size() > 1 else 0
        torch.manual_seed(seed + 7919 * rank)       # per-rank action-sampling stream; weights are broadcast from rank 0
        self.agent = PPOAgent(self.obs_dim, self.act_dim, cfg, self.device)
        T, N = self.T, self.N
        f = dict(device=self.device, dtype=torch.float32)
        # rollout storage, [T(+1), N, .]; the step kernel writes obs[t+1], rew[t], done[t], flags[t], term_obs[t] in place
        self.obs = torch.zeros(T + 1, N, self.obs_dim, **f)
        self.act = torch.zeros(T, N, self.act_dim, **f)
        self.rew = torch.zeros(T, N, **f)
        self.done = torch.zeros(T, N, dtype=torch.uint8, device=self.device)
        self.flags = torch.zeros(T, N, dtype=torch.uint8, device=self.device)
        self.term_obs = torch.zeros(T, N, self.obs_dim, **f)
        self.v = torch.zeros(T, N, **f)
        self.logp = torch.zeros(T, N, **f)
        self._slots = [env.bind_outputs(obs=self.obs[t + 1], reward=self.rew[t], done=self.done[t], flags=self.flags[t],
                                        terminal_obs=self.term_obs[t], state=None, noisy_action=None, c_values=None,
                                        mse=None) for t in range(T)]
        self.total_steps = 0
        self._rollout_graph = None
        # running normalisers (ppo.py:71-76); with several ranks their moment all-reduce cannot sit inside a captured
        # graph, so that combination collects eagerly
        from safe_control_gym_amd.normalization import BaseNormalizer, MeanStdNormalizer, RewardStdNormalizer
        self.obs_normalizer = MeanStdNormalizer((self.obs_dim,), self.device, clip=cfg.clip_obs) if cfg.norm_obs else BaseNormalizer()
        self.reward_normalizer = (RewardStdNormalizer(cfg.gamma, self.device, clip=cfg.clip_reward) if cfg.norm_reward
                                  else BaseNormalizer())
        if cfg.norm_reward:         # eager: graphs captured later (and checkpoint loads, in place) share this storage
            self.reward_normalizer.ret = torch.zeros(self.N, dtype=torch.float64, device=self.device)
        self._normalise = cfg.norm_obs or cfg.norm_reward
        self._graph_rollout = self.agent.use_graphs and not (self._normalise and parallel.world_size() > 1)
        # fused rollout (scg_rollout_policy): the whole T-step collection is ONE launch with the actor on the matrix cores
        # inside the env kernel; values come from one batched critic pass afterwards.  Needs an env built with this
        # policy shape (HipVecEnv(..., policy=(hidden, activation))), the flat parameter vector, no normalisers.
        self._fused_rollout = (self.agent.use_fused and bool(cfg.extra.get('fused_rollout', True)) and not self._normalise
                               and getattr(env, 'policy_shape', None) == (cfg.hidden_dim, cfg.activation))
        self._ret_adv = None
        if self._fused_rollout:
            self._episode_acc = torch.zeros(N, 8, **f)
            self._v_all = torch.zeros(T + 1, N, **f)
            self._tv = torch.zeros(T, N, **f)
            # the collector's post-processing as library launches (scg_learn.h: scg_ppo_returns_*): persistent outputs
            from safe_control_gym_amd import _learn
            D = _learn.lib(self.obs_dim, cfg.hidden_dim, self.act_dim, cfg.activation)
            self._trunc = torch.zeros(T, N, dtype=torch.uint8, device=self.device)
            self._mask = torch.zeros(T, N, **f)
            self._rew_c = torch.zeros(T, N, **f)
            self._adv_n = torch.zeros(T, N, **f)
            self._moments = torch.zeros(3, **f)
            self._ret_scratch = torch.zeros(int(D.scg_ppo_returns_scratch_bytes()) // 4, **f)
        self.obs[0].copy_(self.obs_normalizer(env.reset_tensors()))
        # finished-episode statistics (VecRecordEpisodeStatistics), accumulated on device (four adjacent words: views of one vector)
        self._ep_tot = torch.zeros(4, device=self.device)
        self.ep_count, self.ep_return_sum, self.ep_length_sum, self.ep_violation_sum = (self._ep_tot[k] for k in range(4))

    # ---- rollout (ppo.py:266-284)
    def _collect_body(self):
        ac, env = self.agent.ac, self.env
        for t in range(self.T):
            act, v, logp = ac.step(self.obs[t])
            self.act[t], self.v[t], self.logp[t] = act, v, logp
            out, c_out = self._slots[t]
            env.step_tensors(self.act[t], out=out, c_out=c_out)
            if self._normalise:         # ppo.py:270-271 (the terminal observation stays raw, as upstream)
                self.obs[t + 1].copy_(self.obs_normalizer(self.obs[t + 1]))
                self.rew[t].copy_(self.reward_normalizer(self.rew[t], self.done[t]))
            d = self.done[t].to(torch.float32)
            self.ep_count += d.sum()
            self.ep_return_sum += (out.fin_return * d).sum()
            self.ep_length_sum += (out.fin_length.to(torch.float32) * d).sum()
            self.ep_violation_sum += (out.fin_violation * d).sum()

    def collect(self):
        self._collect_body()
        self.total_steps += self.T * self.N * parallel.world_size()

    @torch.no_grad()
    def _returns_body(self, dense, critic=None, rew_buf=None, v_buf=None):
        """Bootstrap values, returns, advantages (not yet normalised) and the advantage moments.  dense=True evaluates
        the critic on every terminal-observation slot (static shapes, HIP-graph capturable) instead of gathering the
        truncated ones; the values used are the same.  critic / rew_buf / v_buf: another agent's view of the same
        rollout (RARL's adversary)."""
        cfg = self.cfg
        critic = critic or self.agent.ac.critic
        rew_buf = self.rew if rew_buf is None else rew_buf
        v_buf = self.v if v_buf is None else v_buf
        last_val = critic(self.obs[self.T]).squeeze(-1)
        mask = 1.0 - self.done.to(torch.float32)
        # time truncation is not termination: bootstrap with the critic's value of the terminal observation
        trunc = (self.flags & 1).bool() & self.done.bool()
        if dense:
            tv = critic(self.term_obs.reshape(self.T * self.N, self.obs_dim)).reshape(self.T, self.N)
            terminal_v = torch.where(trunc, tv, torch.zeros_like(tv))
        else:
            terminal_v = torch.zeros_like(rew_buf)
            idx = trunc.nonzero(as_tuple=False)
            if idx.numel():
                terminal_v[idx[:, 0], idx[:, 1]] = critic(self.term_obs[idx[:, 0], idx[:, 1]]).squeeze(-1)
        rew = rew_buf.clone()
        ret, adv = self._gae(rew, v_buf, mask, terminal_v, last_val, cfg.gamma, cfg.gae_lambda, cfg.use_gae)
        moments = torch.stack([adv.sum(), (adv * adv).sum(), torch.full((), float(adv.numel()), device=adv.device)])
        return ret, adv, moments

    # ---- fused rollout front end -----------------------------------------------------------------------------
    def _policy_struct(self, deterministic=False):
        from safe_control_gym_amd import _lib as L
        a_lay, _, ls_off, _ = self.agent._layouts()
        return L.Policy(d_params=self.agent._flat['p'].data_ptr(), W1=a_lay.W1, b1=a_lay.b1, W2=a_lay.W2, b2=a_lay.b2,
                        W3=a_lay.W3, b3=a_lay.b3, logstd_off=ls_off, hidden=self.cfg.hidden_dim,
                        activation=L.POLICY_ACTS[self.cfg.activation], deterministic=int(deterministic))

    def _critic_batch(self, x, out, row_mask=None):
        """out[m] = critic(x[m]) through the MFMA forward kernel (scg_mlp_forward); row_mask (uint8 [m]): only 32-row tiles
        holding a flagged row are evaluated, the others return 0."""
        import ctypes as C
        from safe_control_gym_amd import _learn
        D = _learn.lib(self.obs_dim, self.cfg.hidden_dim, self.act_dim, self.cfg.activation)
        _, c_lay, _, _ = self.agent._layouts()
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            _learn.check(D, D.scg_mlp_forward(self.agent._flat['p'].data_ptr(), C.byref(c_lay), 1, x.data_ptr(), int(x.shape[0]),
                                              out.data_ptr(), row_mask.data_ptr() if row_mask is not None else None, st))

    @torch.no_grad()
    def _collect_fused(self, count_steps=True):
        """One launch for the T control steps (policy inside), two batched critic passes, scg_gae; the elementwise work between them
        (time-limit flags, masks, advantage moments, episode totals) as three library launches (round 6: ~25 PyTorch kernels before)."""
        import ctypes as C
        from safe_control_gym_amd import _learn
        cfg, T, N = self.cfg, self.T, self.N
        D = _learn.lib(self.obs_dim, cfg.hidden_dim, self.act_dim, cfg.activation)
        p = lambda t: C.c_void_p(t.data_ptr())                          # noqa: E731
        self.env.rollout_policy(self._policy_struct(), T, self.obs, self.act, self.logp, self.rew, self.done, self.flags,
                                terminal_obs=self.term_obs, episode_acc=self._episode_acc)
        self._critic_batch(self.obs.view((T + 1) * N, self.obs_dim), self._v_all.view(-1))
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            # time truncation is not termination (ppo.py:276-283): trunc = done & (flags & 1); mask = 1 - done; working copies of rew, v
            _learn.check(D, D.scg_ppo_returns_prepare(p(self.done), p(self.flags), p(self.rew), p(self._v_all), T, N, p(self._trunc), p(self._mask),
                                                      p(self._rew_c), p(self.v), st))
        # bootstrap with the critic's value of the terminal observation — evaluated only on the 32-row tiles that hold a truncated row,
        # 0 on every row that is not truncated: the pass's output IS the terminal_v of ppo_utils.py:389
        self._critic_batch(self.term_obs.view(T * N, self.obs_dim), self._tv.view(-1), row_mask=self._trunc.view(-1))
        if self._ret_adv is None:
            self._ret_adv = (torch.empty_like(self._rew_c), torch.empty_like(self._rew_c))
        ret, adv = self._gae(self._rew_c, self.v, self._mask, self._tv, self._v_all[T], cfg.gamma, cfg.gae_lambda, cfg.use_gae, out=self._ret_adv)
        with torch.cuda.device(self.device):
            _learn.check(D, D.scg_ppo_returns_moments(p(adv), T, N, p(self._episode_acc), p(self._ret_scratch), p(self._moments), p(self._ep_tot), st))
        if count_steps:
            self.total_steps += T * N * parallel.world_size()
        return ret, adv, self._moments

    def _build_rollout_graph(self):
        """One HIP graph for the whole iteration front end: T x (policy forward, sampling, value, log-prob, env step
        kernel, episode statistics), then bootstrap values, scg_gae and the advantage moments."""
        dev = self.device
        s = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(3):                  # warm up the torch ops only (no env step: the simulation must not advance)
                self.agent.ac.step(self.obs[0])
                self._returns_body(dense=True)
        torch.cuda.current_stream(dev).wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._collect_body()
            out = self._returns_body(dense=True)
        return g, out

    # ---- returns / advantages / update (ppo.py:286-303)
    def _normalised(self, adv, moments):
        """Global advantage normalisation (ppo.py:300): population std, +1e-6; the moments are summed over the ranks."""
        with torch.no_grad():
            parallel.all_reduce_sum_(moments)
            if self._fused_rollout and moments is self._moments:         # one library launch on the collector's own moments vector
                import ctypes as C
                from safe_control_gym_amd import _learn
                D = _learn.lib(self.obs_dim, self.cfg.hidden_dim, self.act_dim, self.cfg.activation)
                with torch.cuda.device(self.device):
                    _learn.check(D, D.scg_ppo_returns_normalise(C.c_void_p(adv.data_ptr()), C.c_void_p(moments.data_ptr()), self.T, self.N,
                                                                C.c_void_p(self._adv_n.data_ptr()),
                                                                C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
                return self._adv_n
            mean = moments[0] / moments[2]
            std = torch.sqrt(torch.clamp(moments[1] / moments[2] - mean * mean, min=0.0))
            return (adv - mean) / (std + 1e-6)

    def _rollout_data(self, ret, adv):
        M = self.T * self.N
        return {'obs': self.obs[:self.T].reshape(M, self.obs_dim), 'act': self.act.reshape(M, self.act_dim),
                'logp': self.logp.reshape(M), 'adv': adv.reshape(M), 'ret': ret.reshape(M), 'v': self.v.reshape(M)}

    def _iteration_body(self):
        """One whole iteration as device work only — fused collection, bootstrap values, scg_gae, advantage normalisation, the
        update's epochs (keyed permutations, gradient kernels, reduction + gated Adam) — with no host read anywhere: the sequence
        PPO._iteration_graph captures.  Returns the update's lazy statistics."""
        ret, adv, moments = self._collect_fused(count_steps=False)
        return self.agent.update(self._rollout_data(ret, self._normalised(adv, moments)), lazy=True)

    def _iteration_key(self):
        """Everything an iteration's launches carry BY VALUE (kernel arguments at capture time): a change re-captures."""
        c = self.cfg
        return (getattr(self.env, 'seed_epoch', 0), self.T, self.N, c.opt_epochs, c.mini_batch_size, c.extra.get('minibatches_per_epoch'),
                float(c.actor_lr), float(c.critic_lr), float(c.target_kl), float(c.clip_param), float(c.entropy_coef),
                bool(c.use_clipped_value), float(c.gamma), float(c.gae_lambda), bool(c.use_gae), bool(self.agent._fused_step_ok))

    def _iteration_graph_ok(self):
        """One HIP-graph replay per iteration: the fused collector + the fused update on ONE rank (several ranks keep their
        per-epoch graphs with the all-reduces inside, PPOAgent._dp_epoch).  extra['iteration_graph'] = False keeps the per-launch
        enqueue (A/B, tests: same launches, same results bit for bit)."""
        a = self.agent
        return bool(self._fused_rollout and a.fused_update_serves(self.T * self.N) and parallel.world_size() == 1
                    and not self.cfg.extra.get('force_data_parallel') and self.cfg.extra.get('iteration_graph', True))

    def _run_iteration(self):
        """The iteration body, eagerly the first time (libraries load, buffers are built, kernel attributes are set), captured the
        second time, replayed from then on.  Host bookkeeping (step / epoch counters) follows every run."""
        key = self._iteration_key()
        st = getattr(self, '_iter_graph', None)
        if st is None or st['key'] != key:
            res = self._iteration_body()                    # (also after a key change: the eager run rebuilds what depends on it)
            self._iter_graph = {'key': key, 'graph': None, 'res': None}
        else:
            if st['graph'] is None:
                count = self.agent._perm_count              # capture RECORDS the launches (nothing runs): undo its host bookkeeping
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    st['res'] = self._iteration_body()
                self.agent._perm_count = count
                st['graph'] = g
            st['graph'].replay()
            self.agent._perm_count += self.cfg.opt_epochs
            res = dict(st['res'])
        self.total_steps += self.T * self.N * parallel.world_size()
        self._obs_row = self.T
        return res

    def train_step(self, lazy=False):
        """One PPO iteration (ppo.py:259-303).  lazy=True (fused collector + fused update): the iteration is ENQUEUED and the call
        returns without waiting for it — no synchronisation, no host read; the result carries the update's statistics as a device
        tensor ('stats_dev', see PPOAgent.stats_of) and three HIP events ('events': start, collection done, update done) whose
        elapsed times give the iteration's device time once they have completed.  The default reads the statistics back, as the
        reference's train_step returns floats."""
        t0 = time.perf_counter()
        dev_timing = self.device.type == 'cuda'
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if dev_timing else None
        if ev:
            ev[0].record()
        if self._iteration_graph_ok():
            res = self._run_iteration()
            ev[1] = None                                     # (one replay: no boundary between collection and update to mark)
        else:
            if self._fused_rollout:
                ret, adv, moments = self._collect_fused()
            elif self._graph_rollout:
                if self._rollout_graph is None or self._rollout_epoch != getattr(self.env, 'seed_epoch', 0):
                    # capture records the launches without running them (re-captured after env.seed(): the key is a kernel argument)
                    self._rollout_graph, self._rollout_out = self._build_rollout_graph()
                    self._rollout_epoch = getattr(self.env, 'seed_epoch', 0)
                self._rollout_graph.replay()
                self.total_steps += self.T * self.N * parallel.world_size()
                ret, adv, moments = self._rollout_out
                moments = moments.clone()
            else:
                self.collect()
                ret, adv, moments = self._returns_body(dense=False)
            data = self._rollout_data(ret, self._normalised(adv, moments))
            if ev:
                ev[1].record()                              # (an event, not a synchronize(): the stream keeps running into the update)
            res = self.agent.update(data, lazy=lazy and self.agent.fused_update_serves(self.T * self.N))
            if not self._fused_rollout:             # (the fused collector re-derives obs[0] from the simulator state)
                self.obs[0].copy_(self.obs[self.T])
            else:
                self._obs_row = self.T              # the CURRENT observation is the last row the collector wrote (checkpoint 'obs')
        if ev:
            ev[2].record()
        res['step'] = self.total_steps
        if lazy and 'stats_dev' in res:
            res['events'] = ev
            return res
        if 'stats_dev' in res:
            res = dict(PPOAgent.stats_of(res['stats_dev'], res['minibatches']), step=self.total_steps)
        if ev:
            ev[2].synchronize()
            res['collect_time'] = 1e-3 * ev[0].elapsed_time(ev[1]) if ev[1] is not None else None
            res['device_time'] = 1e-3 * ev[0].elapsed_time(ev[2])
        else:
            res['collect_time'] = None
        res['elapsed_time'] = time.perf_counter() - t0
        return res

    # ---- checkpoint / resume (ppo.py:112-148: same keys; the env's "random state" is the whole simulator workspace)
    def save(self, path, training=True):
        import os
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save(self.checkpoint_state(training), path)

    def checkpoint_state(self, training=True):
        """The checkpoint dict (subclasses add their keys: Safe-Explorer's 'safety_layer')."""
        state = {'agent': self.agent.state_dict(), 'obs_normalizer': self.obs_normalizer.state_dict(),
                 'reward_normalizer': self.reward_normalizer.state_dict()}
        if getattr(self.reward_normalizer, 'ret', None) is not None:
            state['reward_normalizer_ret'] = self.reward_normalizer.ret.cpu()
        for name, nz in (('obs', self.obs_normalizer), ('reward', self.reward_normalizer)):
            if hasattr(nz, 'rms'):
                state[f'{name}_normalizer_count'] = float(nz.rms.count)
        if training:
            state.update({'total_steps': self.total_steps, 'obs': self.obs[getattr(self, '_obs_row', 0)].cpu(),
                          'random_state': {'torch': torch.get_rng_state(),
                                           'torch_cuda': torch.cuda.get_rng_state(self.device) if self.device.type == 'cuda' else None},
                          'env_random_state': self.env.get_env_random_state()})
        return state

    def load(self, path, training=True):
        state = torch.load(path, map_location='cpu', weights_only=False)
        self.agent.load_state_dict(state['agent'])
        for name, nz in (('obs', self.obs_normalizer), ('reward', self.reward_normalizer)):
            if state.get(f'{name}_normalizer'):
                nz.load_state_dict(state[f'{name}_normalizer'])
                if f'{name}_normalizer_count' in state and hasattr(nz, 'rms'):
                    nz.rms.count.fill_(state[f'{name}_normalizer_count'])
        if 'reward_normalizer_ret' in state and hasattr(self.reward_normalizer, 'rms'):
            ret = state['reward_normalizer_ret'].to(self.device)
            if self.reward_normalizer.ret is not None and self.reward_normalizer.ret.shape == ret.shape:
                self.reward_normalizer.ret.copy_(ret)       # in place: a captured rollout graph aliases this tensor
            else:
                self.reward_normalizer.ret = ret
                self._rollout_graph = None                  # (re-capture with the new storage)
        if training and 'total_steps' in state:
            self.total_steps = state['total_steps']
            self.obs[0].copy_(state['obs'].to(self.device))
            self._obs_row = 0
            torch.set_rng_state(state['random_state']['torch'])
            if state['random_state'].get('torch_cuda') is not None and self.device.type == 'cuda':
                torch.cuda.set_rng_state(state['random_state']['torch_cuda'], self.device)
            self.env.set_env_random_state(state['env_random_state'])

    def episode_stats(self, reset=True):
        s = torch.stack([self.ep_count, self.ep_return_sum, self.ep_length_sum, self.ep_violation_sum])
        parallel.all_reduce_sum_(s)
        n = max(float(s[0]), 1.0)
        out = {'episodes': float(s[0]), 'ep_return': float(s[1]) / n, 'ep_length': float(s[2]) / n,
               'ep_constraint_violation': float(s[3]) / n}
        if reset:
            for t in (self.ep_count, self.ep_return_sum, self.ep_length_sum, self.ep_violation_sum):
                t.zero_()
        return out

    def learn(self, max_env_steps=None, log=None, target_return=None, eval_env=None, eval_every=1):
        """Train until max_env_steps (or until the evaluation return reaches target_return)."""
        max_env_steps = max_env_steps or self.cfg.max_env_steps
        history = []
        t_start = time.perf_counter()
        it = 0
        while self.total_steps < max_env_steps:
            res = self.train_step()
            it += 1
            res.update(self.episode_stats())
            res['wall_clock'] = time.perf_counter() - t_start
            if eval_env is not None and it % eval_every == 0:
                res['eval_return'] = evaluate(self.agent.ac, eval_env, obs_normalizer=self.obs_normalizer if self.cfg.norm_obs else None)['ep_return']
            history.append(res)
            if log:
                log(res)
            if target_return is not None and res.get('eval_return', -1e30) >= target_return:
                break
        return history


@torch.no_grad()
def _evaluate_fused_device(env, policy, episodes_per_env=1, chunk=None):
    """ONE launch on the current stream: deterministic actor inside the env kernel (scg_rollout_policy), per-env episode
    totals accumulated in the kernel.  Returns (device tensor [episodes, mean return, length, violations, mse], per-env
    accumulator [N, 8]) without touching the host.
    chunk (control steps per launch; None = all in one): the same rollout as a sequence of shorter launches — identical results (the
    simulator state lives in the env handle, the episode accumulators in `acc`, the noise is keyed by (env, episode, step))."""
    N, dev = env.num_envs, env.device
    steps = env.spec.max_episode_steps * episodes_per_env
    buf = getattr(env, '_eval_fused', None)
    if buf is None or buf['rew'].shape[0] != steps:
        f = dict(device=dev, dtype=torch.float32)
        nobs, nu = env.spec.obs_dim, env.spec.nu
        buf = {'obs': torch.zeros(steps + 1, N, nobs, **f), 'act': torch.zeros(steps, N, nu, **f),
               'logp': torch.zeros(steps, N, **f), 'rew': torch.zeros(steps, N, **f),
               'done': torch.zeros(steps, N, dtype=torch.uint8, device=dev), 'flags': torch.zeros(steps, N, dtype=torch.uint8, device=dev),
               'acc': torch.zeros(N, 8, **f)}
        env._eval_fused = buf
    env.reset_tensors()
    buf['acc'].zero_()
    chunk = steps if not chunk else max(1, min(int(chunk), steps))
    for t0 in range(0, steps, chunk):
        k = min(chunk, steps - t0)
        env.rollout_policy(policy, k, buf['obs'][t0:t0 + k + 1], buf['act'][t0:t0 + k], buf['logp'][t0:t0 + k], buf['rew'][t0:t0 + k],
                           buf['done'][t0:t0 + k], buf['flags'][t0:t0 + k], episode_acc=buf['acc'], max_episodes=episodes_per_env)
    a = buf['acc']
    n = a[:, 0].sum().clamp(min=1.0)
    return torch.stack([a[:, 0].sum(), a[:, 1].sum() / n, a[:, 2].sum() / n, a[:, 3].sum() / n, a[:, 4].sum() / n]), a


def _evaluate_fused_result(res, a, episodes_per_env=1):
    out = {'episodes': res[0], 'ep_return': res[1], 'ep_length': res[2], 'ep_constraint_violation': res[3], 'ep_mse': res[4]}
    if episodes_per_env == 1 and a is not None:
        out['metrics'] = episode_metrics(a[:, 1], a[:, 2], a[:, 3], a[:, 4], a[:, 0] > 0)
    return out


class AsyncEvaluator:
    """Deterministic evaluation of policy SNAPSHOTS on a second HIP stream, overlapped with training.

    The fused evaluation is one launch of one or a few workgroups that runs for CTRL_STEPS sequential control steps (6.5 ms for
    256 envs x 250 steps): it needs almost none of the chip but a third of a PPO iteration's time when it sits on the
    training stream.  launch() copies the flat parameters (on the training stream, i.e. after the update that produced them)
    and enqueues the evaluation on the side stream; poll() returns the result once the stream has finished it, without
    blocking; at most one evaluation is in flight.  The reference evaluates every `eval_interval` steps on a separate
    eval env (ppo.py:186-208); which weights get evaluated and what is measured are the same, only the waiting is gone."""

    def __init__(self, ppo, eval_env):
        if not ppo._fused_rollout or getattr(eval_env, 'policy_shape', None) != (ppo.cfg.hidden_dim, ppo.cfg.activation):
            raise ValueError('AsyncEvaluator needs the fused rollout path (envs built with the policy shape)')
        self.ppo, self.env, self.dev = ppo, eval_env, eval_env.device
        self.stream = torch.cuda.Stream(self.dev)
        self.params = ppo.agent._flat['p'].clone()
        self.policy = ppo._policy_struct(True)
        self.policy.d_params = self.params.data_ptr()
        self.host = torch.zeros(5, dtype=torch.float32).pin_memory()
        self.event, self.tag = None, None
        # Control steps per evaluation launch (None = the whole evaluation in one).  What the evaluation costs the training stream was
        # measured in round 6 (profiles/r06_eval_interference.txt): +0.13 ms of device time and +0.33 ms of wall clock per 4.2 ms iteration,
        # the SAME as one launch, as 25- or 5-step launches, as one captured graph per evaluation, and enqueued behind train_step or behind
        # the collector's launch — i.e. neither a collector workgroup waiting for "its" CU nor host time in launch(); not removed.
        self.chunk = ppo.cfg.extra.get('eval_chunk_steps')

    def launch(self, tag=None):
        """Snapshot the current weights and start evaluating them; False if the previous evaluation is still running."""
        if self.event is not None:
            return False
        main = torch.cuda.current_stream(self.dev)
        self.params.copy_(self.ppo.agent._flat['p'])
        self.stream.wait_stream(main)
        with torch.cuda.stream(self.stream):
            res, _ = _evaluate_fused_device(self.env, self.policy, 1, chunk=self.chunk)
            self.host.copy_(res, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record(self.stream)
        self.tag = tag
        return True

    def poll(self, wait=False):
        """Result dict (+ 'tag') of the finished evaluation, or None if none has finished."""
        if self.event is None:
            return None
        if wait:
            self.event.synchronize()
        elif not self.event.query():
            return None
        self.event = None
        out = _evaluate_fused_result(self.host.tolist(), None)
        out['tag'] = self.tag
        return out


@torch.no_grad()
def evaluate(ac, env, episodes_per_env=1, use_graph=None, obs_normalizer=None, policy=None):
    """Deterministic policy (action = mean, ppo_utils.py:233-238) on every env of `env` until each finished
    `episodes_per_env` episodes; returns mean episode return / length / violations / mse (batched counterpart of
    PPO.run, ppo.py:210-257).  An episode lasts at most CTRL_STEPS control steps, so the loop has a fixed length and no
    host synchronisation; on a GPU it is captured once per (policy, env) pair and replayed as one HIP graph.
    policy: an _lib.Policy with deterministic = 1 (PPO._policy_struct(True)) for an env built with that policy shape —
    the whole evaluation is then ONE launch of the fused rollout kernel."""
    N = env.num_envs
    steps = env.spec.max_episode_steps * episodes_per_env
    dev = env.device
    if policy is not None and obs_normalizer is None and getattr(env, 'policy_shape', None) is not None:
        res, a = _evaluate_fused_device(env, policy, episodes_per_env)
        return _evaluate_fused_result(res.tolist(), a, episodes_per_env)
    use_graph = (dev.type == 'cuda') if use_graph is None else use_graph
    cache = getattr(env, '_eval_cache', None)
    key = (id(ac), episodes_per_env, bool(use_graph), id(obs_normalizer))
    if cache is None or cache['key'] != key:
        # per-env totals of the first `episodes_per_env` episodes: one packed [N, 4] block (return, length, violations, mse — the layout
        # of the kernel's finished-episode statistics) + the episode count; `acc` exposes the columns as views
        tot, count = torch.zeros(N, 4, device=dev), torch.zeros(N, device=dev)
        acc = {'count': count, 'ret': tot[:, 0], 'length': tot[:, 1], 'viol': tot[:, 2], 'mse': tot[:, 3], 'tot': tot}

        def policy_obs():
            if obs_normalizer is None:
                return env.out.obs
            frozen = obs_normalizer.read_only       # evaluation never updates the statistics (ppo.py:218)
            obs_normalizer.set_read_only()
            o = obs_normalizer(env.out.obs)
            obs_normalizer.read_only = frozen
            return o

        def body():
            for _ in range(steps):
                out = env.step_tensors(ac.act(policy_obs()))
                # (five launches per step instead of thirteen: the loop is launch-bound — 250 steps of a 256-env batch)
                df = (out.done * (count < episodes_per_env)).to(torch.float32)
                tot.addcmul_(out.fin_stats.to(torch.float32), df[:, None])
                count.add_(df)

        graph = None
        if use_graph:
            env.reset_tensors()
            s = torch.cuda.Stream(dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                for _ in range(3):
                    ac.act(policy_obs())                 # warm up the torch ops (the env must not be stepped here)
            torch.cuda.current_stream(dev).wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                body()
        cache = {'key': key, 'acc': acc, 'body': body, 'graph': graph, 'ac': ac}
        env._eval_cache = cache
    acc = cache['acc']
    env.reset_tensors()
    acc['tot'].zero_()
    acc['count'].zero_()
    if cache['graph'] is not None:
        cache['graph'].replay()
    else:
        cache['body']()
    n = acc['count'].sum().clamp(min=1.0)
    res = torch.stack([acc['count'].sum(), acc['ret'].sum() / n, acc['length'].sum() / n, acc['viol'].sum() / n,
                       acc['mse'].sum() / n]).tolist()
    out = {'episodes': res[0], 'ep_return': res[1], 'ep_length': res[2], 'ep_constraint_violation': res[3], 'ep_mse': res[4]}
    if episodes_per_env == 1:
        out['metrics'] = episode_metrics(acc['ret'], acc['length'], acc['viol'], acc['mse'], acc['count'] > 0)
    return out


@torch.no_grad()
def episode_metrics(ep_return, ep_length, ep_violation_steps, ep_mse_sum, valid=None):
    """MetricExtractor.compute_metrics (experiments/base_experiment.py:392-421) from per-episode totals on the device:
    rmse = sqrt(mean of the step mse), failure = any violation in the episode, CVaR = mean of the worst half of the rmse
    (math_and_models/metrics/performance_metrics.py:6-31, alpha 0.5, upper range)."""
    if valid is not None:
        ep_return, ep_length, ep_violation_steps, ep_mse_sum = (t[valid] for t in (ep_return, ep_length, ep_violation_steps, ep_mse_sum))
    n = ep_return.numel()
    if n == 0:
        return {}
    rmse = torch.sqrt(ep_mse_sum.to(torch.float64) / ep_length.to(torch.float64).clamp(min=1.0))
    k = int(0.5 * n)
    worst = torch.sort(rmse).values[-k:].mean() if k > 0 else torch.sort(rmse).values.mean()     # [-0:] is the whole array upstream
    v = ep_violation_steps.to(torch.float64)
    vals = torch.stack([ep_length.to(torch.float64).mean(), ep_return.to(torch.float64).mean(), rmse.mean(),
                        rmse.std(unbiased=False), worst, (v > 0).to(torch.float64).mean(), v.mean(), v.std(unbiased=False)]).tolist()
    keys = ('average_length', 'average_return', 'average_rmse', 'rmse_std', 'worst_case_rmse_at_0.5', 'failure_rate',
            'average_constraint_violation', 'constraint_violation_std')
    return dict(zip(keys, vals))

"""HipVecEnv: the vectorised-environment boundary, one HIP launch per control step.

Mirrors the reference's VecEnv contract
(/root/reference/safe_control_gym/envs/env_wrappers/vectorized_env/vec_env.py:13-142 and
dummy_vec_env.py:12-119): ``reset() -> (obs[N,.], {'n': infos})``,
``step_async / step_wait -> (obs, rews, dones, {'n': infos})`` with auto-reset,
``terminal_observation`` / ``terminal_info``, ``get_attr / set_attr / env_method``,
``get_env_random_state / set_env_random_state``, ``close``, ``num_envs``,
``observation_space``, ``action_space``.

Two ways to consume a step:
* the reference API above (NumPy in / NumPy out, ``info['n']`` is a lazy per-env view that
  only touches the host when somebody indexes it);
* ``step_tensors(actions)`` — device tensors in, a ``StepTensors`` bundle of device tensors
  out, no host synchronisation at all (used by the PPO/SAC collectors in this package).
"""
import ctypes as C
from abc import ABC, abstractmethod
from collections.abc import Sequence

import numpy as np
import torch

from safe_control_gym_amd import _lib as L
from safe_control_gym_amd.env_config import EnvSpec


class VecEnv(ABC):
    """Same abstract surface as the reference's VecEnv (vec_env.py:13-142)."""
    closed = False
    viewer = None
    metadata = {'render.modes': ['human', 'rgb_array']}

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space

    @abstractmethod
    def reset(self):
        pass

    @abstractmethod
    def step_async(self, actions):
        pass

    @abstractmethod
    def step_wait(self):
        pass

    def close_extras(self):
        pass

    def close(self):
        if self.closed:
            return
        self.close_extras()
        self.closed = True

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    @property
    def unwrapped(self):
        return self

    @abstractmethod
    def get_attr(self, attr_name, indices=None):
        pass

    @abstractmethod
    def set_attr(self, attr_name, values, indices=None):
        pass

    @abstractmethod
    def env_method(self, method_name, method_args=None, method_kwargs=None, indices=None):
        pass

    def _get_indices(self, indices):
        if indices is None:
            indices = range(self.num_envs)
        elif isinstance(indices, int):
            indices = [indices]
        return indices


def checked_seed(seed):
    """BenchmarkEnv.seed -> gymnasium's seeding.np_random (benchmark_env.py:193-214): a seed is a non-negative integer.  gymnasium ^0.28 raises
    its own `error.Error` there; that class does not exist here (no gymnasium), so: ValueError, instead of silently truncating / wrapping."""
    if isinstance(seed, bool) or not isinstance(seed, (int, np.integer)):
        raise ValueError(f'Seed must be a python integer, actual type: {type(seed)}')
    if seed < 0:
        raise ValueError(f'Seed must be greater or equal to zero, actual value: {seed}')
    return int(seed)


class StepTensors:
    """Device-resident outputs of one vectorised step (all torch tensors on the env's device)."""
    __slots__ = ('obs', 'reward', 'done', 'flags', 'c_values', 'mse', 'terminal_obs', 'state', 'noisy_action', 'fin_stats')

    # finished-episode totals where done: columns of the packed [N, 4] array (return, length, violations, mse)
    fin_return = property(lambda self: self.fin_stats[:, 0])
    fin_length = property(lambda self: self.fin_stats[:, 1])
    fin_violation = property(lambda self: self.fin_stats[:, 2])
    fin_mse = property(lambda self: self.fin_stats[:, 3])

    @property
    def truncated(self):
        return (self.flags & L.FLAG_TRUNCATED) != 0

    @property
    def constraint_violation(self):
        return (self.flags & L.FLAG_VIOLATION) != 0

    @property
    def out_of_bounds(self):
        return (self.flags & L.FLAG_OOB) != 0

    @property
    def goal_reached(self):
        return (self.flags & L.FLAG_GOAL) != 0

    @property
    def ground_contact(self):
        """Quadrotors: z at / below the reference world's ground plane (contact is not modelled: see scg_hip.h, bit4)."""
        return (self.flags & L.FLAG_GROUND) != 0


class LazyInfoList(Sequence):
    """``info['n']``: behaves like the reference's tuple of per-env dicts
    (dummy_vec_env.py:41) but copies the columnar device arrays to the host only on first access.
    It is a VIEW of the step's output buffers, which the next step() overwrites (all columns together): read it — or index
    it once, which snapshots every column on the host — before stepping again.  ``step_snapshot``: the ctrl_step_counter
    of every env AT STEP TIME (benchmark_env.py:466); the NumPy-returning facade (which synchronises anyway) passes it, the
    tensor-returning one reads the live counters at first access like every other column."""

    def __init__(self, venv, out, is_reset, step_snapshot=None):
        self._venv, self._out, self._is_reset, self._host = venv, out, is_reset, None
        self._step_snapshot = step_snapshot

    def __len__(self):
        return self._venv.num_envs

    def _fetch(self):
        if self._host is None:
            o = self._out
            h = {k: getattr(o, k).cpu().numpy() for k in ('flags', 'mse', 'done', 'terminal_obs')}
            fin = o.fin_stats.cpu().numpy()
            h.update(fin_return=fin[:, 0], fin_length=fin[:, 1], fin_violation=fin[:, 2], fin_mse=fin[:, 3])
            h['c_values'] = o.c_values.t().cpu().numpy() if o.c_values is not None else None
            if not self._is_reset:      # ctrl_step_counter of the running episodes (benchmark_env.py:466); a reset's is 0 by definition
                h['step'] = self._step_snapshot if self._step_snapshot is not None else self._venv.get_step_counters()
            self._host = h
        return self._host

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        v, h = self._venv, self._fetch()
        spec = v.spec
        if self._is_reset:
            return v._reset_info(h, i, with_constraints=True)
        flags = int(h['flags'][i])
        step = {'current_step': None, 'constraint_violation': int(bool(flags & L.FLAG_VIOLATION)),
                'mse': float(h['mse'][i])}
        if spec.num_constraints_or_zero:
            step['constraint_values'] = h['c_values'][i].astype(np.float64)
        if spec.done_on_out_of_bound:
            step['out_of_bounds'] = bool(flags & L.FLAG_OOB)
        if spec.TASK == 'stabilization' and spec.COST == 'quadratic':
            step['goal_reached'] = bool(flags & L.FLAG_GOAL)
        if flags & L.FLAG_GROUND:           # (extension key, only present when raised: the body is at the unmodelled ground plane)
            step['ground_contact'] = True
        if h['done'][i]:
            step['current_step'] = int(h['fin_length'][i])
            if flags & L.FLAG_TRUNCATED or step['current_step'] >= spec.CTRL_STEPS:
                step['TimeLimit.truncated'] = bool(flags & L.FLAG_TRUNCATED)
            info = v._reset_info(h, i)
            if spec.n_state_con_rows and self._out.state is not None:
                # after_reset of the NEW episode (benchmark_env.py:356-357: state constraints only).  The kernel's c_values row is
                # the finished episode's (-> terminal_info): evaluate the rows on the post-reset state the step returned
                if 'state' not in h:
                    h['state'] = self._out.state.t().cpu()
                info['constraint_values'] = spec.state_constraint_values(h['state'][i:i + 1])[0].numpy()
            info['terminal_observation'] = h['terminal_obs'][i].astype(np.float64)
            info['terminal_info'] = step
            info['episode'] = {'r': float(h['fin_return'][i]), 'l': float(h['fin_length'][i]),
                               'constraint_violation': float(h['fin_violation'][i]), 'mse': float(h['fin_mse'][i])}
            return info
        step['current_step'] = int(h['step'][i])
        return step


class HipVecEnv(VecEnv):
    """N independent copies of one environment stepped by libscg_hip.so on one GPU."""

    def __init__(self, env_id, num_envs, seed=0, device=None, dtype=torch.float32, env_id_offset=0,
                 return_numpy=True, auto_reset=True, specialize='auto', policy=None, **task_config):
        L.lib()                                        # fail loudly, before touching torch.cuda
        if not torch.cuda.is_available():
            raise L.ScgError('HipVecEnv needs a HIP device (torch.cuda.is_available() is False); '
                             'there is no CPU fallback.')
        task_config = dict(task_config)
        cfg_seed = task_config.pop('seed', None)
        if seed is None:
            seed = 0 if cfg_seed is None else cfg_seed
        self.spec = EnvSpec(env_id, task_config)
        spec = self.spec
        spec.num_constraints_or_zero = len(spec.con_rows)
        self.env_id = env_id
        self.device = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
        self.dtype = dtype
        self._cdtype = L.F64 if dtype == torch.float64 else L.F32
        if dtype not in (torch.float32, torch.float64):
            raise ValueError('dtype must be torch.float32 or torch.float64')
        self.seed_value = checked_seed(seed)
        self.env_id_offset = int(env_id_offset)
        self.return_numpy = return_numpy
        VecEnv.__init__(self, int(num_envs), spec.observation_space, spec.action_space)
        cfg, x_goal = spec.to_c_config(self.num_envs, self._cdtype, self.seed_value, self.env_id_offset, auto_reset)
        self.auto_reset = bool(auto_reset)
        # config-specialised library when one was built for this config (specialize=True compiles it now)
        # policy=(hidden, activation): use the variant of the specialised library that also carries the fused
        # policy-in-the-loop rollout kernel for that actor shape (rollout_policy below); float32 only
        self.policy_shape = None
        if policy is not None and dtype == torch.float32 and L.policy_supported(self.spec.obs_dim, int(policy[0]), self.spec.nu, policy[1]) \
                and self.spec.obs_dim in (self.spec.nx, 2 * self.spec.nx):
            self.policy_shape = (int(policy[0]), policy[1])
        self._lib, self.specialized = L.lib_for(cfg, specialize, self.policy_shape)
        self._cfg = cfg
        nbytes = C.c_size_t(0)
        self._chk(self._lib.scg_workspace_bytes(C.byref(cfg), C.byref(nbytes)))
        with torch.cuda.device(self.device):
            self.workspace = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=self.device)
            self._ws_bytes = int(nbytes.value)
            base = self.workspace.data_ptr()
            self._ws_ptr = (base + 255) // 256 * 256
            handle = C.c_void_p()
            self._chk(self._lib.scg_create(C.byref(cfg), x_goal.ctypes.data_as(C.POINTER(C.c_double)),
                                         self.device.index or 0, C.c_void_p(self._ws_ptr), nbytes.value, C.byref(handle)))
        self._h = handle
        N, spec = self.num_envs, self.spec
        # Every default output is carved from ONE arena: the step kernel then reaches all of them through a single
        # buffer resource (include/scg_hip.h, scg_step_out).
        n_rows = len(spec.con_rows)
        esz = torch.empty((), dtype=dtype).element_size()
        shapes = [('obs', (N, spec.obs_dim), dtype), ('reward', (N,), dtype), ('done', (N,), torch.uint8),
                  ('flags', (N,), torch.uint8), ('c_values', (max(n_rows, 1), N), dtype), ('mse', (N,), dtype),
                  ('terminal_obs', (N, spec.obs_dim), dtype), ('state', (spec.nx, N), dtype),
                  ('noisy_action', (spec.nu, N), dtype), ('fin_stats', (N, 4), dtype), ('ep_stats', (N, 4), dtype)]
        offs, total = {}, 0
        for name, shape, dt in shapes:
            offs[name] = total
            nbytes = int(np.prod(shape)) * (1 if dt == torch.uint8 else esz)
            total += (nbytes + 255) // 256 * 256
        self._arena = torch.zeros(total + 256, dtype=torch.uint8, device=self.device)
        base = (-self._arena.data_ptr()) % 256
        view = {}
        for name, shape, dt in shapes:
            nbytes = int(np.prod(shape)) * (1 if dt == torch.uint8 else esz)
            view[name] = self._arena[base + offs[name]: base + offs[name] + nbytes].view(dt).view(*shape)
        o = StepTensors()
        for k in ('obs', 'reward', 'done', 'flags', 'mse', 'terminal_obs', 'state', 'noisy_action', 'fin_stats'):
            setattr(o, k, view[k])
        o.c_values = view['c_values'] if n_rows else None                # [n_con_rows, N]
        self.out = o
        self.ep_stats = view['ep_stats']
        self._c_out = self._make_c_out(o)
        self._actions = None
        self._adv = None
        self.seed_epoch = 0
        self.closed = False

    def _chk(self, rc):
        L.check(rc, self._lib)

    # running totals of the current episodes (columns of the packed accumulator)
    ep_return = property(lambda self: self.ep_stats[:, 0])
    ep_length = property(lambda self: self.ep_stats[:, 1])
    ep_violation = property(lambda self: self.ep_stats[:, 2])
    ep_mse = property(lambda self: self.ep_stats[:, 3])

    # ------------------------------------------------------------------ plumbing
    def _make_c_out(self, o, obs=None):
        s = L.StepOut()
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
        s.d_obs = p(obs if obs is not None else o.obs)
        s.d_reward, s.d_done, s.d_flags = p(o.reward), p(o.done), p(o.flags)
        s.d_c_values, s.d_mse, s.d_terminal_obs = p(o.c_values), p(o.mse), p(o.terminal_obs)
        s.d_state, s.d_noisy_action = p(o.state), p(o.noisy_action)
        s.d_ep_stats, s.d_fin_stats = p(self.ep_stats), p(o.fin_stats)
        return s

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _as_device(self, a, cols):
        t = torch.as_tensor(a, device=self.device).to(self.dtype)
        t = t.reshape(self.num_envs, cols).contiguous()
        return t

    # ------------------------------------------------------------------ fast (tensor) API
    def reset_tensors(self, mask=None):
        """Reset all envs (mask None) or the envs whose mask byte is non-zero.  Returns obs [N, obs_dim]."""
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            self._chk(self._lib.scg_reset(self._h, C.c_void_p(m.data_ptr()) if m is not None else None,
                                        C.byref(self._c_out), self._stream()))
        return self.out.obs

    def step_tensors(self, actions, adv_actions=None, out=None, c_out=None):
        """One control step for every env.  `actions` [N, action_dim] device tensor of the env dtype.
        Pass a prepared (StepTensors, StepOut) pair to write into caller-owned buffers (rollout storage)."""
        a = actions
        if a.dtype != self.dtype or a.device != self.device or not a.is_contiguous():
            a = a.to(device=self.device, dtype=self.dtype).contiguous()
        adv_ptr = None
        if adv_actions is not None:
            adv = adv_actions.to(device=self.device, dtype=self.dtype).contiguous()
            adv_ptr = C.c_void_p(adv.data_ptr())
        with torch.cuda.device(self.device):
            self._chk(self._lib.scg_step(self._h, C.c_void_p(a.data_ptr()), adv_ptr,
                                       C.byref(c_out if c_out is not None else self._c_out), self._stream()))
        return out if out is not None else self.out

    def step_range_tensors(self, first, count, actions, adv_actions=None, out=None, c_out=None):
        """The control step for envs [first, first + count) only (scg_step_range): `actions` (and every output) are the
        full [N, ...] tensors, only that range's rows are read / written.  Disjoint ranges may be advanced from different
        streams concurrently (sub-shard launches: one range's launch latency overlaps another's)."""
        a = actions
        if a.dtype != self.dtype or a.device != self.device or not a.is_contiguous():
            a = a.to(device=self.device, dtype=self.dtype).contiguous()
        adv_ptr = None
        if adv_actions is not None:
            adv = adv_actions.to(device=self.device, dtype=self.dtype).contiguous()
            adv_ptr = C.c_void_p(adv.data_ptr())
        with torch.cuda.device(self.device):
            self._chk(self._lib.scg_step_range(self._h, int(first), int(count), C.c_void_p(a.data_ptr()), adv_ptr,
                                               C.byref(c_out if c_out is not None else self._c_out), self._stream()))
        return out if out is not None else self.out

    def bind_outputs(self, **tensors):
        """A (StepTensors, StepOut) pair whose listed fields point at caller-owned tensors (e.g. slices of a
        rollout buffer) and whose other fields alias this env's default output buffers."""
        o = StepTensors()
        for k in StepTensors.__slots__:
            setattr(o, k, tensors[k] if k in tensors else getattr(self.out, k))    # explicit None = not wanted
        return o, self._make_c_out(o)

    def rollout_random(self, k_steps):
        """K fused control steps with in-kernel U(-1,1) actions (scg_rollout_random).
        Returns (reward_sum [N], done_count [N], violation_count [N], last_obs [N, obs_dim])."""
        N = self.num_envs
        if not hasattr(self, '_ro'):
            self._ro = (torch.zeros(N, dtype=self.dtype, device=self.device),
                        torch.zeros(N, dtype=torch.int32, device=self.device),
                        torch.zeros(N, dtype=torch.int32, device=self.device),
                        torch.zeros(N, self.spec.obs_dim, dtype=self.dtype, device=self.device))
            r = L.RolloutOut()
            r.d_reward_sum, r.d_done_count, r.d_violation_count, r.d_last_obs = (C.c_void_p(t.data_ptr()) for t in self._ro)
            self._ro_c = r
        with torch.cuda.device(self.device):
            self._chk(self._lib.scg_rollout_random(self._h, int(k_steps), C.byref(self._ro_c), self._stream()))
        return self._ro

    def rollout_policy(self, policy, k_steps, obs, act, logp, reward, done, flags, terminal_obs=None, episode_acc=None,
                       max_episodes=0):
        """K control steps in ONE launch with the actor in the loop (scg_rollout_policy): `policy` is an _lib.Policy
        (flat parameter pointer + offsets), the other arguments are the [K(+1), N, .] float32 rollout tensors it fills."""
        if self.policy_shape is None:
            raise L.ScgError('this env was not built with a policy shape (HipVecEnv(..., policy=(hidden, activation)))')
        o = L.PolicyRollout()
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
        o.d_obs, o.d_act, o.d_logp, o.d_reward, o.d_done, o.d_flags = p(obs), p(act), p(logp), p(reward), p(done), p(flags)
        o.d_terminal_obs, o.d_ep_stats, o.d_episode_acc = p(terminal_obs), p(self.ep_stats), p(episode_acc)
        o.max_episodes = int(max_episodes)
        with torch.cuda.device(self.device):
            self._chk(self._lib.scg_rollout_policy(self._h, C.byref(policy), int(k_steps), C.byref(o), self._stream()))

    def step_sequence(self, actions, adv_actions=None, out=None, terminal_obs=True, mse=False, c_values=False, fin_stats=False,
                      state=False, noisy_action=False):
        """K control steps in ONE launch with caller-supplied actions [K, N, action_dim] (scg_step_sequence) — the same
        results as K calls of step_tensors(actions[t]).  Returns (and fills, when passed back as `out`) a dict of
        [K]-stacked tensors: obs [K, N, obs_dim], reward / done / flags [K, N], and on request terminal_obs, mse,
        c_values [K, rows, N], fin_stats [K, N, 4], state [K, nx, N], noisy_action [K, nu, N].  adv_actions: [K, N, adv_dim], already passed through
        set_adversary_control's clip / scale / offset."""
        spec = self.spec
        if actions.dim() != 3 or actions.shape[1:] != (self.num_envs, spec.nu) or actions.dtype != self.dtype or not actions.is_contiguous():
            raise ValueError(f'actions must be a contiguous [K, {self.num_envs}, {spec.nu}] {self.dtype} tensor')
        K = int(actions.shape[0])
        if out is None:
            f = dict(device=self.device, dtype=self.dtype)
            u8 = dict(device=self.device, dtype=torch.uint8)
            out = {'obs': torch.empty(K, self.num_envs, spec.obs_dim, **f), 'reward': torch.empty(K, self.num_envs, **f),
                   'done': torch.empty(K, self.num_envs, **u8), 'flags': torch.empty(K, self.num_envs, **u8)}
            if terminal_obs:
                out['terminal_obs'] = torch.zeros(K, self.num_envs, spec.obs_dim, **f)
            if mse:
                out['mse'] = torch.empty(K, self.num_envs, **f)
            if c_values and len(spec.con_rows):
                out['c_values'] = torch.empty(K, len(spec.con_rows), self.num_envs, **f)
            if fin_stats:
                out['fin_stats'] = torch.zeros(K, self.num_envs, 4, **f)
            if state:
                out['state'] = torch.empty(K, spec.nx, self.num_envs, **f)
            if noisy_action:
                out['noisy_action'] = torch.empty(K, spec.nu, self.num_envs, **f)
        q = L.Sequence()
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
        q.d_actions, q.d_adv_actions = p(actions), p(adv_actions)
        q.d_obs, q.d_reward, q.d_done, q.d_flags = p(out['obs']), p(out['reward']), p(out['done']), p(out['flags'])
        q.d_terminal_obs, q.d_mse, q.d_c_values = p(out.get('terminal_obs')), p(out.get('mse')), p(out.get('c_values'))
        q.d_ep_stats, q.d_fin_stats = p(self.ep_stats), p(out.get('fin_stats'))
        q.d_state, q.d_noisy_action = p(out.get('state')), p(out.get('noisy_action'))
        with torch.cuda.device(self.device):
            self._chk(self._lib.scg_step_sequence(self._h, K, C.byref(q), self._stream()))
        return out

    # ------------------------------------------------------------------ reference VecEnv API
    def reset(self):
        obs = self.reset_tensors()
        info = {'n': LazyInfoList(self, self.out, is_reset=True)}
        return (obs.cpu().numpy().astype(np.float64) if self.return_numpy else obs), info

    def step_async(self, actions):
        self._actions = actions if torch.is_tensor(actions) else self._as_device(np.asarray(actions), self.spec.nu)

    def set_adversary_control(self, actions):
        """Batched BenchmarkEnv.set_adversary_control (benchmark_env.py:216-228)."""
        spec = self.spec
        if spec.adversary_disturbance is None:
            raise RuntimeError('[ERROR] adversary_disturbance does not exist, env.set_adversary_control() cannot be called.')
        a = self._as_device(actions, spec.adversary_dim) if not torch.is_tensor(actions) else actions.to(self.device, self.dtype)
        a = a.clamp(-1.0, 1.0)
        self._adv = a * spec.kw['adversary_disturbance_scale'] + spec.kw['adversary_disturbance_offset']

    def step_wait(self):
        out = self.step_tensors(self._actions, self._adv)
        self._adv = None
        if self.return_numpy:
            info = {'n': LazyInfoList(self, out, is_reset=False, step_snapshot=self.get_step_counters())}
            return (out.obs.cpu().numpy().astype(np.float64), out.reward.cpu().numpy().astype(np.float64),
                    out.done.cpu().numpy().astype(bool), info)
        info = {'n': LazyInfoList(self, out, is_reset=False)}
        return out.obs, out.reward, out.done.bool(), info

    def _reset_info(self, host, i, with_constraints=False):
        spec = self.spec
        info = {'current_step': 0, 'x_reference': spec.X_GOAL, 'u_reference': spec.U_GOAL,
                'physical_parameters': self.physical_parameters(i)}
        if with_constraints and spec.n_state_con_rows and host.get('c_values') is not None:
            # after_reset: state constraints only (benchmark_env.py:356-357)
            info['constraint_values'] = host['c_values'][i][:spec.n_state_con_rows].astype(np.float64)
        return info

    def physical_parameters(self, i):
        p = self.get_params(i, 1)[0]
        if self.spec.name == 'cartpole':
            return {'pole_effective_length': p[0], 'pole_mass': p[2], 'cart_mass': p[1]}    # cartpole.py:706-710
        return {'quadrotor_mass': p[0], 'quadrotor_inertia': [p[1], p[2], p[3]]}

    # ------------------------------------------------------------------ host accessors
    def _host_io(self, fn, width, first, n, data=None):
        buf = np.zeros((n, width), dtype=np.float64) if data is None else np.ascontiguousarray(data, dtype=np.float64).reshape(n, width)
        with torch.cuda.device(self.device):
            self._chk(fn(self._h, buf.ctypes.data_as(C.POINTER(C.c_double)), int(first), int(n), self._stream()))
        return buf

    def get_raw_state(self, first=0, n=None):
        n = self.num_envs - first if n is None else n
        return self._host_io(self._lib.scg_get_state, self._n_state_arrays(), first, n)

    def set_raw_state(self, states, first=0):
        states = np.asarray(states, dtype=np.float64)
        self._host_io(self._lib.scg_set_state, self._n_state_arrays(), first, states.shape[0], states)

    def get_params(self, first=0, n=None):
        n = self.num_envs - first if n is None else n
        return self._host_io(self._lib.scg_get_params, len(self.spec.param_labels), first, n)

    def set_params(self, params, first=0):
        params = np.asarray(params, dtype=np.float64)
        self._host_io(self._lib.scg_set_params, len(self.spec.param_labels), first, params.shape[0], params)

    def get_step_counters(self):
        """ctrl_step_counter of every env (the one array: what step_wait snapshots for info['n'][i]['current_step'])."""
        step = np.empty(self.num_envs, dtype=np.int32)
        with torch.cuda.device(self.device):
            self._chk(self._lib.scg_get_counters(self._h, step.ctypes.data_as(C.POINTER(C.c_int32)), None, 0, self.num_envs, self._stream()))
        return step

    def get_counters(self):
        step = np.zeros(self.num_envs, dtype=np.int32)
        ep = np.zeros(self.num_envs, dtype=np.uint32)
        with torch.cuda.device(self.device):
            self._chk(self._lib.scg_get_counters(self._h, step.ctypes.data_as(C.POINTER(C.c_int32)),
                                               ep.ctypes.data_as(C.POINTER(C.c_uint32)), 0, self.num_envs, self._stream()))
        return step, ep

    def set_step_launch(self, split_max=None, wide_min=None, wsback=None):
        """Tuning knobs of the specialised libraries (scg_set_step_launch / scg_set_step_wsback): which launch geometry scg_step uses
        by shard size — `wide_min`: 256-thread workgroups from this many envs; `wsback` = (min, max): the range of shard sizes whose
        workspace arrays are stored write-back (Quadrotor systems), (1, 0) = never.  None keeps a threshold; (None, 2**31 - 1, (1, 0)) =
        the plain one-wave-per-64-envs launch always.  `split_max` is accepted and ignored (the split launch of rounds 5-6 was removed).
        Results are identical either way."""
        f = lambda v: -1 if v is None else int(v)       # noqa: E731
        self._chk(self._lib.scg_set_step_launch(self._h, f(split_max), f(wide_min)))
        if wsback is not None:
            self._chk(self._lib.scg_set_step_wsback(self._h, int(wsback[0]), int(wsback[1])))

    def set_counters(self, step=None, episode=None):
        sp = None if step is None else np.ascontiguousarray(step, dtype=np.int32)
        ep = None if episode is None else np.ascontiguousarray(episode, dtype=np.uint32)
        with torch.cuda.device(self.device):
            self._chk(self._lib.scg_set_counters(
                self._h, sp.ctypes.data_as(C.POINTER(C.c_int32)) if sp is not None else None,
                ep.ctypes.data_as(C.POINTER(C.c_uint32)) if ep is not None else None, 0, self.num_envs, self._stream()))

    def prior_model(self, x, u, want=('f', 'A', 'B', 'xnext'), eps=None):
        """Batched prior-model services (scg_prior_model): x [n, state_dim], u [n, action_dim] device tensors (physical
        units) -> dict with the requested entries: 'f' [n, nx] continuous-time dynamics, 'A' [n, nx, nx] and 'B' [n, nx, nu]
        Jacobians, 'xnext' [n, nx] one RK4 step of the control period.  n is independent of num_envs."""
        x = torch.as_tensor(x, device=self.device).to(self.dtype).contiguous()
        u = torch.as_tensor(u, device=self.device).to(self.dtype).contiguous()
        n, nx, nu = x.shape[0], self.spec.nx, self.spec.nu
        assert x.shape == (n, nx) and u.shape == (n, nu)
        if eps is None:
            eps = 1e-6 if self.dtype == torch.float64 else 1e-3
        shapes = {'f': (n, nx), 'A': (n, nx, nx), 'B': (n, nx, nu), 'xnext': (n, nx)}
        out = {k: torch.empty(shapes[k], dtype=self.dtype, device=self.device) for k in want}
        p = lambda k: C.c_void_p(out[k].data_ptr()) if k in out else None      # noqa: E731
        with torch.cuda.device(self.device):
            self._chk(self._lib.scg_prior_model(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(u.data_ptr()), n, float(eps),
                                                p('f'), p('A'), p('B'), p('xnext'), self._stream()))
        return out

    def seed(self, seed):
        """New Philox key for every env of the batch (BenchmarkEnv.seed, benchmark_env.py:193-214)."""
        self.seed_value = checked_seed(seed)
        self._chk(self._lib.scg_set_seed(self._h, C.c_uint64(self.seed_value & 0xFFFFFFFFFFFFFFFF)))
        # the key is a kernel argument: HIP graphs captured before this call replay the OLD key — owners of such graphs
        # (ppo.evaluate's cache here, PPO's rollout graph via seed_epoch) re-capture
        self._eval_cache = None
        self.seed_epoch += 1
        return [seed]

    def _n_state_arrays(self):
        return {L.CARTPOLE: 4, L.QUAD_1D: 2, L.QUAD_2D: 6, L.QUAD_3D: 13}[self.spec.system]

    # RNG state = Philox key + per-env (episode, step) counters + the workspace (disturbance offsets);
    # round-trips through checkpoints like dummy_vec_env.py:68-74.
    def get_env_random_state(self):
        step, ep = self.get_counters()
        return [{'seed': self.seed_value, 'env_id_offset': self.env_id_offset, 'step': step, 'episode': ep,
                 'workspace': self._ws_view().cpu(), 'ep_stats': self.ep_stats.cpu(),
                 'rng_layout_version': int(self._lib.scg_rng_layout_version())}]      # (include/scg_hip.h: SCG_RNG_LAYOUT_VERSION)

    def _ws_view(self):
        """The bytes the kernels really use: [_ws_ptr, _ws_ptr + scg_workspace_bytes) — NOT the padded allocation, whose
        alignment slack may differ between the saving and the loading process."""
        off = self._ws_ptr - self.workspace.data_ptr()
        return self.workspace[off: off + self._ws_bytes]

    def set_env_random_state(self, worker_random_states):
        st = worker_random_states[0]
        if st['seed'] != self.seed_value or st['env_id_offset'] != self.env_id_offset:
            raise ValueError('random state belongs to a different seed / env shard')
        have, want = st.get('rng_layout_version', 1), int(self._lib.scg_rng_layout_version())
        if have != want:                # (checkpoints written before round 6 carry no version: rounds 1-4 used layout 1)
            import warnings
            warnings.warn(f'env random state was saved under Philox word layout {have}, this build draws with layout {want} '
                          f'(include/scg_hip.h: SCG_RNG_LAYOUT_VERSION): the resumed run will not reproduce the original one\'s resets')
        ws = st['workspace'].to(self.device)
        if ws.numel() != self._ws_bytes:
            if ws.numel() == self.workspace.numel():            # checkpoint of an older build: whole padded allocation
                self.workspace.copy_(ws)
                ws = None
            else:
                raise ValueError('random state belongs to a different env batch (workspace size)')
        if ws is not None:
            self._ws_view().copy_(ws)
        if 'ep_stats' in st:
            self.ep_stats.copy_(st['ep_stats'].to(self.device))

    # ------------------------------------------------------------------ attribute access
    def get_attr(self, attr_name, indices=None):
        idx = list(self._get_indices(indices))
        if attr_name == 'state':
            return list(self.out.state.t()[idx].cpu().numpy().astype(np.float64))
        val = getattr(self.spec, attr_name) if hasattr(self.spec, attr_name) else getattr(self, attr_name)
        return [val for _ in idx]

    def set_attr(self, attr_name, values, indices=None):
        raise NotImplementedError('per-env attribute mutation is not supported: all envs of a HipVecEnv share one config')

    def env_method(self, method_name, method_args=None, method_kwargs=None, indices=None):
        idx = list(self._get_indices(indices))
        if method_name == 'set_adversary_control':
            if len(idx) != self.num_envs:
                raise NotImplementedError('set_adversary_control must address every env')
            self.set_adversary_control(np.stack([a[0] for a in method_args]))
            return [None] * len(idx)
        raise NotImplementedError(f'env_method({method_name!r})')

    def close_extras(self):
        if getattr(self, '_h', None) is not None and self._h:
            self._lib.scg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:                               # noqa: BLE001
            pass


class GroupedVecEnv(VecEnv):
    """Heterogeneous `env_configs` of make_vec_envs (envs/env_wrappers/vectorized_env/__init__.py:42-66: upstream builds env k as
    `env_func(**env_configs[k])`): the envs are grouped by identical config, every group is ONE HipVecEnv (one kernel launch per
    group and control step), and this wrapper scatters / gathers rows in the caller's env order.  Each env therefore keeps
    its OWN config across auto-resets exactly as upstream's per-env objects do (own init_state, inertial_prop, episode
    length, constraints ...); only the observation / action dimensions must agree.  Reference (NumPy) API; a homogeneous batch
    should use HipVecEnv directly (one launch, device tensors)."""

    def __init__(self, groups):
        """groups: list of (HipVecEnv, indices of its envs in the caller's order)."""
        self.groups = [(env, np.asarray(idx, dtype=np.int64)) for env, idx in groups]
        n = sum(len(idx) for _, idx in self.groups)
        order = np.concatenate([idx for _, idx in self.groups])
        if sorted(order.tolist()) != list(range(n)):
            raise ValueError('group indices must partition range(num_envs)')
        first = self.groups[0][0]
        for env, _ in self.groups[1:]:
            if env.observation_space.shape != first.observation_space.shape or env.action_space.shape != first.action_space.shape:
                raise ValueError('env_configs differ in observation / action dimensions: they cannot share one batch')
        VecEnv.__init__(self, n, first.observation_space, first.action_space)
        self.spec, self.device, self.dtype = first.spec, first.device, first.dtype
        self._where = np.empty((n, 2), dtype=np.int64)                      # env -> (group, row)
        for g, (_, idx) in enumerate(self.groups):
            self._where[idx, 0], self._where[idx, 1] = g, np.arange(len(idx))

    def _merge_info(self, infos):
        merged = [None] * self.num_envs
        for (_, idx), info in zip(self.groups, infos):
            lst = info['n']
            for j, i in enumerate(idx):
                merged[i] = lst[j]
        return {'n': merged}

    def reset(self):
        obs, infos = np.empty((self.num_envs,) + self.observation_space.shape, dtype=np.float64), []
        for env, idx in self.groups:
            o, info = env.reset()
            obs[idx] = o if isinstance(o, np.ndarray) else o.cpu().numpy()
            infos.append(info)
        return obs, self._merge_info(infos)

    def step_async(self, actions):
        actions = np.asarray(actions)
        for env, idx in self.groups:
            env.step_async(actions[idx])

    def step_wait(self):
        N = self.num_envs
        obs = np.empty((N,) + self.observation_space.shape, dtype=np.float64)
        rew, done, infos = np.empty(N, dtype=np.float64), np.empty(N, dtype=bool), []
        for env, idx in self.groups:
            o, r, d, info = env.step_wait()
            conv = (lambda t: t if isinstance(t, np.ndarray) else t.cpu().numpy())
            obs[idx], rew[idx], done[idx] = conv(o), conv(r), conv(d)
            infos.append(info)
        return obs, rew, done, self._merge_info(infos)

    def _by_group(self, indices):
        sel = list(self._get_indices(indices))
        per = {}
        for pos, i in enumerate(sel):
            g, r = self._where[i]
            per.setdefault(int(g), []).append((pos, int(r)))
        return sel, per

    def get_attr(self, attr_name, indices=None):
        sel, per = self._by_group(indices)
        out = [None] * len(sel)
        for g, items in per.items():
            vals = self.groups[g][0].get_attr(attr_name, [r for _, r in items])
            for (pos, _), v in zip(items, vals):
                out[pos] = v
        return out

    def set_attr(self, attr_name, values, indices=None):
        raise NotImplementedError('per-env attribute mutation is not supported (configs are fixed at construction)')

    def env_method(self, method_name, method_args=None, method_kwargs=None, indices=None):
        sel, per = self._by_group(indices)
        out = [None] * len(sel)
        for g, items in per.items():
            env = self.groups[g][0]
            args = [method_args[pos] for pos, _ in items] if method_args is not None else None
            kw = [method_kwargs[pos] for pos, _ in items] if method_kwargs is not None else None
            vals = env.env_method(method_name, args, kw, [r for _, r in items] if len(items) != env.num_envs else None)
            for (pos, _), v in zip(items, vals):
                out[pos] = v
        return out

    def get_env_random_state(self):
        return [env.get_env_random_state()[0] for env, _ in self.groups]

    def set_env_random_state(self, worker_random_states):
        for (env, _), st in zip(self.groups, worker_random_states):
            env.set_env_random_state([st])

    def close_extras(self):
        for env, _ in self.groups:
            env.close()

"""Batched float64 restatement of BenchmarkEnv / CartPole / Quadrotor.  ORACLE — test infra only.

One oracle object simulates ``num_envs`` independent copies of the reference's single
environment; per-env semantics (including RNG draw order) follow, line by line:

* envs/benchmark_env.py: __init__ :54-191, seed :193-214, _randomize_values_by_info
  :237-268, before_reset :320-341, after_reset :343-359, before_step :400-420, extend_obs
  :422-445, after_step :447-502.
* envs/gym_pybullet_drones/quadrotor.py: __init__ :150-326, reset :328-392, step :394-445,
  _set_action_space :606-638, _set_observation_space :640-712, _preprocess_control :722-747,
  (de)normalize_action :749-775, _get_observation :777-817, _get_reward :819-862, _get_done
  :864-894, _get_info :896-923; base_aviary.py :77-131 (constants), :232-286, :364-384;
  quadrotor_utils.py:16-60 (cmd2pwm / pwm2rpm); assets/cf2x.urdf.
* envs/gym_control/cartpole.py: __init__ :125-236, step :238-264, reset :266-352,
  _set_action_space :439-447, _set_observation_space :449-477, _preprocess_control :479-502,
  _advance_simulation :532-583, _get_observation :585-609, _get_reward :611-652, _get_done
  :654-672, _get_info :674-696; assets/cartpole_template.urdf.
* math_and_models/normalization.py:8-10 (normalize_angle).

The physics engine call is replaced by oracle/bullet.py (see there for pinning status).
"""
import copy

import numpy as np

from oracle import bullet
from oracle.constraints import create_constraint_list
from oracle.disturbances import create_disturbance_list
from oracle.rng import (CH_ACTION, CH_DYNAMICS, CH_OBSERVATION, CH_RESET, NumpyEnvRng, PhiloxEnvRng,
                        make_tag, u01_from_word)
from oracle.trajectory import generate_trajectory, transform_trajectory

CHANNEL_OF_MODE = {'action': CH_ACTION, 'dynamics': CH_DYNAMICS, 'observation': CH_OBSERVATION}
# Philox reset-draw groups (must match scg_rng.h): item = group; j = INIT_STATE_LABELS index | inertial parameter
# index | 4 * (channel - 1) + list index for disturbance offsets.  Variable j of a group whose randomised variables all
# need ONE word (uniform / choice) uses the 21-bit field j % 6 of block j // 6 ("compact"); as soon as one of them is a normal draw
# (two words) every variable of the group uses block j // 2 and the word pair (2*(j%2), 2*(j%2)+1); disturbance offsets
# always use the pair layout.
GROUP_INIT, GROUP_INERTIAL, GROUP_DISTURB = 0, 1, 2


def normalize_angle(x):
    """math_and_models/normalization.py:8-10."""
    return ((x + np.pi) % (2 * np.pi)) - np.pi


def get_cost_weight_matrix(weights, dim):
    """controllers/lqr/lqr_utils.py:77-99."""
    weights = np.asarray(weights, dtype=float).reshape(-1)
    if len(weights) == dim:
        return np.diag(weights)
    if len(weights) == 1:
        return np.diag(weights.item() * np.ones(dim))
    raise Exception('Wrong dimension for cost weights.')


# --------------------------------------------------------------------------- #
# Random draws, in the reference's order, on either back-end.
# --------------------------------------------------------------------------- #
class Draws:
    def __init__(self, env, rng):
        self.env = env
        self.rng = rng

    # ---- reset-time scalar draws (benchmark_env.py:257-267; disturbances.py:102,148) ----
    def reset_integer(self, idx, channel, k, bound):
        if self.rng.kind == 'numpy':
            return np.array([self.rng.gens[i].integers(bound) for i in idx], dtype=np.int64)
        j = 4 * (channel - 1) + k
        tag = make_tag(CH_RESET, GROUP_DISTURB, j // 2)
        return self.rng.integer_below(idx, self.env.episode, 0, tag, bound, word=2 * (j % 2))

    def reset_scalar(self, idx, group, j, spec, compact=False):
        """One draw per env in ``idx`` from a {distrib, args, **kwargs} spec (variable j of ``group``)."""
        spec = copy.deepcopy(spec)
        distrib = spec.pop('distrib')
        d_args = spec.pop('args', [])
        if self.rng.kind == 'numpy':
            return np.array([getattr(self.rng.gens[i], distrib)(*d_args, **spec) for i in idx], dtype=np.float64)
        tag = make_tag(CH_RESET, group, j // 6 if compact else j // 2)
        w0 = 2 * (j % 2)
        if compact:
            word = self.rng.compact_field(idx, self.env.episode, 0, tag, j % 6)
        if distrib == 'uniform':
            low = d_args[0] if len(d_args) > 0 else spec.get('low', 0.0)
            high = d_args[1] if len(d_args) > 1 else spec.get('high', 1.0)
            if compact:
                return low + (high - low) * u01_from_word(word)
            return low + (high - low) * self.rng.uniform01(idx, self.env.episode, 0, tag, word=w0)
        if distrib == 'normal':
            loc = d_args[0] if len(d_args) > 0 else spec.get('loc', 0.0)
            scale = d_args[1] if len(d_args) > 1 else spec.get('scale', 1.0)
            return loc + scale * self.rng.normal01_words(idx, self.env.episode, 0, tag, w0, w0 + 1)
        if distrib == 'choice':
            opts = np.asarray(d_args[0] if len(d_args) > 0 else spec['a'], dtype=np.float64)
            if compact:
                return opts[((word.astype(np.uint64) * np.uint64(len(opts))) >> np.uint64(32)).astype(np.int64)]
            return opts[self.rng.integer_below(idx, self.env.episode, 0, tag, len(opts), word=w0)]
        raise NotImplementedError(f'oracle/philox: distribution {distrib}')

    # ---- per-step vector draws (disturbances.py:188,219,253) ----
    def _step_index(self, channel):
        # Philox addressing: observation noise is indexed by the observation's position in the
        # episode (0 at reset, k after the k-th step) so the reset observation and the first step
        # observation do not share a counter; action / dynamics noise by the pre-increment counter.
        env = self.env
        if channel == CH_OBSERVATION and not env._at_reset:
            return env.ctrl_step_counter + 1
        return env.ctrl_step_counter

    def step_uniform(self, channel, k, low, high):
        env, dim = self.env, len(low)
        if self.rng.kind == 'numpy':
            return np.stack([g.uniform(low, high, size=dim) for g in self.rng.gens])
        idx = np.arange(env.num_envs)
        out = np.empty((env.num_envs, dim))
        step = self._step_index(channel)
        for j in range(dim):
            tag = make_tag(channel, k, j // 4)
            out[:, j] = low[j] + (high[j] - low[j]) * self.rng.uniform01(idx, env.episode, step, tag, j % 4)
        return out

    def step_normal(self, channel, k, std):
        env, dim = self.env, len(std)
        if self.rng.kind == 'numpy':
            return np.stack([g.normal(0, std, size=dim) for g in self.rng.gens])
        idx = np.arange(env.num_envs)
        out = np.empty((env.num_envs, dim))
        step = self._step_index(channel)
        for j in range(dim):
            tag = make_tag(channel, k, j // 2)
            out[:, j] = std[j] * self.rng.normal01(idx, env.episode, step, tag, j % 2)
        return out


# --------------------------------------------------------------------------- #
# BenchmarkEnv
# --------------------------------------------------------------------------- #
class OracleBenchmarkEnv:
    NAME = 'base'
    DISTURBANCE_MODES = None
    INERTIAL_PROP_RAND_INFO = None
    INIT_STATE_RAND_INFO = None
    TASK_INFO = None

    def _base_init(self, num_envs, rng, seed=None, normalized_rl_action_space=False,
                   task='stabilization', task_info=None, cost='rl_reward', pyb_freq=50, ctrl_freq=50,
                   episode_len_sec=5, init_state=None, randomized_init=True,
                   init_state_randomization_info=None, prior_prop=None, inertial_prop=None,
                   randomized_inertial_prop=False, inertial_prop_randomization_info=None,
                   constraints=None, done_on_violation=False, use_constraint_penalty=False,
                   constraint_penalty=1.0, disturbances=None, adversary_disturbance=None,
                   adversary_disturbance_offset=0.0, adversary_disturbance_scale=0.01,
                   integrator='pyb_euler', rk4_substeps=1, **unused):
        # benchmark_env.py:125-191
        self.num_envs = num_envs
        # extension (not a reference key): 'rk4' integrates the prior model instead of the PyBullet step (oracle/symbolic.py)
        self.integrator, self.rk4_substeps = integrator, int(rk4_substeps)
        self.TASK = str(getattr(task, 'value', task))
        if task_info is not None:
            self.TASK_INFO = task_info
        self.CTRL_FREQ, self.PYB_FREQ = ctrl_freq, pyb_freq
        if self.PYB_FREQ % self.CTRL_FREQ != 0:
            raise ValueError('[ERROR] in BenchmarkEnv.__init__(), pyb_freq is not divisible by env_freq.')
        self.PYB_STEPS_PER_CTRL = int(self.PYB_FREQ / self.CTRL_FREQ)
        self.CTRL_TIMESTEP = 1. / self.CTRL_FREQ
        self.PYB_TIMESTEP = 1. / self.PYB_FREQ
        self.EPISODE_LEN_SEC = episode_len_sec
        self.CTRL_STEPS = self.EPISODE_LEN_SEC * self.CTRL_FREQ
        self.INIT_STATE = init_state
        self.RANDOMIZED_INIT = randomized_init
        if init_state_randomization_info is not None:
            self.INIT_STATE_RAND_INFO = init_state_randomization_info
        self.INERTIAL_PROP = inertial_prop
        self.RANDOMIZED_INERTIAL_PROP = randomized_inertial_prop
        if inertial_prop_randomization_info is not None:
            self.INERTIAL_PROP_RAND_INFO = inertial_prop_randomization_info
        self.NORMALIZED_RL_ACTION_SPACE = normalized_rl_action_space
        self.COST = str(getattr(cost, 'value', cost))
        self._set_action_space()
        self._set_observation_space()
        self.CONSTRAINTS = constraints
        self.DONE_ON_VIOLATION = done_on_violation
        self.use_constraint_penalty = use_constraint_penalty
        self.constraint_penalty = constraint_penalty
        self.constraints = None
        self.num_constraints = 0
        if constraints is not None:
            self.constraints = create_constraint_list(constraints, self)
            self.num_constraints = self.constraints.num_constraints
        self.DISTURBANCES = disturbances
        self.adversary_disturbance = adversary_disturbance
        self.adversary_disturbance_offset = adversary_disturbance_offset
        self.adversary_disturbance_scale = adversary_disturbance_scale
        self._setup_disturbances()
        self.rng = rng
        self.draws = Draws(self, rng)
        self.adv_action = None
        # batched bookkeeping
        self.ctrl_step_counter = np.zeros(num_envs, dtype=np.int64)
        self.pyb_step_counter = np.zeros(num_envs, dtype=np.int64)
        self.episode = -np.ones(num_envs, dtype=np.int64)      # becomes 0 at the first reset
        self.initial_reset = False
        self._at_reset = False

    def _setup_disturbances(self):
        # benchmark_env.py:279-295
        self.disturbances = {}
        if self.DISTURBANCES is not None:
            for mode, specs in self.DISTURBANCES.items():
                assert mode in self.DISTURBANCE_MODES
                self.disturbances[mode] = create_disturbance_list(
                    specs, self.DISTURBANCE_MODES[mode], self, CHANNEL_OF_MODE[mode])
        if self.adversary_disturbance is not None:
            assert self.adversary_disturbance in self.DISTURBANCE_MODES
            self.adversary_dim = self.DISTURBANCE_MODES[self.adversary_disturbance]['dim']

    def set_adversary_control(self, action):
        """benchmark_env.py:216-228 (batched: action (N, adv_dim))."""
        if self.adversary_disturbance is None:
            raise RuntimeError('[ERROR] adversary_disturbance does not exist, env.set_adversary_control() cannot be called.')
        clipped = np.clip(np.asarray(action, dtype=np.float64), -1.0, 1.0)
        self.adv_action = clipped * self.adversary_disturbance_scale + self.adversary_disturbance_offset

    # benchmark_env.py:237-268 — additive randomisation, keys in ``original_values`` order.
    def _randomize_values_by_info(self, idx, names, base_values, info, group):
        out = {}
        compact = all(info[name]['distrib'] != 'normal' for name in names if name in info)
        for j, name in enumerate(names):
            val = np.full(len(idx), float(base_values[name]))
            if name in info:
                val = val + self.draws.reset_scalar(idx, group, j, info[name], compact)
            out[name] = val
        return out

    # benchmark_env.py:320-341
    def _before_reset(self, idx, seed=None):
        self.initial_reset = True
        self.pyb_step_counter[idx] = 0
        self.ctrl_step_counter[idx] = 0
        self.episode[idx] += 1
        for mode in self.disturbances:
            self.disturbances[mode].reset(self, idx)
        if self.adversary_disturbance is not None:
            self.adv_action = None
        if seed is not None:
            assert self.rng.kind == 'numpy'
            for i in idx:
                self.rng.reseed(i, seed)

    # benchmark_env.py:422-445
    def extend_obs(self, obs, next_step):
        """obs (n, state_dim); next_step (n,) int."""
        if self.COST == 'rl_reward' and self.TASK == 'traj_tracking' and self.obs_goal_horizon > 0:
            last = self.X_GOAL.shape[0] - 1
            wp = np.minimum(next_step[:, None] + np.arange(self.obs_goal_horizon)[None, :], last)
            goal = self.X_GOAL[wp].reshape(obs.shape[0], -1)
            return np.concatenate([obs, goal], axis=1)
        if self.COST == 'rl_reward' and self.TASK == 'stabilization' and self.obs_goal_horizon > 0:
            goal = np.broadcast_to(self.X_GOAL.reshape(1, -1), (obs.shape[0], self.X_GOAL.size))
            return np.concatenate([obs, goal], axis=1)
        return obs

    # benchmark_env.py:447-502
    def _after_step(self, rew, done):
        self.pyb_step_counter += self.PYB_STEPS_PER_CTRL
        self.ctrl_step_counter += 1
        info = {'current_step': self.ctrl_step_counter.copy()}
        violation = np.zeros(self.num_envs, dtype=bool)
        if self.constraints is not None:
            c_value = self.constraints.get_values(self.state, self.current_noisy_physical_action)
            info['constraint_values'] = c_value
            violation = self.constraints.is_violated(c_value)
            if self.DONE_ON_VIOLATION:
                done = done | violation
                if self.COST == 'rl_reward' and self.use_constraint_penalty:
                    rew = np.where(violation, 0.0, rew)
        info['constraint_violation'] = violation.astype(np.int64)
        if self.COST == 'rl_reward' and self.constraints is not None and self.use_constraint_penalty:
            if self.rew_exponential:
                with np.errstate(divide='ignore'):
                    pen = np.exp(np.log(rew) - self.constraint_penalty)
            else:
                pen = rew - self.constraint_penalty
            rew = np.where(violation, pen, rew)
        time_up = self.ctrl_step_counter >= self.CTRL_STEPS
        info['TimeLimit.truncated'] = time_up & ~done          # only defined where time_up (:499-501)
        info['time_limit_reached'] = time_up
        done = done | time_up
        return rew, done, info

    def _stale_oob(self, oob, flags):
        """``self.out_of_bounds`` upstream is an attribute that _get_done only refreshes when it does
        not return early on goal_reached (quadrotor.py:871-892, cartpole.py:661-671) and that reset()
        never clears, so on a goal_reached step _get_info reports the value left by the previous
        evaluation (possibly from the previous episode).  Replicated with a persistent per-env flag."""
        if not hasattr(self, '_oob_attr'):
            self._oob_attr = np.zeros(self.num_envs, dtype=bool)
        if 'goal_reached' in flags:
            oob = np.where(flags['goal_reached'], self._oob_attr, oob)
        self._oob_attr = oob.copy()
        return oob

    # shared reward / info pieces ------------------------------------------------
    def _reference_row(self, offset):
        """X_GOAL row min(ctrl_step_counter + offset, len-1) per env (tracking only)."""
        return self.X_GOAL[np.minimum(self.ctrl_step_counter + offset, self.X_GOAL.shape[0] - 1)]


# --------------------------------------------------------------------------- #
# Quadrotor
# --------------------------------------------------------------------------- #
class OracleQuadrotor(OracleBenchmarkEnv):
    NAME = 'quadrotor'
    # assets/cf2x.urdf
    URDF_MASS, URDF_ARM = 0.027, 0.0397
    URDF_J = (1.4e-5, 1.4e-5, 2.17e-5)
    KF, KM = 3.16e-10, 7.94e-12
    PWM2RPM_SCALE, PWM2RPM_CONST, MIN_PWM, MAX_PWM = 0.2685, 4070.3, 20000.0, 65535.0
    PROP_OFFSET = 0.028                        # cf2x.urdf:42,54,66,78
    GRAVITY_ACC = 9.8                          # base_aviary.py:77
    GROUND_PLANE_Z = -0.05                     # base_aviary.py:107

    BASE_INERTIAL_PROP_RAND_INFO = {           # quadrotor.py:47-68
        'M': {'distrib': 'uniform', 'low': 0.022, 'high': 0.032},
        'Ixx': {'distrib': 'uniform', 'low': 1.3e-5, 'high': 1.5e-5},
        'Iyy': {'distrib': 'uniform', 'low': 1.3e-5, 'high': 1.5e-5},
        'Izz': {'distrib': 'uniform', 'low': 2.07e-5, 'high': 2.27e-5}}
    BASE_INIT_STATE_RAND_INFO = {              # quadrotor.py:70-136
        'init_x': {'distrib': 'uniform', 'low': -0.5, 'high': 0.5},
        'init_x_dot': {'distrib': 'uniform', 'low': -0.01, 'high': 0.01},
        'init_y': {'distrib': 'uniform', 'low': -0.5, 'high': 0.5},
        'init_y_dot': {'distrib': 'uniform', 'low': -0.01, 'high': 0.01},
        'init_z': {'distrib': 'uniform', 'low': 0.1, 'high': 1.5},
        'init_z_dot': {'distrib': 'uniform', 'low': -0.01, 'high': 0.01},
        'init_phi': {'distrib': 'uniform', 'low': -0.3, 'high': 0.3},
        'init_theta': {'distrib': 'uniform', 'low': -0.3, 'high': 0.3},
        'init_psi': {'distrib': 'uniform', 'low': -0.3, 'high': 0.3},
        'init_p': {'distrib': 'uniform', 'low': -0.01, 'high': 0.01},
        'init_theta_dot': {'distrib': 'uniform', 'low': -0.01, 'high': 0.01},
        'init_q': {'distrib': 'uniform', 'low': -0.01, 'high': 0.01},
        'init_r': {'distrib': 'uniform', 'low': -0.01, 'high': 0.01}}
    TASK_INFO = {'stabilization_goal': [0, 1], 'stabilization_goal_tolerance': 0.05,
                 'trajectory_type': 'circle', 'num_cycles': 1, 'trajectory_plane': 'zx',
                 'trajectory_position_offset': [0.5, 0], 'trajectory_scale': -0.5,
                 'proj_point': [0, 0, 0.5], 'proj_normal': [0, 1, 1]}
    INIT_STATE_LABELS = {
        1: ['init_x', 'init_x_dot'],
        2: ['init_x', 'init_x_dot', 'init_z', 'init_z_dot', 'init_theta', 'init_theta_dot'],
        3: ['init_x', 'init_x_dot', 'init_y', 'init_y_dot', 'init_z', 'init_z_dot',
            'init_phi', 'init_theta', 'init_psi', 'init_p', 'init_q', 'init_r']}
    INERTIAL_NAMES = ['M', 'Ixx', 'Iyy', 'Izz']

    def __init__(self, num_envs, rng, init_state=None, inertial_prop=None, quad_type=2,
                 norm_act_scale=0.1, obs_goal_horizon=0, rew_state_weight=1.0, rew_act_weight=0.0001,
                 rew_exponential=True, done_on_out_of_bound=True, info_mse_metric_state_weight=None,
                 respect_randomization_info=False, physics='pyb', **kwargs):
        assert str(getattr(physics, 'value', physics)) == 'pyb', 'oracle restates Physics.PYB only'
        self.QUAD_TYPE = int(quad_type)
        self.norm_act_scale = norm_act_scale
        self.obs_goal_horizon = obs_goal_horizon
        self.rew_state_weight = np.array(rew_state_weight, ndmin=1, dtype=float)
        self.rew_act_weight = np.array(rew_act_weight, ndmin=1, dtype=float)
        self.rew_exponential = rew_exponential
        self.done_on_out_of_bound = done_on_out_of_bound
        nx = {1: 2, 2: 6, 3: 12}[self.QUAD_TYPE]
        if info_mse_metric_state_weight is None:       # quadrotor.py:186-194
            w = {1: [1, 0], 2: [1, 0, 1, 0, 0, 0], 3: [1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0]}[self.QUAD_TYPE]
        else:
            if len(info_mse_metric_state_weight) != nx:
                raise ValueError('[ERROR] in Quadrotor.__init__(), wrong info_mse_metric_state_weight argument size.')
            w = info_mse_metric_state_weight
        self.info_mse_metric_state_weight = np.array(w, ndmin=1, dtype=float)
        self.MASS, self.L = self.URDF_MASS, self.URDF_ARM
        self.J = np.array(self.URDF_J, dtype=float)
        self.DISTURBANCE_MODES = {'observation': {'dim': -1}, 'action': {'dim': -1}, 'dynamics': {'dim': -1}}
        self._base_init(num_envs, rng, init_state=init_state, inertial_prop=inertial_prop, **kwargs)
        user_init_info = kwargs.get('init_state_randomization_info')
        user_inertial_info = kwargs.get('inertial_prop_randomization_info')
        # quadrotor.py:208 — the class re-installs BASE_INIT_STATE_RAND_INFO *after*
        # BenchmarkEnv.__init__ stored the user's dict, i.e. the YAML key
        # ``init_state_randomization_info`` is ignored upstream (same for the inertial info, :233).
        # ``respect_randomization_info=True`` is a non-reference extension that honours the YAML.
        self.INIT_STATE_RAND_INFO = copy.deepcopy(self.BASE_INIT_STATE_RAND_INFO)
        if respect_randomization_info and user_init_info is not None:
            self.INIT_STATE_RAND_INFO = copy.deepcopy(user_init_info)
        labels = self.INIT_STATE_LABELS[self.QUAD_TYPE]
        self.init_values = {}
        if init_state is None:
            for name in labels:
                self.init_values[name] = 0.
        elif isinstance(init_state, np.ndarray):
            for i, name in enumerate(labels):
                self.init_values[name] = init_state[i]
        elif isinstance(init_state, dict):
            for name in labels:
                self.init_values[name] = init_state.get(name, 0.)
        else:
            raise ValueError('[ERROR] in Quadrotor.__init__(), init_state incorrect format.')
        for name in list(self.INIT_STATE_RAND_INFO.keys()):     # :229-231
            if name not in labels:
                self.INIT_STATE_RAND_INFO.pop(name, None)
        self.INERTIAL_PROP_RAND_INFO = copy.deepcopy(self.BASE_INERTIAL_PROP_RAND_INFO)   # :233
        if respect_randomization_info and user_inertial_info is not None:
            self.INERTIAL_PROP_RAND_INFO = copy.deepcopy(user_inertial_info)
        if self.QUAD_TYPE == 1:
            for k in ('Ixx', 'Iyy', 'Izz'):
                self.INERTIAL_PROP_RAND_INFO.pop(k, None)
        elif self.QUAD_TYPE == 2:
            for k in ('Ixx', 'Izz'):
                self.INERTIAL_PROP_RAND_INFO.pop(k, None)
        # :244-259
        if inertial_prop is None:
            pass
        elif self.QUAD_TYPE == 1 and np.array(inertial_prop).shape == (1,):
            self.MASS = inertial_prop[0]
        elif self.QUAD_TYPE == 2 and np.array(inertial_prop).shape == (2,):
            self.MASS, self.J[1] = inertial_prop
        elif self.QUAD_TYPE == 3 and np.array(inertial_prop).shape == (4,):
            self.MASS, self.J[0], self.J[1], self.J[2] = inertial_prop
        elif isinstance(inertial_prop, dict):
            self.MASS = inertial_prop.get('M', self.MASS)
            self.J[0] = inertial_prop.get('Ixx', self.J[0])
            self.J[1] = inertial_prop.get('Iyy', self.J[1])
            self.J[2] = inertial_prop.get('Izz', self.J[2])
        else:
            raise ValueError('[ERROR] in Quadrotor.__init__(), inertial_prop incorrect format.')
        # :262-323
        self.U_GOAL = np.ones(self.action_dim) * self.MASS * self.GRAVITY_ACC / self.action_dim
        ti = self.TASK_INFO
        if self.TASK == 'stabilization':
            g = ti['stabilization_goal']
            if self.QUAD_TYPE == 1:
                self.X_GOAL = np.hstack([g[1], 0.0])
            elif self.QUAD_TYPE == 2:
                self.X_GOAL = np.hstack([g[0], 0.0, g[1], 0.0, 0.0, 0.0])
            else:
                self.X_GOAL = np.hstack([g[0], 0.0, g[1], 0.0, g[2], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
            self.X_GOAL = np.asarray(self.X_GOAL, dtype=float)
        elif self.TASK == 'traj_tracking':
            P, V, _ = generate_trajectory(traj_type=ti['trajectory_type'], traj_length=self.EPISODE_LEN_SEC,
                                          num_cycles=ti['num_cycles'], traj_plane=ti['trajectory_plane'],
                                          position_offset=ti['trajectory_position_offset'],
                                          scaling=ti['trajectory_scale'], sample_time=self.CTRL_TIMESTEP)
            z = np.zeros(P.shape[0])
            if self.QUAD_TYPE == 1:
                self.X_GOAL = np.vstack([P[:, 2], V[:, 2]]).T
            elif self.QUAD_TYPE == 2:
                self.X_GOAL = np.vstack([P[:, 0], V[:, 0], P[:, 2], V[:, 2], z, z]).T
            else:
                PT, VT = transform_trajectory(P, V, ti['proj_point'], ti['proj_normal'])
                self.X_GOAL = np.vstack([PT[:, 0], VT[:, 0], PT[:, 1], VT[:, 1], PT[:, 2], VT[:, 2],
                                         z, z, z, z, z, z]).T
        else:
            raise ValueError(self.TASK)
        self.Q = get_cost_weight_matrix(self.rew_state_weight, nx)
        self.R = get_cost_weight_matrix(self.rew_act_weight, self.action_dim)
        N = num_envs
        self.pos, self.quat = np.zeros((N, 3)), np.tile([0., 0., 0., 1.], (N, 1))
        self.vel, self.ang_v, self.rpy = np.zeros((N, 3)), np.zeros((N, 3)), np.zeros((N, 3))
        self.mass_env = np.full(N, float(self.MASS))
        self.J_env = np.tile(self.J, (N, 1))
        self.state = np.zeros((N, nx))

    # quadrotor.py:606-638
    def _set_action_space(self):
        self.action_dim = {1: 1, 2: 2, 3: 4}[self.QUAD_TYPE]
        n_mot = 4 / self.action_dim
        a_low = self.KF * n_mot * (self.PWM2RPM_SCALE * self.MIN_PWM + self.PWM2RPM_CONST) ** 2
        a_high = self.KF * n_mot * (self.PWM2RPM_SCALE * self.MAX_PWM + self.PWM2RPM_CONST) ** 2
        self.physical_action_bounds = (np.full(self.action_dim, a_low, np.float32),
                                       np.full(self.action_dim, a_high, np.float32))
        if self.NORMALIZED_RL_ACTION_SPACE:
            # NB: uses the URDF mass — inertial_prop overrides are applied later (:244-259).
            self.hover_thrust = self.GRAVITY_ACC * self.MASS / self.action_dim
            self.action_space_low = -np.ones(self.action_dim, dtype=np.float32)
            self.action_space_high = np.ones(self.action_dim, dtype=np.float32)
        else:
            self.action_space_low, self.action_space_high = self.physical_action_bounds

    # quadrotor.py:640-712
    def _set_observation_space(self):
        xt, xd, zt, zd = 2, 30, 2, 30
        ang = 85 * np.pi / 180
        psi = 180 * np.pi / 180
        rate = 500 * np.pi / 180
        gz = self.GROUND_PLANE_Z
        if self.QUAD_TYPE == 1:
            low = np.array([gz, -zd]); high = np.array([zt, zd])
        elif self.QUAD_TYPE == 2:
            low = np.array([-xt, -xd, gz, -zd, -ang, -rate]); high = np.array([xt, xd, zt, zd, ang, rate])
        else:
            low = np.array([-xt, -xd, -xt, -xd, gz, -zd, -ang, -ang, -psi, -rate, -rate, -rate])
            high = np.array([xt, xd, xt, xd, zt, zd, ang, ang, psi, rate, rate, rate])
        self.state_space_low = low.astype(np.float32)
        self.state_space_high = high.astype(np.float32)
        self.state_dim = low.shape[0]
        if self.COST == 'rl_reward' and self.TASK == 'traj_tracking' and self.obs_goal_horizon > 0:
            mul = 1 + self.obs_goal_horizon
        elif self.COST == 'rl_reward' and self.TASK == 'stabilization' and self.obs_goal_horizon > 0:
            mul = 2
        else:
            mul = 1
        self.observation_space_low = np.concatenate([low] * mul).astype(np.float32)
        self.observation_space_high = np.concatenate([high] * mul).astype(np.float32)
        self.obs_dim = self.observation_space_low.shape[0]

    def _setup_disturbances(self):
        # quadrotor.py:714-720
        self.DISTURBANCE_MODES['observation']['dim'] = self.obs_dim
        self.DISTURBANCE_MODES['action']['dim'] = self.action_dim
        self.DISTURBANCE_MODES['dynamics']['dim'] = int(self.QUAD_TYPE)
        super()._setup_disturbances()

    # quadrotor.py:328-392
    def reset(self, idx=None, seed=None):
        idx = np.arange(self.num_envs) if idx is None else np.asarray(idx)
        self._before_reset(idx, seed)
        base = {'M': self.MASS, 'Ixx': self.J[0], 'Iyy': self.J[1], 'Izz': self.J[2]}
        if self.RANDOMIZED_INERTIAL_PROP:
            prop = self._randomize_values_by_info(idx, self.INERTIAL_NAMES, base,
                                                  self.INERTIAL_PROP_RAND_INFO, GROUP_INERTIAL)
            if any(np.any(v < 0) for v in prop.values()):
                raise ValueError('[ERROR] in Quadrotor.reset(), negative randomized inertial properties.')
        else:
            prop = {k: np.full(len(idx), float(v)) for k, v in base.items()}
        self.mass_env[idx] = prop['M']
        self.J_env[idx] = np.stack([prop['Ixx'], prop['Iyy'], prop['Izz']], axis=1)
        labels = self.INIT_STATE_LABELS[self.QUAD_TYPE]
        if self.RANDOMIZED_INIT:
            iv = self._randomize_values_by_info(idx, labels, self.init_values,
                                                self.INIT_STATE_RAND_INFO, GROUP_INIT)
        else:
            iv = {k: np.full(len(idx), float(self.init_values[k])) for k in labels}
        zero = np.zeros(len(idx))
        get = lambda k: iv.get(k, zero)
        xyz = np.stack([get('init_x'), get('init_y'), get('init_z')], axis=1)
        vel = np.stack([get('init_x_dot'), get('init_y_dot'), get('init_z_dot')], axis=1)
        rpy = np.stack([get('init_phi'), get('init_theta'), get('init_psi')], axis=1)
        if self.QUAD_TYPE == 2:
            ang_v = np.stack([zero, get('init_theta_dot'), zero], axis=1)
        else:
            ang_v = np.stack([get('init_p'), get('init_q'), get('init_r')], axis=1)   # world frame (:379)
        self.pos[idx] = xyz
        self.quat[idx] = bullet.quaternion_from_euler(rpy)
        self.vel[idx] = vel
        self.ang_v[idx] = ang_v
        self.rpy[idx] = bullet.euler_from_quaternion(self.quat[idx])
        self._update_state(idx)
        self._at_reset = True
        obs = self._get_observation(idx, at_reset=True)
        self._at_reset = False
        info = {'current_step': np.zeros(len(idx), dtype=np.int64),
                'physical_parameters': {'quadrotor_mass': self.mass_env[idx].copy(),
                                        'quadrotor_inertia': self.J_env[idx].copy()}}
        if self.constraints is not None and self.constraints.state_constraints:
            info['constraint_values'] = self.constraints.get_values(self.state[idx], None, only_state=True)
        return obs, info

    def set_body_state(self, idx, pos, quat, vel, ang_v):
        """Parity-test hook: overwrite the rigid-body state (host-injected initial states)."""
        self.pos[idx], self.quat[idx], self.vel[idx], self.ang_v[idx] = pos, quat, vel, ang_v
        self.rpy[idx] = bullet.euler_from_quaternion(self.quat[idx])
        self._update_state(idx)

    def _update_state(self, idx):
        # quadrotor.py:784-802
        pos, vel, rpy, ang_v = self.pos[idx], self.vel[idx], self.rpy[idx], self.ang_v[idx]
        if self.QUAD_TYPE == 1:
            s = np.stack([pos[:, 2], vel[:, 2]], axis=1)
        elif self.QUAD_TYPE == 2:
            s = np.stack([pos[:, 0], vel[:, 0], pos[:, 2], vel[:, 2], rpy[:, 1], ang_v[:, 1]], axis=1)
        else:
            R = bullet.matrix_from_quaternion(self.quat[idx])
            body = np.einsum('nji,nj->ni', R, ang_v)
            s = np.concatenate([np.stack([pos[:, 0], vel[:, 0], pos[:, 1], vel[:, 1], pos[:, 2], vel[:, 2]], axis=1),
                                rpy, body], axis=1)
        self.state[idx] = s

    # quadrotor.py:777-817
    def _get_observation(self, idx, at_reset):
        obs = self.state[idx].copy()
        if 'observation' in self.disturbances:
            obs = self._apply_obs_disturbance(obs, idx)
        nxt = np.ones(len(idx), dtype=np.int64) if at_reset else self.ctrl_step_counter[idx] + 2
        return self.extend_obs(obs, nxt)

    def _apply_obs_disturbance(self, obs, idx):
        # The observation disturbance list has dim = obs_dim (quadrotor.py:717) while it is
        # applied to the state-sized vector BEFORE the goal extension (:805-807): upstream this
        # only works when obs_dim == state_dim (no goal horizon) or with masks/broadcastable noise.
        if len(idx) == self.num_envs:
            return self.disturbances['observation'].apply(obs, self)
        # reset of a subset: evaluate for everybody, keep the subset (numpy mode consumes only
        # the subset's generators).
        return _subset_apply(self, 'observation', obs, idx)

    # quadrotor.py:722-775 + quadrotor_utils.py:16-60
    def _preprocess_control(self, action):
        if self.NORMALIZED_RL_ACTION_SPACE:
            action = (1 + self.norm_act_scale * action) * self.hover_thrust
        self.current_physical_action = action
        if 'action' in self.disturbances:
            action = self.disturbances['action'].apply(action, self)
        if self.adversary_disturbance == 'action':
            action = action + self.adv_action
        self.current_noisy_physical_action = action
        thrust = np.clip(action, self.physical_action_bounds[0], self.physical_action_bounds[1])
        self.current_clipped_action = thrust
        n_motor = 4 // self.action_dim
        thrust = np.clip(thrust, 0.0, None)
        pwm = (np.sqrt(thrust / n_motor / self.KF) - self.PWM2RPM_CONST) / self.PWM2RPM_SCALE
        if self.action_dim == 1:
            pwm = np.repeat(pwm, 4, axis=1)
        elif self.action_dim == 2:
            pwm = np.concatenate([pwm, pwm[:, ::-1]], axis=1)
        pwm = np.clip(pwm, self.MIN_PWM, self.MAX_PWM)
        return self.PWM2RPM_SCALE * pwm + self.PWM2RPM_CONST

    # quadrotor.py:394-445
    def step(self, action):
        assert self.initial_reset, '[ERROR] You must call env.reset() at least once before using env.step().'
        action = np.asarray(action, dtype=np.float64).reshape(self.num_envs, self.action_dim)
        self.current_raw_action = action
        rpm = self._preprocess_control(action)
        disturb = None
        passive = 'dynamics' in self.disturbances
        adv = self.adversary_disturbance == 'dynamics'
        if passive or adv:
            disturb = np.zeros((self.num_envs, self.QUAD_TYPE))
        if passive:
            disturb = self.disturbances['dynamics'].apply(disturb, self)
        if adv and self.adv_action is not None:
            disturb = disturb + self.adv_action
            self.adv_action = None
        if disturb is not None:
            z = np.zeros(self.num_envs)
            if self.QUAD_TYPE == 1:
                disturb = np.stack([z, z, disturb[:, 0]], axis=1)
            elif self.QUAD_TYPE == 2:
                disturb = np.stack([disturb[:, 0], z, disturb[:, 1]], axis=1)
        # base_aviary.py:232-286, :364-384
        forces = rpm ** 2 * self.KF
        torques = rpm ** 2 * self.KM
        z_torque = -torques[:, 0] + torques[:, 1] - torques[:, 2] + torques[:, 3]
        pos, quat, vel, ang_v = self.pos, self.quat, self.vel, self.ang_v
        if self.integrator == 'rk4':
            assert disturb is None, 'the rk4 mode integrates the disturbance-free prior model'
            pos, quat, vel, ang_v = self._rk4_step()
        else:
            # base_aviary.py:272 — posObj is the position cached at the START of the control step.
            dist_point = None if disturb is None else self.pos.copy()
            for _ in range(self.PYB_STEPS_PER_CTRL):
                pos, quat, vel, ang_v = bullet.quadrotor_substep(
                    pos, quat, vel, ang_v, forces, z_torque, disturb, self.mass_env, self.J_env,
                    self.PROP_OFFSET, self.GRAVITY_ACC, self.PYB_TIMESTEP, dist_point)
        self.pos, self.quat, self.vel, self.ang_v = pos, quat, vel, ang_v
        self.rpy = bullet.euler_from_quaternion(quat)
        all_idx = np.arange(self.num_envs)
        self._update_state(all_idx)
        obs = self._get_observation(all_idx, at_reset=False)
        rew = self._get_reward()
        done, flags = self._get_done()
        info_step = self._get_info(flags)
        rew, done, info = self._after_step(rew, done)
        info.update(info_step)
        return obs, rew, done, info

    def _rk4_step(self):
        """One control period of the prior model (quadrotor.py:485-563) with the clipped thrusts as input."""
        from oracle import symbolic
        u = self.current_clipped_action
        h = self.CTRL_TIMESTEP / self.rk4_substeps
        arm, g, m = self.L / np.sqrt(2.0), self.GRAVITY_ACC, self.mass_env
        x = self.state.copy()
        N = self.num_envs
        z = np.zeros(N)
        if self.QUAD_TYPE == 1:
            x = symbolic.rk4(lambda a, b: symbolic.f_quad1d(a, b, m, g), x, u, h, self.rk4_substeps)
            pos = np.stack([self.pos[:, 0], self.pos[:, 1], x[:, 0]], axis=1)
            vel = np.stack([z, z, x[:, 1]], axis=1)
            return pos, self.quat, vel, self.ang_v
        if self.QUAD_TYPE == 2:
            x = symbolic.rk4(lambda a, b: symbolic.f_quad2d(a, b, m, self.J_env[:, 1], arm, g), x, u, h, self.rk4_substeps)
            pos = np.stack([x[:, 0], z, x[:, 2]], axis=1)
            vel = np.stack([x[:, 1], z, x[:, 3]], axis=1)
            quat = bullet.quaternion_from_euler(np.stack([z, x[:, 4], z], axis=1))
            return pos, quat, vel, np.stack([z, x[:, 5], z], axis=1)
        x = symbolic.rk4(lambda a, b: symbolic.f_quad3d(a, b, m, self.J_env, arm, self.KM / self.KF, g), x, u, h, self.rk4_substeps)
        pos, vel = x[:, [0, 2, 4]], x[:, [1, 3, 5]]
        quat = bullet.quaternion_from_euler(x[:, 6:9])
        R = bullet.matrix_from_quaternion(quat)
        return pos, quat, vel, np.einsum('nij,nj->ni', R, x[:, 9:12])

    # quadrotor.py:819-862
    def _get_reward(self):
        if self.COST == 'rl_reward':
            act_error = self.current_noisy_physical_action - self.U_GOAL
            if self.TASK == 'stabilization':
                err = self.state - self.X_GOAL
            else:
                err = self.state - self._reference_row(1)
            dist = np.sum(self.rew_state_weight * err * err, axis=1)
            dist = dist + np.sum(self.rew_act_weight * act_error * act_error, axis=1)
            rew = -dist
            return np.exp(rew) if self.rew_exponential else rew
        # quadratic: -(1/2 e'Qe + 1/2 du'R du) with the CLIPPED action (:848-862)
        if self.TASK == 'stabilization':
            err = self.state - self.X_GOAL
        else:
            err = self.state - self.X_GOAL[self.ctrl_step_counter + 1]      # no clamp upstream (:858)
        du = self.current_clipped_action - self.U_GOAL
        return -(0.5 * np.einsum('ni,ij,nj->n', err, self.Q, err) + 0.5 * np.einsum('ni,ij,nj->n', du, self.R, du))

    # quadrotor.py:864-894
    def _get_done(self):
        flags = {}
        done = np.zeros(self.num_envs, dtype=bool)
        if self.TASK == 'stabilization':
            goal = np.linalg.norm(self.state - self.X_GOAL, axis=1) < self.TASK_INFO['stabilization_goal_tolerance']
            flags['goal_reached'] = goal
            done |= goal
        if self.done_on_out_of_bound:
            mask = {1: [1, 0], 2: [1, 0, 1, 0, 1, 0], 3: [1, 0, 1, 0, 1, 0, 1, 1, 1, 0, 0, 0]}[self.QUAD_TYPE]
            oob = (self.state < self.state_space_low) | (self.state > self.state_space_high)
            oob = np.any(oob & np.array(mask, dtype=bool), axis=1)
            oob = self._stale_oob(oob, flags)
            flags['out_of_bounds'] = oob
            done |= oob & ~flags.get('goal_reached', np.zeros(self.num_envs, dtype=bool))
        return done, flags

    # quadrotor.py:896-923
    def _get_info(self, flags):
        info = {}
        if self.TASK == 'stabilization' and self.COST == 'quadratic':
            info['goal_reached'] = flags['goal_reached']
        if self.done_on_out_of_bound:
            info['out_of_bounds'] = flags['out_of_bounds']
        state = self.state.copy()
        if self.TASK == 'stabilization':
            err = state - self.X_GOAL
        else:
            if self.QUAD_TYPE == 2:
                state[:, 4] = normalize_angle(state[:, 4])
            elif self.QUAD_TYPE == 3:
                state[:, 6:9] = normalize_angle(state[:, 6:9])
            err = state - self._reference_row(1)
        err = err * self.info_mse_metric_state_weight
        info['mse'] = np.sum(err ** 2, axis=1)
        return info


def _subset_apply(env, mode, target, idx):
    """Apply a disturbance list to a subset of envs (used by partial resets)."""
    full = np.zeros((env.num_envs, target.shape[1]))
    full[idx] = target
    if env.rng.kind == 'numpy':
        gens = env.rng.gens
        env.rng.gens = [gens[i] for i in idx]
        n = env.num_envs
        sub_counters = (env.ctrl_step_counter, env.pyb_step_counter)
        env.num_envs = len(idx)
        env.ctrl_step_counter, env.pyb_step_counter = sub_counters[0][idx], sub_counters[1][idx]
        saved = []
        for d in env.disturbances[mode].disturbances:
            st = {k: getattr(d, k) for k in ('current_step_offset', 'current_peak_step') if hasattr(d, k)}
            saved.append(st)
            for k, v in st.items():
                setattr(d, k, v[idx])
        try:
            out = env.disturbances[mode].apply(target, env)
        finally:
            env.rng.gens = gens
            env.num_envs = n
            env.ctrl_step_counter, env.pyb_step_counter = sub_counters
            for d, st in zip(env.disturbances[mode].disturbances, saved):
                for k, v in st.items():
                    setattr(d, k, v)
        return out
    return env.disturbances[mode].apply(full, env)[idx]


# --------------------------------------------------------------------------- #
# CartPole
# --------------------------------------------------------------------------- #
class OracleCartPole(OracleBenchmarkEnv):
    NAME = 'cartpole'
    # assets/cartpole_template.urdf:37,52,61 (half pole length, pole mass, cart mass)
    URDF_POLE_LENGTH, URDF_POLE_MASS, URDF_CART_MASS = 0.5, 0.1, 1.0
    GRAVITY_ACC = 9.8
    INERTIAL_PROP_RAND_INFO = {                # cartpole.py:75-90
        'pole_length': {'distrib': 'choice', 'args': [[1, 5, 10]]},
        'cart_mass': {'distrib': 'uniform', 'low': 0.5, 'high': 1.5},
        'pole_mass': {'distrib': 'uniform', 'low': 0.05, 'high': 0.15}}
    INIT_STATE_RAND_INFO = {                   # cartpole.py:92-113
        'init_x': {'distrib': 'uniform', 'low': -0.05, 'high': 0.05},
        'init_x_dot': {'distrib': 'uniform', 'low': -0.05, 'high': 0.05},
        'init_theta': {'distrib': 'uniform', 'low': -0.05, 'high': 0.05},
        'init_theta_dot': {'distrib': 'uniform', 'low': -0.05, 'high': 0.05}}
    TASK_INFO = {'stabilization_goal': [0], 'stabilization_goal_tolerance': 0.05,
                 'trajectory_type': 'circle', 'num_cycles': 1, 'trajectory_plane': 'zx',
                 'trajectory_position_offset': [0, 0], 'trajectory_scale': 0.2}
    INIT_NAMES = ['init_x', 'init_x_dot', 'init_theta', 'init_theta_dot']
    INERTIAL_NAMES = ['pole_length', 'cart_mass', 'pole_mass']

    def __init__(self, num_envs, rng, init_state=None, inertial_prop=None, obs_goal_horizon=0,
                 obs_wrap_angle=False, rew_state_weight=1.0, rew_act_weight=0.0001, rew_exponential=True,
                 done_on_out_of_bound=True, info_mse_metric_state_weight=None, pole_inertia_mode='box',
                 **kwargs):
        self.obs_goal_horizon = obs_goal_horizon
        self.obs_wrap_angle = obs_wrap_angle
        self.rew_state_weight = np.array(rew_state_weight, ndmin=1, dtype=float)
        self.rew_act_weight = np.array(rew_act_weight, ndmin=1, dtype=float)
        self.Q = get_cost_weight_matrix(self.rew_state_weight, 4)
        self.R = get_cost_weight_matrix(self.rew_act_weight, 1)
        self.rew_exponential = rew_exponential
        self.done_on_out_of_bound = done_on_out_of_bound
        self.pole_inertia_mode = pole_inertia_mode
        if info_mse_metric_state_weight is None:
            self.info_mse_metric_state_weight = np.array([1, 0, 1, 0], ndmin=1, dtype=float)
        elif len(info_mse_metric_state_weight) == 4:
            self.info_mse_metric_state_weight = np.array(info_mse_metric_state_weight, ndmin=1, dtype=float)
        else:
            raise ValueError('[ERROR] in CartPole.__init__(), wrong info_mse_metric_state_weight argument size.')
        self.DISTURBANCE_MODES = {'observation': {'dim': 4}, 'action': {'dim': 1}, 'dynamics': {'dim': 2}}
        self._base_init(num_envs, rng, init_state=init_state, inertial_prop=inertial_prop, **kwargs)
        if init_state is None:
            vals = np.zeros(4)
        elif isinstance(init_state, np.ndarray):
            vals = init_state
        elif isinstance(init_state, dict):
            vals = [init_state.get(k, 0) for k in self.INIT_NAMES]
        else:
            raise ValueError('[ERROR] in CartPole.__init__(), init_state incorrect format.')
        self.init_values = dict(zip(self.INIT_NAMES, vals))
        if inertial_prop is None:
            self.EFFECTIVE_POLE_LENGTH, self.POLE_MASS, self.CART_MASS = \
                self.URDF_POLE_LENGTH, self.URDF_POLE_MASS, self.URDF_CART_MASS
        elif isinstance(inertial_prop, dict):
            self.EFFECTIVE_POLE_LENGTH = inertial_prop.get('pole_length', self.URDF_POLE_LENGTH)
            self.POLE_MASS = inertial_prop.get('pole_mass', self.URDF_POLE_MASS)
            self.CART_MASS = inertial_prop.get('cart_mass', self.URDF_CART_MASS)
        else:
            raise ValueError('[ERROR] in CartPole.__init__(), inertial_prop incorrect format.')
        self.U_GOAL = np.zeros(1)
        ti = self.TASK_INFO
        if self.TASK == 'stabilization':
            self.X_GOAL = np.hstack([ti['stabilization_goal'][0], 0., 0., 0.]).astype(float)
        elif self.TASK == 'traj_tracking':
            P, V, _ = generate_trajectory(traj_type=ti['trajectory_type'], traj_length=self.EPISODE_LEN_SEC,
                                          num_cycles=ti['num_cycles'], traj_plane=ti['trajectory_plane'],
                                          position_offset=np.array(ti['trajectory_position_offset']),
                                          scaling=ti['trajectory_scale'], sample_time=self.CTRL_TIMESTEP)
            z = np.zeros(P.shape[0])
            self.X_GOAL = np.vstack([P[:, 0], V[:, 0], z, z]).T
        N = num_envs
        self.state = np.zeros((N, 4))
        self.pole_length_env = np.full(N, float(self.EFFECTIVE_POLE_LENGTH))
        self.cart_mass_env = np.full(N, float(self.CART_MASS))
        self.pole_mass_env = np.full(N, float(self.POLE_MASS))

    # cartpole.py:439-447
    def _set_action_space(self):
        self.action_scale = 10
        self.action_dim = 1
        self.physical_action_bounds = (-1 * np.atleast_1d(self.action_scale), np.atleast_1d(self.action_scale))
        self.action_threshold = 1 if self.NORMALIZED_RL_ACTION_SPACE else self.action_scale
        self.action_space_low = np.full(1, -self.action_threshold, dtype=np.float32)
        self.action_space_high = np.full(1, self.action_threshold, dtype=np.float32)

    # cartpole.py:449-477
    def _set_observation_space(self):
        self.x_threshold = 2.4
        self.theta_threshold_radians = 90 * np.pi / 180
        bound = np.array([self.x_threshold * 2, 20, self.theta_threshold_radians * 2, 20])
        self.state_space_low = (-bound).astype(np.float32)
        self.state_space_high = bound.astype(np.float32)
        self.state_dim = 4
        if self.COST == 'rl_reward' and self.TASK == 'traj_tracking' and self.obs_goal_horizon > 0:
            mul = 1 + self.obs_goal_horizon
        elif self.COST == 'rl_reward' and self.TASK == 'stabilization' and self.obs_goal_horizon > 0:
            mul = 2
        else:
            mul = 1
        ob = np.concatenate([bound] * mul)
        self.observation_space_low = (-ob).astype(np.float32)
        self.observation_space_high = ob.astype(np.float32)
        self.obs_dim = ob.shape[0]

    # cartpole.py:266-352
    def reset(self, idx=None, seed=None):
        idx = np.arange(self.num_envs) if idx is None else np.asarray(idx)
        self._before_reset(idx, seed)
        base = {'pole_length': self.EFFECTIVE_POLE_LENGTH, 'cart_mass': self.CART_MASS, 'pole_mass': self.POLE_MASS}
        if self.RANDOMIZED_INERTIAL_PROP:
            prop = self._randomize_values_by_info(idx, self.INERTIAL_NAMES, base,
                                                  self.INERTIAL_PROP_RAND_INFO, GROUP_INERTIAL)
            if any(np.any(v < 0) for v in prop.values()):
                raise ValueError('[ERROR] in CartPole.reset(), negative randomized inertial properties.')
        else:
            prop = {k: np.full(len(idx), float(v)) for k, v in base.items()}
        self.pole_length_env[idx] = prop['pole_length']
        self.cart_mass_env[idx] = prop['cart_mass']
        self.pole_mass_env[idx] = prop['pole_mass']
        if self.RANDOMIZED_INIT:
            iv = self._randomize_values_by_info(idx, self.INIT_NAMES, self.init_values,
                                                self.INIT_STATE_RAND_INFO, GROUP_INIT)
        else:
            iv = {k: np.full(len(idx), float(self.init_values[k])) for k in self.INIT_NAMES}
        self.state[idx] = np.stack([iv[k] for k in self.INIT_NAMES], axis=1)
        self._at_reset = True
        obs = self._get_observation(idx, at_reset=True)
        self._at_reset = False
        info = {'current_step': np.zeros(len(idx), dtype=np.int64),
                'physical_parameters': {'pole_effective_length': self.pole_length_env[idx].copy(),
                                        'pole_mass': self.pole_mass_env[idx].copy(),
                                        'cart_mass': self.cart_mass_env[idx].copy()}}
        if self.constraints is not None and self.constraints.state_constraints:
            info['constraint_values'] = self.constraints.get_values(self.state[idx], None, only_state=True)
        return obs, info

    def set_body_state(self, idx, state):
        self.state[idx] = state

    # cartpole.py:585-609
    def _get_observation(self, idx, at_reset):
        obs = self.state[idx].copy()
        if 'observation' in self.disturbances:
            if len(idx) == self.num_envs:
                obs = self.disturbances['observation'].apply(obs, self)
            else:
                obs = _subset_apply(self, 'observation', obs, idx)
        if self.obs_wrap_angle:
            obs[:, 2] = normalize_angle(obs[:, 2])
        nxt = np.ones(len(idx), dtype=np.int64) if at_reset else self.ctrl_step_counter[idx] + 2
        return self.extend_obs(obs, nxt)

    # cartpole.py:479-530
    def _preprocess_control(self, action):
        if self.NORMALIZED_RL_ACTION_SPACE:
            action = self.action_scale * action
        self.current_physical_action = action
        if 'action' in self.disturbances:
            action = self.disturbances['action'].apply(action, self)
        if self.adversary_disturbance == 'action' and self.adv_action is not None:
            action = action + self.adv_action
        self.current_noisy_physical_action = action
        force = np.clip(action, self.physical_action_bounds[0], self.physical_action_bounds[1])
        self.current_clipped_action = force
        return force[:, 0]

    # cartpole.py:238-264, :532-583
    def step(self, action):
        assert self.initial_reset, '[ERROR] You must call env.reset() at least once before using env.step().'
        action = np.asarray(action, dtype=np.float64).reshape(self.num_envs, 1)
        self.current_raw_action = action
        force = self._preprocess_control(action)
        tab = None
        passive = 'dynamics' in self.disturbances
        adv = self.adversary_disturbance == 'dynamics'
        if passive or adv:
            tab = np.zeros((self.num_envs, 2))
        if passive:
            tab = self.disturbances['dynamics'].apply(tab, self)
        if adv and self.adv_action is not None:
            tab = tab + self.adv_action
            self.adv_action = None
        x, xd, th, thd = (self.state[:, i] for i in range(4))
        if self.integrator == 'rk4':
            from oracle import symbolic
            assert tab is None, 'the rk4 mode integrates the disturbance-free prior model'
            xs = symbolic.rk4(lambda a, b: symbolic.f_cartpole(a, b, self.pole_length_env, self.cart_mass_env,
                                                               self.pole_mass_env, self.GRAVITY_ACC),
                              self.state.copy(), force.reshape(self.num_envs, 1), self.CTRL_TIMESTEP / self.rk4_substeps,
                              self.rk4_substeps)
            x, xd, th, thd = (xs[:, i] for i in range(4))
        else:
            ip = bullet.pole_inertia(self.pole_mass_env, self.pole_length_env, self.pole_inertia_mode)
            for _ in range(self.PYB_STEPS_PER_CTRL):
                x, xd, th, thd = bullet.cartpole_substep(x, xd, th, thd, force, tab, self.cart_mass_env,
                                                         self.pole_mass_env, self.pole_length_env, ip,
                                                         self.GRAVITY_ACC, self.PYB_TIMESTEP)
        self.state = np.stack([x, xd, th, thd], axis=1)
        all_idx = np.arange(self.num_envs)
        obs = self._get_observation(all_idx, at_reset=False)
        rew = self._get_reward()
        done, flags = self._get_done()
        info_step = self._get_info(flags)
        rew, done, info = self._after_step(rew, done)
        info.update(info_step)
        return obs, rew, done, info

    # cartpole.py:611-652
    def _get_reward(self):
        if self.COST == 'rl_reward':
            state = self.state.copy()
            state[:, 2] = normalize_angle(state[:, 2])
            act = self.current_noisy_physical_action
            if self.TASK == 'stabilization':
                err = state - self.X_GOAL
            else:
                err = state - self._reference_row(1)
            dist = np.sum(self.rew_state_weight * err * err, axis=1)
            dist = dist + np.sum(self.rew_act_weight * act * act, axis=1)
            rew = -dist
            return np.exp(rew) if self.rew_exponential else rew
        if self.TASK == 'stabilization':
            err = self.state - self.X_GOAL
        else:
            err = self.state - self.X_GOAL[self.ctrl_step_counter]          # index c, not c+1 (:648)
        du = self.current_clipped_action - self.U_GOAL
        return -(0.5 * np.einsum('ni,ij,nj->n', err, self.Q, err) + 0.5 * np.einsum('ni,ij,nj->n', du, self.R, du))

    # cartpole.py:654-672
    def _get_done(self):
        flags = {}
        done = np.zeros(self.num_envs, dtype=bool)
        if self.TASK == 'stabilization':
            goal = np.linalg.norm(self.state - self.X_GOAL, axis=1) < self.TASK_INFO['stabilization_goal_tolerance']
            flags['goal_reached'] = goal
            done |= goal
        if self.done_on_out_of_bound:
            x, th = self.state[:, 0], self.state[:, 2]
            oob = (x < -self.x_threshold) | (x > self.x_threshold) | \
                  (th < -self.theta_threshold_radians) | (th > self.theta_threshold_radians)
            oob = self._stale_oob(oob, flags)
            flags['out_of_bounds'] = oob
            done |= oob & ~flags.get('goal_reached', np.zeros(self.num_envs, dtype=bool))
        return done, flags

    # cartpole.py:674-696
    def _get_info(self, flags):
        info = {}
        if self.TASK == 'stabilization' and self.COST == 'quadratic':
            info['goal_reached'] = flags['goal_reached']
        if self.done_on_out_of_bound:
            info['out_of_bounds'] = flags['out_of_bounds']
        state = self.state.copy()
        if self.TASK == 'stabilization':
            err = state - self.X_GOAL
        else:
            state[:, 2] = normalize_angle(state[:, 2])
            err = state - self._reference_row(1)
        err = err * self.info_mse_metric_state_weight
        info['mse'] = np.sum(err ** 2, axis=1)
        return info


def make_oracle_env(env_id, num_envs, rng, **config):
    """Counterpart of utils/registration.py:123-125 ``make(idx, **task_config)`` for the oracle."""
    if env_id == 'cartpole':
        return OracleCartPole(num_envs, rng, **config)
    if env_id == 'quadrotor':
        return OracleQuadrotor(num_envs, rng, **config)
    raise KeyError(env_id)


def make_rng(kind, num_envs, seed, rank_offset=0):
    if kind == 'numpy':
        return NumpyEnvRng([None if seed is None else seed + rank_offset + i for i in range(num_envs)])
    return PhiloxEnvRng(seed, rank_offset + np.arange(num_envs))

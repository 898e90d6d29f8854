"""YAML task_config -> EnvSpec -> scg_config.

Host-side counterpart of the constructors of the reference environments: it accepts the SAME
keyword arguments (the YAML keys of /root/reference/safe_control_gym/envs/gym_control/cartpole.yaml
and envs/gym_pybullet_drones/quadrotor.yaml, i.e. the signatures at benchmark_env.py:54-87,
cartpole.py:125-137, quadrotor.py:150-163) and derives everything the kernels need as plain
numbers: spaces, X_GOAL / U_GOAL, action pre-processing constants, flattened constraint rows,
disturbance tables, randomisation specs.
"""
import math
from dataclasses import dataclass, field

import numpy as np

from safe_control_gym_amd import _lib as L
from safe_control_gym_amd.spaces import Box
from safe_control_gym_amd.trajectory import planar_reference, project_on_plane

# ---- robot constants (assets/cf2x.urdf, assets/cartpole_template.urdf of the reference) ----
CF2X = dict(mass=0.027, arm=0.0397, J=(1.4e-5, 1.4e-5, 2.17e-5), kf=3.16e-10, km=7.94e-12,
            pwm2rpm_scale=0.2685, pwm2rpm_const=4070.3, pwm_min=20000.0, pwm_max=65535.0,
            prop_offset=0.028)                 # cf2x.urdf:5,11-12,42-78
CARTPOLE_URDF = dict(pole_length=0.5, pole_mass=0.1, cart_mass=1.0, pole_box_width=0.05)  # :37,52,61,63
GRAVITY = 9.8                                  # base_aviary.py:77, cartpole.py:200
GROUND_PLANE_Z = -0.05                         # base_aviary.py:107
BULLET_MAX_COORDINATE_VELOCITY = 100.0

QUAD_INIT_LABELS = {
    1: ['init_x', 'init_x_dot'],
    2: ['init_x', 'init_x_dot', 'init_z', 'init_z_dot', 'init_theta', 'init_theta_dot'],
    3: ['init_x', 'init_x_dot', 'init_y', 'init_y_dot', 'init_z', 'init_z_dot',
        'init_phi', 'init_theta', 'init_psi', 'init_p', 'init_q', 'init_r']}
CARTPOLE_INIT_LABELS = ['init_x', 'init_x_dot', 'init_theta', 'init_theta_dot']
QUAD_PARAM_LABELS = ['M', 'Ixx', 'Iyy', 'Izz']
CARTPOLE_PARAM_LABELS = ['pole_length', 'cart_mass', 'pole_mass']


def _u(lo, hi):
    return {'distrib': 'uniform', 'low': lo, 'high': hi}


# class-level defaults of the reference (quadrotor.py:47-136, cartpole.py:75-113)
QUAD_BASE_INERTIAL_RAND = {'M': _u(0.022, 0.032), 'Ixx': _u(1.3e-5, 1.5e-5), 'Iyy': _u(1.3e-5, 1.5e-5),
                           'Izz': _u(2.07e-5, 2.27e-5)}
QUAD_BASE_INIT_RAND = {'init_x': _u(-0.5, 0.5), 'init_x_dot': _u(-0.01, 0.01), 'init_y': _u(-0.5, 0.5),
                       'init_y_dot': _u(-0.01, 0.01), 'init_z': _u(0.1, 1.5), 'init_z_dot': _u(-0.01, 0.01),
                       'init_phi': _u(-0.3, 0.3), 'init_theta': _u(-0.3, 0.3), 'init_psi': _u(-0.3, 0.3),
                       'init_p': _u(-0.01, 0.01), 'init_theta_dot': _u(-0.01, 0.01), 'init_q': _u(-0.01, 0.01),
                       'init_r': _u(-0.01, 0.01)}
CARTPOLE_INERTIAL_RAND = {'pole_length': {'distrib': 'choice', 'args': [[1, 5, 10]]},
                          'cart_mass': _u(0.5, 1.5), 'pole_mass': _u(0.05, 0.15)}
CARTPOLE_INIT_RAND = {k: _u(-0.05, 0.05) for k in CARTPOLE_INIT_LABELS}
QUAD_TASK_INFO = {'stabilization_goal': [0, 1], 'stabilization_goal_tolerance': 0.05,
                  'trajectory_type': 'circle', 'num_cycles': 1, 'trajectory_plane': 'zx',
                  'trajectory_position_offset': [0.5, 0], 'trajectory_scale': -0.5,
                  'proj_point': [0, 0, 0.5], 'proj_normal': [0, 1, 1]}
CARTPOLE_TASK_INFO = {'stabilization_goal': [0], 'stabilization_goal_tolerance': 0.05,
                      'trajectory_type': 'circle', 'num_cycles': 1, 'trajectory_plane': 'zx',
                      'trajectory_position_offset': [0, 0], 'trajectory_scale': 0.2}

BASE_DEFAULTS = dict(                         # benchmark_env.py:54-87
    output_dir=None, seed=None, gui=False, verbose=False, normalized_rl_action_space=False,
    task='stabilization', task_info=None, cost='rl_reward', pyb_freq=50, ctrl_freq=50, episode_len_sec=5,
    init_state=None, randomized_init=True, init_state_randomization_info=None, prior_prop=None,
    inertial_prop=None, randomized_inertial_prop=False, inertial_prop_randomization_info=None,
    constraints=None, done_on_violation=False, use_constraint_penalty=False, constraint_penalty=1.0,
    disturbances=None, adversary_disturbance=None, adversary_disturbance_offset=0.0,
    adversary_disturbance_scale=0.01)
QUAD_DEFAULTS = dict(quad_type=2, norm_act_scale=0.1, obs_goal_horizon=0, rew_state_weight=1.0,
                     rew_act_weight=0.0001, rew_exponential=True, done_on_out_of_bound=True,
                     info_mse_metric_state_weight=None, physics='pyb', drone_model='cf2x', record=False)
CARTPOLE_DEFAULTS = dict(obs_goal_horizon=0, obs_wrap_angle=False, rew_state_weight=1.0, rew_act_weight=0.0001,
                         rew_exponential=True, done_on_out_of_bound=True, info_mse_metric_state_weight=None)
# extensions that are NOT reference keys (documented in DESIGN.md)
EXTENSION_DEFAULTS = dict(respect_randomization_info=False, pole_inertia='box', engine_arm=None, integrator='pyb_euler',
                          rk4_substeps=1)


def _enum_str(v):
    return str(getattr(v, 'value', v)).lower()


def _diag_weights(w, dim, what):
    w = np.array(w, ndmin=1, dtype=float)
    if w.size == dim:
        return w
    if w.size == 1:
        return np.full(dim, w.item())
    raise Exception(f'Wrong dimension for cost weights ({what}).')


class ConstraintInfo(dict):
    """One entry of the `constraints:` list after compilation.  A dict (form / var / first_row / n_rows / strict, what the host side of
    this package reads) that also answers the attribute reads of the reference's `Constraint` objects (constraints.py:21-87, 186-231,
    234-283): `constrained_variable`, `dim`, `num_constraints`, `strict`, `decimals`, `constraint_filter`, `A` / `b` or `P` / `b`, and the
    symbolic form `sym_func` / `get_symbolic_model()` — upstream's own lambdas, which evaluate on NumPy vectors as well as on CasADi
    symbols (a SymmetricStateConstraint keeps the 2n-row linear form of the BoundedConstraint it derives from, :436-444, while its VALUES
    are the n rows |x| - bound)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    @property
    def sym_func(self):
        if 'P' in self:
            return lambda x: x.T @ self.constraint_filter.T @ self.P @ self.constraint_filter @ x - self.b
        return lambda x: self.A @ self.constraint_filter @ x - self.b

    def get_symbolic_model(self):
        return self.sym_func

    # values / flags of THIS constraint on the single-env facade (constraints.py:97-165): sliced out of the rows the step kernel evaluated
    def get_value(self, env):
        return np.asarray(env.constraints.get_values(env))[self.first_row:self.first_row + self.n_rows]

    def is_violated(self, env, c_value=None):
        c = self.get_value(env) if c_value is None else np.asarray(c_value)
        return bool(np.any(np.greater_equal(c, 0.) if self.strict else np.greater(c, 0.)))

    def is_almost_active(self, env, c_value=None):
        if self.get('tolerance') is None:
            return False
        c = self.get_value(env) if c_value is None else np.asarray(c_value)
        return bool(np.any(np.greater(c + self['tolerance'], 0.)))


@dataclass
class EnvSpec:
    """Everything derived from the YAML config for one environment type (shared by all N copies)."""
    name: str
    kwargs: dict
    system: int = 0
    quad_type: int = 0
    nx: int = 0
    nu: int = 0
    obs_dim: int = 0
    state_labels: list = field(default_factory=list)
    state_units: list = field(default_factory=list)
    action_labels: list = field(default_factory=list)
    action_units: list = field(default_factory=list)

    # ------------------------------------------------------------------ construction
    def __post_init__(self):
        kw = dict(BASE_DEFAULTS)
        kw.update(CARTPOLE_DEFAULTS if self.name == 'cartpole' else QUAD_DEFAULTS)
        kw.update(EXTENSION_DEFAULTS)
        # BenchmarkEnv.__init__ swallows unknown keys through **kwargs (benchmark_env.py:86), e.g. the
        # `physics: pyb` line the cartpole example overrides carry; same here.
        self.ignored_keys = sorted(set(self.kwargs) - set(kw))
        kw.update(self.kwargs)
        self.kw = kw
        self.TASK = _enum_str(kw['task'])
        self.COST = _enum_str(kw['cost'])
        if self.TASK not in ('stabilization', 'traj_tracking'):
            raise ValueError(f'{self.TASK!r} is not a valid Task')
        if self.COST not in ('rl_reward', 'quadratic'):
            raise ValueError(f'{self.COST!r} is not a valid Cost')
        self.CTRL_FREQ, self.PYB_FREQ = kw['ctrl_freq'], kw['pyb_freq']
        if self.PYB_FREQ % self.CTRL_FREQ != 0:
            raise ValueError('[ERROR] in BenchmarkEnv.__init__(), pyb_freq is not divisible by env_freq.')
        self.PYB_STEPS_PER_CTRL = int(self.PYB_FREQ / self.CTRL_FREQ)
        self.CTRL_TIMESTEP, self.PYB_TIMESTEP = 1. / self.CTRL_FREQ, 1. / self.PYB_FREQ
        self.EPISODE_LEN_SEC = kw['episode_len_sec']
        self.CTRL_STEPS = self.EPISODE_LEN_SEC * self.CTRL_FREQ
        self.max_episode_steps = int(math.ceil(self.CTRL_STEPS))      # first counter value with counter >= CTRL_STEPS
        self.NORMALIZED_RL_ACTION_SPACE = bool(kw['normalized_rl_action_space'])
        self.obs_goal_horizon = int(kw['obs_goal_horizon'])
        if self.obs_goal_horizon > L.MAX_GOAL_HORIZON:
            raise ValueError(f'obs_goal_horizon > {L.MAX_GOAL_HORIZON} is not supported by the kernels')
        self.rew_exponential = bool(kw['rew_exponential'])
        self.done_on_out_of_bound = bool(kw['done_on_out_of_bound'])
        self.GRAVITY_ACC = GRAVITY
        if self.name == 'cartpole':
            self._init_cartpole()
        else:
            self._init_quadrotor()
        self._compile_constraints()
        self._compile_disturbances()

    # ---- shared helpers
    def _obs_multiplier(self):
        if self.COST == 'rl_reward' and self.obs_goal_horizon > 0:
            return 1 + self.obs_goal_horizon if self.TASK == 'traj_tracking' else 2
        return 1

    def _trajectory(self, task_info, offset):
        return planar_reference(task_info['trajectory_type'], self.EPISODE_LEN_SEC, task_info['num_cycles'],
                                task_info['trajectory_plane'], offset, task_info['trajectory_scale'],
                                self.CTRL_TIMESTEP)

    # ------------------------------------------------------------------ cartpole
    def _init_cartpole(self):
        kw = self.kw
        self.system, self.nx, self.nu = L.CARTPOLE, 4, 1
        self.state_labels, self.state_units = ['x', 'x_dot', 'theta', 'theta_dot'], ['m', 'm/s', 'rad', 'rad/s']
        self.action_labels = ['U']
        self.action_units = ['-'] if self.NORMALIZED_RL_ACTION_SPACE else ['N']
        self.TASK_INFO = kw['task_info'] if kw['task_info'] is not None else dict(CARTPOLE_TASK_INFO)
        self.obs_wrap_angle = bool(kw['obs_wrap_angle'])
        self.rew_state_weight = np.array(kw['rew_state_weight'], ndmin=1, dtype=float)
        self.rew_act_weight = np.array(kw['rew_act_weight'], ndmin=1, dtype=float)
        self.Q = np.diag(_diag_weights(self.rew_state_weight, 4, 'state'))
        self.R = np.diag(_diag_weights(self.rew_act_weight, 1, 'input'))
        w = kw['info_mse_metric_state_weight']
        if w is None:
            w = [1, 0, 1, 0]
        elif len(w) != 4:
            raise ValueError('[ERROR] in CartPole.__init__(), wrong info_mse_metric_state_weight argument size.')
        self.info_mse_metric_state_weight = np.array(w, ndmin=1, dtype=float)
        # action space (cartpole.py:439-447)
        self.action_scale = 10
        self.physical_action_bounds = (-1 * np.atleast_1d(self.action_scale), np.atleast_1d(self.action_scale))
        thr = 1 if self.NORMALIZED_RL_ACTION_SPACE else self.action_scale
        self.action_space = Box(low=-thr, high=thr, shape=(1,))
        self.hover_thrust = 0.0
        # observation / state space (cartpole.py:449-477)
        self.x_threshold, self.theta_threshold_radians = 2.4, 90 * math.pi / 180
        bound = np.array([self.x_threshold * 2, 20, self.theta_threshold_radians * 2, 20])
        self.state_space = Box(low=-bound, high=bound, dtype=np.float32)
        ob = np.concatenate([bound] * self._obs_multiplier())
        self.observation_space = Box(low=-ob, high=ob, dtype=np.float32)
        self.obs_dim = ob.shape[0]
        # initial state / inertial parameters
        init = kw['init_state']
        if init is None:
            vals = np.zeros(4)
        elif isinstance(init, np.ndarray):
            vals = init
        elif isinstance(init, dict):
            vals = [init.get(k, 0) for k in CARTPOLE_INIT_LABELS]
        else:
            raise ValueError('[ERROR] in CartPole.__init__(), init_state incorrect format.')
        self.init_labels = CARTPOLE_INIT_LABELS
        self.init_values = [float(v) for v in vals]
        self.init_rand_info = kw['init_state_randomization_info'] or CARTPOLE_INIT_RAND
        ip = kw['inertial_prop']
        u = CARTPOLE_URDF
        if ip is None:
            ip = {}
        elif not isinstance(ip, dict):
            raise ValueError('[ERROR] in CartPole.__init__(), inertial_prop incorrect format.')
        self.param_labels = CARTPOLE_PARAM_LABELS
        self.param_values = [float(ip.get('pole_length', u['pole_length'])), float(ip.get('cart_mass', u['cart_mass'])),
                             float(ip.get('pole_mass', u['pole_mass']))]
        self.EFFECTIVE_POLE_LENGTH, self.CART_MASS, self.POLE_MASS = self.param_values
        self.param_rand_info = kw['inertial_prop_randomization_info'] or CARTPOLE_INERTIAL_RAND
        if kw['pole_inertia'] not in ('box', 'rod'):
            raise ValueError("pole_inertia must be 'box' (PyBullet) or 'rod' (URDF / CasADi prior)")
        self.pole_box_width = u['pole_box_width'] if kw['pole_inertia'] == 'box' else 0.0
        # references (cartpole.py:215-233)
        self.U_GOAL = np.zeros(1)
        ti = self.TASK_INFO
        if self.TASK == 'stabilization':
            self.X_GOAL = np.hstack([ti['stabilization_goal'][0], 0., 0., 0.]).astype(float)
        else:
            pos, vel = self._trajectory(ti, np.array(ti['trajectory_position_offset']))
            z = np.zeros(pos.shape[0])
            self.X_GOAL = np.vstack([pos[:, 0], vel[:, 0], z, z]).T
        self.goal_tolerance = float(ti.get('stabilization_goal_tolerance', 0.0)) if self.TASK == 'stabilization' else 0.0
        self.dyn_dim = 2
        self.engine_arm = 0.0

    # ------------------------------------------------------------------ quadrotor
    def _init_quadrotor(self):
        kw = self.kw
        if _enum_str(kw['physics']) not in ('pyb', 'dyn', 'pyb_gnd', 'pyb_drag', 'pyb_dw', 'pyb_gnd_drag_dw'):
            raise ValueError(f"'{kw['physics']}' is not a valid Physics")                              # base_aviary.py:32-40, quadrotor.py:…Physics(physics)
        if _enum_str(kw['physics']) != 'pyb':
            # base_aviary.py:32-40: only Physics.PYB is used by the shipped configs (DYN is broken upstream).
            raise NotImplementedError("only physics='pyb' is implemented by the HIP kernels")
        qt = int(kw['quad_type'])
        if qt not in (1, 2, 3):
            raise ValueError('[ERROR] in Quadrotor.__init__(), not implemented quad type.')
        self.quad_type = qt
        self.system = {1: L.QUAD_1D, 2: L.QUAD_2D, 3: L.QUAD_3D}[qt]
        self.nx, self.nu = {1: 2, 2: 6, 3: 12}[qt], {1: 1, 2: 2, 3: 4}[qt]
        self.TASK_INFO = kw['task_info'] if kw['task_info'] is not None else dict(QUAD_TASK_INFO)
        self.norm_act_scale = kw['norm_act_scale']
        self.rew_state_weight = np.array(kw['rew_state_weight'], ndmin=1, dtype=float)
        self.rew_act_weight = np.array(kw['rew_act_weight'], ndmin=1, dtype=float)
        w = kw['info_mse_metric_state_weight']
        if w is None:
            w = {1: [1, 0], 2: [1, 0, 1, 0, 0, 0], 3: [1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0]}[qt]
        elif len(w) != self.nx:
            raise ValueError('[ERROR] in Quadrotor.__init__(), wrong info_mse_metric_state_weight argument size.')
        self.info_mse_metric_state_weight = np.array(w, ndmin=1, dtype=float)
        c = CF2X
        self.MASS, self.L = c['mass'], c['arm']
        self.J = np.diag(c['J']).astype(float)
        self.KF, self.KM = c['kf'], c['km']
        self.PWM2RPM_SCALE, self.PWM2RPM_CONST = c['pwm2rpm_scale'], c['pwm2rpm_const']
        self.MIN_PWM, self.MAX_PWM = c['pwm_min'], c['pwm_max']
        self.GROUND_PLANE_Z = GROUND_PLANE_Z
        # action space (quadrotor.py:606-638); hover_thrust uses the URDF mass (inertial_prop is applied later)
        n_mot = 4 / self.nu
        a_low = self.KF * n_mot * (self.PWM2RPM_SCALE * self.MIN_PWM + self.PWM2RPM_CONST) ** 2
        a_high = self.KF * n_mot * (self.PWM2RPM_SCALE * self.MAX_PWM + self.PWM2RPM_CONST) ** 2
        self.physical_action_bounds = (np.full(self.nu, a_low, np.float32), np.full(self.nu, a_high, np.float32))
        self.action_labels = {1: ['T'], 2: ['T1', 'T2'], 3: ['T1', 'T2', 'T3', 'T4']}[qt]
        if self.NORMALIZED_RL_ACTION_SPACE:
            self.hover_thrust = self.GRAVITY_ACC * self.MASS / self.nu
            self.action_space = Box(low=-np.ones(self.nu), high=np.ones(self.nu), dtype=np.float32)
            self.action_units = ['-'] * self.nu
        else:
            self.hover_thrust = 0.0
            self.action_space = Box(low=self.physical_action_bounds[0], high=self.physical_action_bounds[1], dtype=np.float32)
            self.action_units = ['N'] * self.nu
        # state / observation space (quadrotor.py:640-712)
        pos_t, vel_t = 2, 30
        ang, yaw, rate = 85 * math.pi / 180, 180 * math.pi / 180, 500 * math.pi / 180
        gz = self.GROUND_PLANE_Z
        if qt == 1:
            low, high = np.array([gz, -vel_t]), np.array([pos_t, vel_t])
            self.state_labels, self.state_units = ['z', 'z_dot'], ['m', 'm/s']
        elif qt == 2:
            low = np.array([-pos_t, -vel_t, gz, -vel_t, -ang, -rate])
            high = np.array([pos_t, vel_t, pos_t, vel_t, ang, rate])
            self.state_labels = ['x', 'x_dot', 'z', 'z_dot', 'theta', 'theta_dot']
            self.state_units = ['m', 'm/s', 'm', 'm/s', 'rad', 'rad/s']
        else:
            low = np.array([-pos_t, -vel_t, -pos_t, -vel_t, gz, -vel_t, -ang, -ang, -yaw, -rate, -rate, -rate])
            high = np.array([pos_t, vel_t, pos_t, vel_t, pos_t, vel_t, ang, ang, yaw, rate, rate, rate])
            self.state_labels = ['x', 'x_dot', 'y', 'y_dot', 'z', 'z_dot', 'phi', 'theta', 'psi', 'p', 'q', 'r']
            self.state_units = ['m', 'm/s'] * 3 + ['rad'] * 3 + ['rad/s'] * 3
        self.state_space = Box(low=low, high=high, dtype=np.float32)
        mul = self._obs_multiplier()
        self.observation_space = Box(low=np.concatenate([low] * mul), high=np.concatenate([high] * mul), dtype=np.float32)
        self.obs_dim = self.nx * mul
        # initial state (quadrotor.py:207-231).  NB: the reference re-installs its BASE randomisation
        # tables after BenchmarkEnv.__init__ stored the YAML ones (:208, :233), so the YAML keys
        # `init_state_randomization_info` / `inertial_prop_randomization_info` are ignored upstream.
        # respect_randomization_info=True (extension) honours them instead.
        labels = QUAD_INIT_LABELS[qt]
        init = kw['init_state']
        if init is None:
            vals = [0.] * len(labels)
        elif isinstance(init, np.ndarray):
            vals = [init[i] for i in range(len(labels))]
        elif isinstance(init, dict):
            vals = [init.get(k, 0.) for k in labels]
        else:
            raise ValueError('[ERROR] in Quadrotor.__init__(), init_state incorrect format.')
        self.init_labels = labels
        self.init_values = [float(v) for v in vals]
        respect = bool(kw['respect_randomization_info'])
        user_init = kw['init_state_randomization_info']
        self.init_rand_info = dict(user_init) if (respect and user_init is not None) else dict(QUAD_BASE_INIT_RAND)
        self.init_rand_info = {k: v for k, v in self.init_rand_info.items() if k in labels}
        user_par = kw['inertial_prop_randomization_info']
        pri = dict(user_par) if (respect and user_par is not None) else dict(QUAD_BASE_INERTIAL_RAND)
        if qt == 1:
            for k in ('Ixx', 'Iyy', 'Izz'):
                pri.pop(k, None)
        elif qt == 2:
            for k in ('Ixx', 'Izz'):
                pri.pop(k, None)
        self.param_rand_info = pri
        # inertial_prop override (quadrotor.py:244-259)
        ip = kw['inertial_prop']
        if ip is None:
            pass
        elif qt == 1 and np.array(ip).shape == (1,):
            self.MASS = ip[0]
        elif qt == 2 and np.array(ip).shape == (2,):
            self.MASS, self.J[1, 1] = ip
        elif qt == 3 and np.array(ip).shape == (4,):
            self.MASS, self.J[0, 0], self.J[1, 1], self.J[2, 2] = ip
        elif isinstance(ip, dict):
            self.MASS = ip.get('M', self.MASS)
            self.J[0, 0] = ip.get('Ixx', self.J[0, 0])
            self.J[1, 1] = ip.get('Iyy', self.J[1, 1])
            self.J[2, 2] = ip.get('Izz', self.J[2, 2])
        else:
            raise ValueError('[ERROR] in Quadrotor.__init__(), inertial_prop incorrect format.')
        self.param_labels = QUAD_PARAM_LABELS
        self.param_values = [float(self.MASS), float(self.J[0, 0]), float(self.J[1, 1]), float(self.J[2, 2])]
        # references (quadrotor.py:261-323)
        self.U_GOAL = np.ones(self.nu) * self.MASS * self.GRAVITY_ACC / self.nu
        ti = self.TASK_INFO
        if self.TASK == 'stabilization':
            g = ti['stabilization_goal']
            # (quadrotor.py:264-278 indexes the goal: [x, z] for 1-D — z is entry 1 — and 2-D, [x, y, z] for 3-D; a shorter list
            #  raises IndexError there, and here — not a NaN reference)
            if len(g) < (3 if qt == 3 else 2):
                raise IndexError('list index out of range')
            self.X_GOAL = np.asarray({1: [g[1], 0.0], 2: [g[0], 0.0, g[1], 0.0, 0.0, 0.0],
                                      3: [g[0], 0.0, g[1], 0.0] + [g[2] if qt == 3 else 0.0, 0.0] + [0.0] * 6}[qt], dtype=float)
            self.goal_tolerance = float(ti['stabilization_goal_tolerance'])
        else:
            pos, vel = self._trajectory(ti, ti['trajectory_position_offset'])
            z = np.zeros(pos.shape[0])
            if qt == 1:
                self.X_GOAL = np.vstack([pos[:, 2], vel[:, 2]]).T
            elif qt == 2:
                self.X_GOAL = np.vstack([pos[:, 0], vel[:, 0], pos[:, 2], vel[:, 2], z, z]).T
            else:
                pt, vt = project_on_plane(pos, vel, ti['proj_point'], ti['proj_normal'])
                self.X_GOAL = np.vstack([pt[:, 0], vt[:, 0], pt[:, 1], vt[:, 1], pt[:, 2], vt[:, 2], z, z, z, z, z, z]).T
            self.goal_tolerance = 0.0
        self.Q = np.diag(_diag_weights(self.rew_state_weight, self.nx, 'state'))
        self.R = np.diag(_diag_weights(self.rew_act_weight, self.nu, 'input'))
        self.dyn_dim = qt
        self.obs_wrap_angle = False
        self.x_threshold, self.theta_threshold_radians = 0.0, 0.0
        self.pole_box_width = 0.0
        if kw['engine_arm'] is None:     # the prior-model integrator uses the prior model's arm
            kw['engine_arm'] = 'symbolic' if kw['integrator'] == 'rk4' else 'pybullet'
        if kw['engine_arm'] == 'pybullet':
            self.engine_arm = c['prop_offset']            # cf2x.urdf:42-78 (what Bullet integrates)
        elif kw['engine_arm'] == 'symbolic':
            self.engine_arm = c['arm'] / math.sqrt(2.0)   # quadrotor.py:509,555-556 (CasADi prior)
        else:
            raise ValueError("engine_arm must be 'pybullet' or 'symbolic'")

    # ------------------------------------------------------------------ constraints -> rows
    def _compile_constraints(self):
        """Flatten `constraints:` (constraints.py:639-665) into scalar rows, in evaluation order."""
        self.con_rows, self.quad_P, self.con_meta = [], [], []
        specs = self.kw['constraints']
        self.CONSTRAINTS = specs
        if specs is None:
            return
        for spec in specs:
            assert isinstance(spec, dict), '[ERROR]: Each constraint must be specified as a dict.'
            assert 'constraint_form' in spec, "[ERROR]: Each constraint must have a key 'constraint_form'"
            form = spec['constraint_form']
            forms = ('linear_constraint', 'quadratic_constraint', 'bounded_constraint', 'default_constraint')
            if self.name == 'cartpole':
                forms += ('abs_bound',)
            assert form in forms, '[ERROR]. constraint not in list of available constraints'
            cfg = {k: v for k, v in spec.items() if k != 'constraint_form'}
            var = _enum_str(cfg.pop('constrained_variable'))
            if var not in ('state', 'input'):
                if form != 'default_constraint' and var != 'input_and_state':
                    raise ValueError(f"'{var}' is not a valid ConstrainedVariableType")               # constraints.py:55
                raise NotImplementedError('[ERROR] DefaultConstraint can only be of type STATE or INPUT' if form == 'default_constraint' else
                                          'only STATE and INPUT constraints are supported (INPUT_AND_STATE is not evaluable upstream either)')
            var_id = 0 if var == 'state' else 1
            full_dim = self.nx if var_id == 0 else self.nu
            strict = bool(cfg.pop('strict', False))
            decimals = cfg.pop('decimals', 8)
            tolerance = cfg.pop('tolerance', None)
            active = cfg.pop('active_dims', None)
            # the checks each class makes BEFORE Constraint.__init__ looks at active_dims (same exception types, same order)
            if form == 'default_constraint' and active is not None:
                raise TypeError("DefaultConstraint.__init__() got an unexpected keyword argument 'active_dims'")
            if form == 'bounded_constraint':
                missing = [k for k in ('lower_bounds', 'upper_bounds') if k not in cfg]
                if missing:
                    raise TypeError(f'BoundedConstraint.__init__() missing {len(missing)} required positional argument' + ('s' if len(missing) > 1 else '') +
                                    ': ' + ' and '.join(f"'{k}'" for k in missing))
                n_lb, n_ub = np.array(cfg['lower_bounds'], ndmin=1).shape[0], np.array(cfg['upper_bounds'], ndmin=1).shape[0]
                if isinstance(active, list):                                                          # constraints.py:314-319
                    assert n_lb == len(active), '[Error] active_dims and lower_bounds must have the same dimension.'
                if isinstance(active, int):
                    assert n_lb == 1, '[Error] active_dims and lower_bounds must have the same dimension.'
                    assert n_ub == 1, '[Error] active_dims and upper_bounds must have the same dimension.'
            if form == 'abs_bound':
                assert cfg.get('bound') is not None
                if isinstance(cfg['bound'], (list, tuple)):
                    raise TypeError("bad operand type for unary -: 'list'")   # same failure as upstream (:433)
            if isinstance(active, int):
                active = [active]
            if active is not None:                                                                    # constraints.py:68-78
                assert isinstance(active, (list, np.ndarray)), '[ERROR] active_dims is not a list/array.'
                assert len(active) <= full_dim, '[ERROR] more active_dim than constrainable self.dim'
                assert all(isinstance(n, int) for n in active), '[ERROR] non-integer active_dim.'
                assert all(n < full_dim for n in active), '[ERROR] active_dim not stricly smaller than self.dim.'
                assert len(active) == len(set(active)), '[ERROR] duplicates in active_dim'
            idx = list(range(full_dim)) if active is None else list(active)
            rs = 10.0 ** decimals
            first = len(self.con_rows)
            if form in ('default_constraint', 'bounded_constraint'):
                if form == 'default_constraint':
                    assert active is None
                    if var_id == 0:
                        lo_d, hi_d = self.state_space.low, self.state_space.high
                    else:
                        lo_d = np.asarray(self.physical_action_bounds[0], dtype=np.float32)
                        hi_d = np.asarray(self.physical_action_bounds[1], dtype=np.float32)
                    lb, ub = cfg.pop('lower_bounds', None), cfg.pop('upper_bounds', None)
                    lb = lo_d if lb is None else np.array(lb, ndmin=1)
                    ub = hi_d if ub is None else np.array(ub, ndmin=1)
                    assert len(ub) == full_dim, '[ERROR]: Upper bound must have length equal to space dimension.'      # constraints.py:380-387
                    assert len(lb) == full_dim, '[ERROR]: Lower bound must have length equal to space dimension.'
                    lb, ub = lb.astype(np.float64), ub.astype(np.float64)
                else:
                    lb = np.array(cfg.pop('lower_bounds'), ndmin=1, dtype=np.float64)
                    ub = np.array(cfg.pop('upper_bounds'), ndmin=1, dtype=np.float64)
                    assert lb.shape[0] == len(idx) and ub.shape[0] == len(idx), '[ERROR] Dimension 0 of b does not match A!'   # constraints.py:272
                # A = [-I; I], b = [-lb; ub], both stored as float32 (constraints.py:267-268,320-321)
                b32 = np.hstack((-lb, ub)).astype(np.float32)
                n = len(idx)
                sym = dict(A=np.vstack((-np.eye(n), np.eye(n))).astype(np.float32), b=b32, lower_bounds=lb, upper_bounds=ub)
                for j in range(n):
                    self.con_rows.append(dict(kind=L.ROW_SPARSE, var=var_id, index=idx[j], strict=strict, sign=-1.0,
                                              b=float(b32[j]), round_scale=rs))
                for j in range(n):
                    self.con_rows.append(dict(kind=L.ROW_SPARSE, var=var_id, index=idx[j], strict=strict, sign=1.0,
                                              b=float(b32[n + j]), round_scale=rs))
            elif form == 'linear_constraint':
                A = np.asarray(cfg.pop('A'), dtype=np.float32).reshape(-1, len(idx))
                b = np.asarray(cfg.pop('b'), dtype=np.float32).reshape(-1)
                assert b.shape[0] == A.shape[0], '[ERROR] Dimension 0 of b does not match A!'
                full = A @ np.eye(full_dim)[idx]               # A @ constraint_filter, float64
                sym = dict(A=A, b=b)
                for r in range(A.shape[0]):
                    self.con_rows.append(dict(kind=L.ROW_DENSE, var=var_id, index=0, strict=strict, sign=1.0,
                                              b=float(b[r]), round_scale=rs, coef=full[r].tolist()))
            elif form == 'quadratic_constraint':
                P = np.array(cfg.pop('P'), ndmin=1, dtype=float)
                assert P.shape == (len(idx), len(idx)), ('[ERROR] P has the wrong dimension! It should match the dimension of'
                                                         'the constrained_variable or the same length as active_dims.')
                b = cfg.pop('b')
                assert isinstance(b, float), '[ERROR] b is not a scalar!'
                if len(self.quad_P) >= L.MAX_QUAD_CON:
                    raise ValueError(f'at most {L.MAX_QUAD_CON} quadratic constraints are supported')
                F = np.eye(full_dim)[idx]
                self.quad_P.append(F.T @ P @ F)
                sym = dict(P=P, b=b)
                self.con_rows.append(dict(kind=L.ROW_QUADRATIC, var=var_id, index=len(self.quad_P) - 1, strict=strict,
                                          sign=1.0, b=float(b), round_scale=rs))
            else:   # abs_bound (SymmetricStateConstraint, cartpole only)
                bound = cfg.pop('bound')
                bound = np.array(bound, ndmin=1, dtype=float)
                assert bound.shape[0] == len(idx)
                n = len(idx)
                sym = dict(A=np.vstack((-np.eye(n), np.eye(n))).astype(np.float32), b=np.hstack((bound, bound)).astype(np.float32), bound=bound)
                if tolerance is not None and len(np.array(tolerance, ndmin=1)) != n:                  # constraints.py:449-455
                    raise ValueError('[ERROR] the tolerance dimension does not match the number of constraints.')
                assert self.COST == 'rl_reward', '[ERROR] SymmetricStateConstraint is meant for RL environments'   # after super().__init__ (:446-447)
                for j in range(len(idx)):
                    self.con_rows.append(dict(kind=L.ROW_ABS, var=var_id, index=idx[j], strict=strict, sign=1.0,
                                              b=float(bound[j]), round_scale=rs))
            if cfg:
                raise TypeError(f'unexpected constraint argument(s) {sorted(cfg)} for {form}')
            if tolerance is not None and len(np.array(tolerance, ndmin=1)) != len(self.con_rows) - first:      # check_tolerance_shape, constraints.py:175-178
                raise ValueError('[ERROR] the tolerance dimension does not match the number of constraints.')
            self.con_meta.append(ConstraintInfo(form=form, var=var, first_row=first, n_rows=len(self.con_rows) - first, strict=strict,
                                                constrained_variable=var, dim=len(idx), num_constraints=len(self.con_rows) - first,
                                                decimals=decimals, constraint_filter=np.eye(full_dim)[idx],
                                                tolerance=None if tolerance is None else np.array(tolerance, ndmin=1), **sym))
        if len(self.con_rows) > L.MAX_CON_ROWS:
            raise ValueError(f'more than {L.MAX_CON_ROWS} scalar constraint rows')
        self.num_constraints = len(self.con_rows)

    def state_constraint_values(self, state):
        """Values of the state-constraint rows for a batch of env.state vectors (torch tensor [n, nx], any device): what
        `info['constraint_values']` holds after a reset (benchmark_env.py:356-357, constraints.py:97-109).  The step kernel
        returns the pre-reset values of a finished episode; collectors that feed the constraint values of the NEW episode
        to a policy (Safe-Explorer) evaluate them here from the returned state."""
        import torch
        x = state.to(torch.float64)
        cols = []
        for r in self.con_rows:
            if r['var'] != 0:
                continue
            if r['kind'] == L.ROW_SPARSE:
                c = r['sign'] * x[:, r['index']] - r['b']
            elif r['kind'] == L.ROW_ABS:
                c = x[:, r['index']].abs() - r['b']
            elif r['kind'] == L.ROW_DENSE:
                c = x @ torch.as_tensor(r['coef'][:x.shape[1]], dtype=torch.float64, device=x.device) - r['b']
            else:
                P = torch.as_tensor(self.quad_P[r['index']], dtype=torch.float64, device=x.device)
                c = ((x @ P) * x).sum(-1) - r['b']
            if r['round_scale'] > 0:
                c = torch.round(c * r['round_scale']) / r['round_scale']
            cols.append(c)
        return torch.stack(cols, dim=1) if cols else x.new_zeros((x.shape[0], 0))

    @property
    def n_state_con_rows(self):
        return sum(1 for r in self.con_rows if r['var'] == 0)

    # ------------------------------------------------------------------ disturbances -> tables
    def _compile_disturbances(self):
        """`disturbances:` (benchmark_env.py:279-295, disturbances.py:285-303) -> per-channel tables."""
        self.dist = {L.CH_ACTION: [], L.CH_DYNAMICS: [], L.CH_OBSERVATION: []}
        self.DISTURBANCES = self.kw['disturbances']
        dims = {'observation': self.obs_dim, 'action': self.nu, 'dynamics': self.dyn_dim}
        chan = {'observation': L.CH_OBSERVATION, 'action': L.CH_ACTION, 'dynamics': L.CH_DYNAMICS}
        self.DISTURBANCE_MODES = {k: {'dim': v} for k, v in dims.items()}
        max_step = int(self.EPISODE_LEN_SEC / self.CTRL_TIMESTEP)
        if self.DISTURBANCES is not None:
            for mode, specs in self.DISTURBANCES.items():
                assert mode in dims, '[ERROR] in BenchmarkEnv._setup_disturbances(), disturbance mode not available.'
                dim = dims[mode]
                if mode == 'observation':
                    if dim != self.nx and any(s.get('disturbance_func') in ('uniform', 'white_noise', 'periodic') or
                                              s.get('mask') is not None for s in specs):
                        # upstream adds an obs_dim-sized noise vector to the state-sized observation
                        # (quadrotor.py:717,805-807) which raises when a goal horizon is configured — after its constructors have
                        # checked the spec against dim = obs_dim (their AssertionErrors come first)
                        for s in specs:
                            assert 'disturbance_func' in s.keys(), '[ERROR]: Every distrubance must specify a disturbance_func.'
                            self._compile_disturbance(s, self.obs_dim, max_step)
                        raise ValueError('observation disturbances with per-dimension noise need obs_dim == state_dim')
                    dim = self.nx
                if len(specs) > L.MAX_DISTURB:
                    raise ValueError(f'at most {L.MAX_DISTURB} disturbances per channel')
                for s in specs:
                    assert 'disturbance_func' in s.keys(), '[ERROR]: Every distrubance must specify a disturbance_func.'
                    self.dist[chan[mode]].append(self._compile_disturbance(s, dim, max_step))
        adv = self.kw['adversary_disturbance']
        self.adversary_disturbance = adv
        self.adversary_channel = -1
        if adv is not None:
            assert adv in dims, '[ERROR] in Cartpole._setup_disturbances()'      # (benchmark_env.py:290 — the base class says Cartpole for both robots)
            if adv == 'observation':
                raise NotImplementedError('adversary on the observation channel is never applied upstream either')
            self.adversary_channel = chan[adv]
            self.adversary_dim = dims[adv]
            self.adversary_action_space = Box(low=-1, high=1, shape=(self.adversary_dim,))
        self._check_quad1d_lateral_drift()

    def _check_quad1d_lateral_drift(self):
        """Upstream's 1-D quadrotor is a free 3-D body that REPORTS (z, z_dot): its `init_x` / `init_x_dot` entries (and their
        default randomisation, U(-0.5, 0.5) / U(-0.01, 0.01)) set the drone's X position and velocity (quadrotor.py:209-211,
        :361-384), which the observation never shows.  Alone that drift is invisible.  With a force on the `dynamics` channel it
        is not: base_aviary.py:271-279 applies the disturbance at the position cached at the start of the control step, so a
        drone that has moved dx since then gets the torque -dx * F_z about y, pitches, and its thrust leaves the z axis (the
        oracle restates this; found by tests/test_gpu_config_fuzz.py).  The 1-D kernel integrates (z, z_dot) only and would
        silently return different trajectories — refuse the combination instead."""
        if self.name != 'quadrotor' or self.system != L.QUAD_1D:
            return
        if not self.dist[L.CH_DYNAMICS] and self.adversary_channel != L.CH_DYNAMICS:
            return
        x_dot0 = self.init_values[self.init_labels.index('init_x_dot')]
        drawn = bool(self.kw['randomized_init']) and 'init_x_dot' in self.init_rand_info
        if x_dot0 != 0.0 or drawn:
            raise NotImplementedError(
                'quad_type 1 with a dynamics disturbance / adversary and a non-zero (or randomised) init_x_dot: upstream lets the '
                'unobserved X drift turn the disturbance into a pitch torque (base_aviary.py:272); the 1-D kernel does not model '
                "that.  Use init_x_dot = 0 with randomized_init off (or respect_randomization_info with a table that leaves "
                "init_x_dot out), or quad_type 2.")

    @property
    def adversary_observation_space(self):
        """benchmark_env.py:206-208: the adversary observes what the protagonist observes (read through
        `env.get_attr('adversary_observation_space')` by rarl.py:70 / rap.py:70)."""
        if self.adversary_disturbance is None:
            raise AttributeError('adversary_observation_space (no adversary_disturbance configured)')
        return self.observation_space

    @staticmethod
    def _compile_disturbance(spec, dim, max_step):
        kind = spec['disturbance_func']
        cfg = {k: v for k, v in spec.items() if k != 'disturbance_func'}
        mask = cfg.pop('mask', None)
        m32 = None
        if mask is not None:
            m32 = np.asarray(mask, dtype=np.float32)
            assert dim == len(m32)
        d = dict(kind=L.DIST_NONE, dim=dim, step_offset=-1, max_step=max_step, duration=1.0, decay_rate=1.0,
                 frequency=1.0, a=[0.0] * dim, b=[0.0] * dim, mask=[1.0] * dim)
        if kind in ('impulse', 'step'):
            mag = cfg.pop('magnitude', 1)
            off = cfg.pop('step_offset', None)
            d['step_offset'] = -1 if off is None else int(off)
            if kind == 'impulse':
                d['kind'] = L.DIST_IMPULSE
                d['duration'] = float(cfg.pop('duration', 1))
                d['decay_rate'] = float(cfg.pop('decay_rate', 1))
                assert d['duration'] >= 1 and 0 < d['decay_rate'] <= 1
                # magnitude * decay is a NumPy float64 scalar upstream -> `* mask` stays float64
                d['a'] = [float(mag) * (float(m32[j]) if m32 is not None else 1.0) for j in range(dim)]
            else:
                d['kind'] = L.DIST_STEP
                # `noise` is a Python scalar upstream -> `noise *= mask` is evaluated in float32 (NumPy 2)
                if m32 is not None:
                    d['a'] = [float(v) for v in (np.float32(mag) * m32)]
                else:
                    d['a'] = [float(mag)] * dim
        elif kind == 'uniform':
            low, high = cfg.pop('low', 0.0), cfg.pop('high', 1.0)
            if isinstance(low, float):
                lo = [low] * dim
            elif isinstance(low, list):
                lo = list(low)
            else:
                raise ValueError('[ERROR] UniformNoise.__init__(): low must be specified as a float or list.')
            if isinstance(high, float):
                hi = [high] * dim
            elif isinstance(low, list):
                hi = list(high)
            else:
                raise ValueError('[ERROR] UniformNoise.__init__(): high must be specified as a float or list.')
            # np_random.uniform(low, high, size=dim) (disturbances.py:188): the bounds broadcast against (dim,) — a one-element list is
            # fine, any other length mismatch is numpy's ValueError (upstream: at the first step; here: at construction)
            lo, hi = np.broadcast_to(np.asarray(lo, dtype=float), (dim,)), np.broadcast_to(np.asarray(hi, dtype=float), (dim,))
            d.update(kind=L.DIST_UNIFORM, a=[float(v) for v in lo], b=[float(v) for v in hi])
        elif kind == 'white_noise':
            std = cfg.pop('std', 1.0)
            if isinstance(std, float):
                sd = [std] * dim
            elif isinstance(std, list):
                sd = list(std)
            else:
                raise ValueError('[ERROR] WhiteNoise.__init__(): std must be specified as a float or list.')
            assert dim == len(sd), 'std shape should be the same as dim.'
            d.update(kind=L.DIST_WHITE, a=[float(v) for v in sd])
        elif kind == 'periodic':
            d.update(kind=L.DIST_PERIODIC, a=[float(cfg.pop('scale', 1.0))] * dim, frequency=float(cfg.pop('frequency', 1.0)))
            m32 = None        # PeriodicNoise drops the mask (disturbances.py:244)
        else:
            raise AssertionError('[ERROR] in BenchmarkEnv._setup_disturbances(), disturbance type not available.')
        if m32 is not None and kind in ('uniform', 'white_noise'):
            d['mask'] = [float(v) for v in m32]
        return d

    # ------------------------------------------------------------------ randomisation specs
    @staticmethod
    def _rand_spec(info):
        r = L.Rand()
        if info is None:
            r.kind = L.RAND_NONE
            return r
        info = dict(info)
        distrib = info.pop('distrib')
        args = list(info.pop('args', []))
        # upstream: getattr(np_random, distrib)(*args, **kwargs) at the first reset (benchmark_env.py:232-262) — same exception types, at construction
        if not hasattr(np.random.Generator, distrib):
            raise AttributeError(f"'numpy.random._generator.Generator' object has no attribute '{distrib}'")
        allowed = {'uniform': ('low', 'high'), 'normal': ('loc', 'scale'), 'choice': ('a',)}.get(distrib, ())
        numpy_only = {'uniform': ('size',), 'normal': ('size',), 'choice': ('size', 'replace', 'p', 'axis', 'shuffle')}.get(distrib, ())
        for k in info:
            if k in numpy_only:
                raise NotImplementedError(f"{distrib}(..., {k}=) is not available in the HIP kernels")
            if allowed and k not in allowed:
                raise TypeError(f"{distrib}() got an unexpected keyword argument '{k}'")
        if distrib == 'uniform':
            r.kind = L.RAND_UNIFORM
            r.p0 = float(args[0] if len(args) > 0 else info.get('low', 0.0))
            r.p1 = float(args[1] if len(args) > 1 else info.get('high', 1.0))
        elif distrib == 'normal':
            r.kind = L.RAND_NORMAL
            r.p0 = float(args[0] if len(args) > 0 else info.get('loc', 0.0))
            r.p1 = float(args[1] if len(args) > 1 else info.get('scale', 1.0))
        elif distrib == 'choice':
            opts = list(args[0] if args else info['a'])
            if len(opts) > L.MAX_CHOICE:
                raise ValueError(f'choice() with more than {L.MAX_CHOICE} options is not supported')
            r.kind, r.n_choice = L.RAND_CHOICE, len(opts)
            for k, v in enumerate(opts):
                r.choices[k] = float(v)
        else:
            raise NotImplementedError(f"randomisation distribution '{distrib}' is not available in the HIP kernels "
                                      "(supported: uniform, normal, choice)")
        return r

    # ------------------------------------------------------------------ -> scg_config
    def to_c_config(self, num_envs, dtype, seed, env_id_offset=0, auto_reset=True):
        kw = self.kw
        c = L.Config()
        c.auto_reset = int(bool(auto_reset))
        if self.kw['integrator'] not in ('pyb_euler', 'rk4'):
            raise ValueError("integrator must be 'pyb_euler' or 'rk4'")
        rk4 = self.kw['integrator'] == 'rk4'
        c.abi_version, c.system, c.dtype = L.SCG_ABI_VERSION, self.system, dtype
        c.integrator = L.INT_RK4 if rk4 else L.INT_PYB_EULER
        c.num_envs, c.env_id_offset, c.seed = int(num_envs), int(env_id_offset), int(seed) & 0xFFFFFFFFFFFFFFFF
        # benchmark_env.py:148,499: CTRL_STEPS = EPISODE_LEN_SEC * CTRL_FREQ stays a FLOAT upstream and the time limit is
        # `ctrl_step_counter >= CTRL_STEPS` — for 3.5 s at 15 Hz (52.5) the episode ends at step 53: the integer the kernels
        # compare with is the ceiling (truncating gave 52: one step early; found by tests/test_gpu_config_fuzz.py)
        c.substeps, c.ctrl_steps = self.PYB_STEPS_PER_CTRL, self.max_episode_steps
        c.pyb_dt, c.ctrl_dt = self.PYB_TIMESTEP, self.CTRL_TIMESTEP
        if rk4:     # `rk4_substeps` RK4 steps of the prior model per control period (mpc_utils.py:42-64 uses one)
            c.substeps = int(self.kw['rk4_substeps'])
            c.pyb_dt = self.CTRL_TIMESTEP / c.substeps
        c.task = L.TASK_TRAJ_TRACKING if self.TASK == 'traj_tracking' else L.TASK_STABILIZATION
        c.cost = L.COST_QUADRATIC if self.COST == 'quadratic' else L.COST_RL_REWARD
        c.obs_goal_horizon = self.obs_goal_horizon
        xg = np.atleast_2d(self.X_GOAL)
        c.goal_rows = xg.shape[0]
        c.rew_exponential = int(self.rew_exponential)
        c.done_on_out_of_bound = int(self.done_on_out_of_bound)
        c.done_on_violation = int(bool(kw['done_on_violation']))
        c.use_constraint_penalty = int(bool(kw['use_constraint_penalty']))
        c.obs_wrap_angle = int(self.obs_wrap_angle)
        c.normalized_action = int(self.NORMALIZED_RL_ACTION_SPACE)
        c.info_goal_reached = int(self.TASK == 'stabilization' and self.COST == 'quadratic')
        c.goal_tolerance = self.goal_tolerance
        c.constraint_penalty = float(kw['constraint_penalty'])
        rsw = _diag_weights(self.rew_state_weight, self.nx, 'state') if self.rew_state_weight.size in (1, self.nx) else None
        raw = _diag_weights(self.rew_act_weight, self.nu, 'input') if self.rew_act_weight.size in (1, self.nu) else None
        if rsw is None or raw is None:
            raise Exception('Wrong dimension for cost weights.')
        for k in range(self.nx):
            c.rew_state_weight[k] = rsw[k]
            c.q_diag[k] = self.Q[k, k]
            c.mse_weight[k] = self.info_mse_metric_state_weight[k]
            c.state_low[k] = float(self.state_space.low[k])
            c.state_high[k] = float(self.state_space.high[k])
            c.init_state[k] = self.init_values[k]
            c.init_rand[k] = self._rand_spec(self.init_rand_info.get(self.init_labels[k]))
        for j in range(self.nu):
            c.rew_act_weight[j] = raw[j]
            c.r_diag[j] = self.R[j, j]
            c.u_goal[j] = self.U_GOAL[j]
            c.act_low[j] = float(self.physical_action_bounds[0][j])
            c.act_high[j] = float(self.physical_action_bounds[1][j])
        c.x_threshold, c.theta_threshold = self.x_threshold, self.theta_threshold_radians
        if self.name == 'cartpole':
            c.act_scale = float(self.action_scale)
        else:
            c.act_scale = float(self.norm_act_scale)
            c.kf, c.km = self.KF, self.KM
            c.pwm2rpm_scale, c.pwm2rpm_const = self.PWM2RPM_SCALE, self.PWM2RPM_CONST
            c.pwm_min, c.pwm_max = self.MIN_PWM, self.MAX_PWM
        c.hover_thrust = self.hover_thrust
        c.gravity, c.arm = self.GRAVITY_ACC, self.engine_arm
        c.max_coordinate_velocity = BULLET_MAX_COORDINATE_VELOCITY
        c.pole_box_width = self.pole_box_width
        for k, v in enumerate(self.param_values):
            c.base_param[k] = v
            c.param_rand[k] = self._rand_spec(self.param_rand_info.get(self.param_labels[k]))
        c.randomized_inertial_prop = int(bool(kw['randomized_inertial_prop']))
        c.randomized_init = int(bool(kw['randomized_init']))
        for ch, lst in self.dist.items():
            c.n_dist[ch] = len(lst)
            for k, d in enumerate(lst):
                t = c.dist[ch][k]
                t.kind, t.dim, t.step_offset, t.max_step = d['kind'], d['dim'], d['step_offset'], d['max_step']
                t.duration, t.decay_rate, t.frequency = d['duration'], d['decay_rate'], d['frequency']
                for j in range(d['dim']):
                    t.a[j], t.b[j], t.mask[j] = d['a'][j], d['b'][j], d['mask'][j]
        c.adversary_channel = self.adversary_channel
        c.adversary_scale = float(kw['adversary_disturbance_scale'])
        c.adversary_offset = float(kw['adversary_disturbance_offset'])
        c.n_con_rows, c.n_state_con_rows = len(self.con_rows), self.n_state_con_rows
        for r, row in enumerate(self.con_rows):
            t = c.con[r]
            t.kind, t.var, t.index, t.strict = row['kind'], row['var'], row['index'], int(row['strict'])
            t.sign, t.b, t.round_scale = row['sign'], row['b'], row['round_scale']
            for j, v in enumerate(row.get('coef', [])):
                t.coef[j] = v
        for q, P in enumerate(self.quad_P):
            n = P.shape[0]
            for a in range(n):
                for b in range(n):
                    c.quad_P[q][a * n + b] = P[a, b]
        return c, np.ascontiguousarray(xg, dtype=np.float64)

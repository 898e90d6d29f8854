"""ctypes binding of libscg_hip.so (include/scg_hip.h).

The shared library is the product: if it cannot be loaded this module raises — there is no
CPU fallback of any kind (the float64 NumPy oracle under /oracle is test infrastructure and
is never imported from here).
"""
import ctypes as C
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(PKG_DIR, 'csrc')
LIB_PATH = os.path.join(PKG_DIR, 'libscg_hip.so')
SOURCES = ['scg_kernels.hip']
HEADERS = ['scg_env_core.h', 'scg_params.h', 'scg_rng.h', os.path.join('..', '..', 'include', 'scg_hip.h')]

SCG_ABI_VERSION = 1
MAX_STATE, MAX_ACTION, MAX_GOAL_HORIZON = 12, 4, 4
MAX_CON_ROWS, MAX_QUAD_CON, MAX_DISTURB, MAX_PARAM, MAX_CHOICE = 64, 2, 4, 4, 8

F32, F64 = 0, 1
CARTPOLE, QUAD_1D, QUAD_2D, QUAD_3D = 0, 1, 2, 3
TASK_STABILIZATION, TASK_TRAJ_TRACKING = 0, 1
COST_RL_REWARD, COST_QUADRATIC = 0, 1
INT_PYB_EULER, INT_RK4 = 0, 1
DIST_NONE, DIST_IMPULSE, DIST_STEP, DIST_UNIFORM, DIST_WHITE, DIST_PERIODIC = range(6)
CH_ACTION, CH_DYNAMICS, CH_OBSERVATION = 0, 1, 2
RAND_NONE, RAND_UNIFORM, RAND_NORMAL, RAND_CHOICE = range(4)
ROW_SPARSE, ROW_DENSE, ROW_ABS, ROW_QUADRATIC = range(4)
FLAG_TRUNCATED, FLAG_VIOLATION, FLAG_OOB, FLAG_GOAL = 1, 2, 4, 8

c_i32, c_u64, c_f64, c_vp = C.c_int32, C.c_uint64, C.c_double, C.c_void_p


class Disturbance(C.Structure):
    _fields_ = [('kind', c_i32), ('dim', c_i32), ('step_offset', c_i32), ('max_step', c_i32),
                ('duration', c_f64), ('decay_rate', c_f64), ('frequency', c_f64),
                ('a', c_f64 * MAX_STATE), ('b', c_f64 * MAX_STATE), ('mask', c_f64 * MAX_STATE)]


class Rand(C.Structure):
    _fields_ = [('kind', c_i32), ('n_choice', c_i32), ('p0', c_f64), ('p1', c_f64),
                ('choices', c_f64 * MAX_CHOICE)]


class ConRow(C.Structure):
    _fields_ = [('kind', c_i32), ('var', c_i32), ('index', c_i32), ('strict', c_i32),
                ('sign', c_f64), ('b', c_f64), ('round_scale', c_f64), ('coef', c_f64 * MAX_STATE)]


class Config(C.Structure):
    _fields_ = [
        ('abi_version', c_i32), ('system', c_i32), ('dtype', c_i32), ('integrator', c_i32),
        ('num_envs', c_i32), ('env_id_offset', c_i32), ('seed', c_u64),
        ('substeps', c_i32), ('ctrl_steps', c_i32), ('pyb_dt', c_f64), ('ctrl_dt', c_f64),
        ('task', c_i32), ('cost', c_i32), ('obs_goal_horizon', c_i32), ('goal_rows', c_i32),
        ('rew_exponential', c_i32), ('done_on_out_of_bound', c_i32), ('done_on_violation', c_i32),
        ('use_constraint_penalty', c_i32), ('obs_wrap_angle', c_i32), ('normalized_action', c_i32),
        ('info_goal_reached', c_i32), ('auto_reset', c_i32),
        ('goal_tolerance', c_f64), ('constraint_penalty', c_f64),
        ('rew_state_weight', c_f64 * MAX_STATE), ('rew_act_weight', c_f64 * MAX_ACTION),
        ('q_diag', c_f64 * MAX_STATE), ('r_diag', c_f64 * MAX_ACTION),
        ('mse_weight', c_f64 * MAX_STATE), ('u_goal', c_f64 * MAX_ACTION),
        ('state_low', c_f64 * MAX_STATE), ('state_high', c_f64 * MAX_STATE),
        ('x_threshold', c_f64), ('theta_threshold', c_f64),
        ('act_scale', c_f64), ('hover_thrust', c_f64),
        ('act_low', c_f64 * MAX_ACTION), ('act_high', c_f64 * MAX_ACTION),
        ('kf', c_f64), ('km', c_f64), ('pwm2rpm_scale', c_f64), ('pwm2rpm_const', c_f64),
        ('pwm_min', c_f64), ('pwm_max', c_f64),
        ('gravity', c_f64), ('arm', c_f64), ('max_coordinate_velocity', c_f64), ('pole_box_width', c_f64),
        ('base_param', c_f64 * MAX_PARAM),
        ('randomized_inertial_prop', c_i32), ('randomized_init', c_i32),
        ('param_rand', Rand * MAX_PARAM),
        ('init_state', c_f64 * MAX_STATE), ('init_rand', Rand * MAX_STATE),
        ('n_dist', c_i32 * 3), ('adversary_channel', c_i32),
        ('dist', (Disturbance * MAX_DISTURB) * 3),
        ('adversary_scale', c_f64), ('adversary_offset', c_f64),
        ('n_con_rows', c_i32), ('n_state_con_rows', c_i32),
        ('con', ConRow * MAX_CON_ROWS),
        ('quad_P', (c_f64 * (MAX_STATE * MAX_STATE)) * MAX_QUAD_CON),
    ]


class StepOut(C.Structure):
    _fields_ = [(n, c_vp) for n in (
        'd_obs', 'd_reward', 'd_done', 'd_flags', 'd_c_values', 'd_mse', 'd_terminal_obs', 'd_state',
        'd_noisy_action', 'd_ep_return', 'd_ep_length', 'd_ep_violation', 'd_ep_mse',
        'd_fin_return', 'd_fin_length', 'd_fin_violation', 'd_fin_mse')]


class RolloutOut(C.Structure):
    _fields_ = [(n, c_vp) for n in ('d_reward_sum', 'd_done_count', 'd_violation_count', 'd_last_obs')]


EXPORTS = ['scg_dims', 'scg_workspace_bytes', 'scg_create', 'scg_destroy', 'scg_reset', 'scg_step',
           'scg_rollout_random', 'scg_set_state', 'scg_get_state', 'scg_set_params', 'scg_get_params',
           'scg_set_counters', 'scg_get_counters', 'scg_gae', 'scg_last_error', 'scg_abi_version',
           'scg_sizeof_config', 'scg_sizeof_step_out']


class ScgError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile libscg_hip.so for gfx950 in-tree with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC_DIR, s) for s in SOURCES]
    deps = srcs + [os.path.normpath(os.path.join(CSRC_DIR, h)) for h in HEADERS]
    if not force and os.path.exists(LIB_PATH):
        if os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(d) for d in deps):
            return LIB_PATH
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-o', LIB_PATH] + srcs
    if verbose:
        print(' '.join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise ScgError('hipcc failed:\n' + res.stdout + res.stderr)
    return LIB_PATH


_lib = None


def lib():
    """The loaded library.  Raises ScgError (never falls back) if it is missing or inconsistent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ScgError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                       '(hipcc --offload-arch=gfx950).  There is no CPU fallback.')
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as exc:
        raise ScgError(f'cannot load {LIB_PATH}: {exc}') from exc
    for name in EXPORTS:
        if not hasattr(L, name):
            raise ScgError(f'{LIB_PATH} does not export {name}')
    L.scg_last_error.restype = C.c_char_p
    L.scg_sizeof_config.restype = C.c_size_t
    L.scg_sizeof_step_out.restype = C.c_size_t
    if L.scg_abi_version() != SCG_ABI_VERSION:
        raise ScgError('libscg_hip.so ABI version mismatch')
    if L.scg_sizeof_config() != C.sizeof(Config) or L.scg_sizeof_step_out() != C.sizeof(StepOut):
        raise ScgError(f'struct layout mismatch: scg_config {L.scg_sizeof_config()} vs ctypes {C.sizeof(Config)}, '
                       f'scg_step_out {L.scg_sizeof_step_out()} vs {C.sizeof(StepOut)}')
    L.scg_dims.argtypes = [C.POINTER(Config)] + [C.POINTER(c_i32)] * 5
    L.scg_workspace_bytes.argtypes = [C.POINTER(Config), C.POINTER(C.c_size_t)]
    L.scg_create.argtypes = [C.POINTER(Config), C.POINTER(c_f64), C.c_int, c_vp, C.c_size_t, C.POINTER(c_vp)]
    L.scg_destroy.argtypes = [c_vp]
    L.scg_reset.argtypes = [c_vp, c_vp, C.POINTER(StepOut), c_vp]
    L.scg_step.argtypes = [c_vp, c_vp, c_vp, C.POINTER(StepOut), c_vp]
    L.scg_rollout_random.argtypes = [c_vp, C.c_int, C.POINTER(RolloutOut), c_vp]
    for fn in (L.scg_set_state, L.scg_get_state, L.scg_set_params, L.scg_get_params):
        fn.argtypes = [c_vp, C.POINTER(c_f64), C.c_int, C.c_int, c_vp]
    L.scg_set_counters.argtypes = [c_vp, C.POINTER(c_i32), C.POINTER(C.c_uint32), C.c_int, C.c_int, c_vp]
    L.scg_get_counters.argtypes = [c_vp, C.POINTER(c_i32), C.POINTER(C.c_uint32), C.c_int, C.c_int, c_vp]
    L.scg_gae.argtypes = [C.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int,
                          c_f64, c_f64, C.c_int, c_vp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise ScgError(f'libscg_hip error {rc}: {lib().scg_last_error().decode()}')

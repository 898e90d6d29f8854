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
HEADERS = ['scg_env_core.h', 'scg_env_kernels.h', 'scg_gae_kernels.h', 'scg_mlp.h', 'scg_once.h', 'scg_params.h', 'scg_rng.h', 'scg_spec.h',
           os.path.join('..', '..', 'include', 'scg_hip.h')]         # (scg_mlp.h: the policy-in-the-loop variants include it)
SPEC_DIR = os.path.join(PKG_DIR, 'spec')

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
FLAG_TRUNCATED, FLAG_VIOLATION, FLAG_OOB, FLAG_GOAL, FLAG_GROUND = 1, 2, 4, 8, 16

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
        'd_noisy_action', 'd_ep_stats', 'd_fin_stats')]


class Policy(C.Structure):
    _fields_ = [('d_params', c_vp)] + [(n, c_i32) for n in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3', 'logstd_off', 'hidden',
                                                            'activation', 'deterministic')]


class PolicyRollout(C.Structure):
    _fields_ = [(n, c_vp) for n in ('d_obs', 'd_act', 'd_logp', 'd_reward', 'd_done', 'd_flags', 'd_terminal_obs',
                                    'd_ep_stats', 'd_episode_acc')] + [('max_episodes', c_i32)]


class Sequence(C.Structure):
    _fields_ = [(n, c_vp) for n in ('d_actions', 'd_adv_actions', 'd_obs', 'd_reward', 'd_done', 'd_flags', 'd_terminal_obs',
                                    'd_mse', 'd_c_values', 'd_ep_stats', 'd_fin_stats', 'd_state', 'd_noisy_action')]


class RolloutOut(C.Structure):
    _fields_ = [(n, c_vp) for n in ('d_reward_sum', 'd_done_count', 'd_violation_count', 'd_last_obs')]


EXPORTS = ['scg_dims', 'scg_workspace_bytes', 'scg_create', 'scg_destroy', 'scg_reset', 'scg_step', 'scg_step_range', 'scg_step_sequence', 'scg_rollout_policy',
           'scg_rollout_random', 'scg_set_state', 'scg_get_state', 'scg_set_params', 'scg_get_params',
           'scg_set_counters', 'scg_get_counters', 'scg_set_seed', 'scg_set_step_launch', 'scg_set_step_wsback', 'scg_rng_layout_version', 'scg_gae', 'scg_prior_model', 'scg_last_error', 'scg_abi_version',
           'scg_sizeof_config', 'scg_sizeof_step_out', 'scg_spec_source', 'scg_spec_hash', 'scg_source_hash']


class ScgError(RuntimeError):
    pass


# Code-generation flags that change results; part of the staleness hash.  -ffp-contract=on: a*b+c is fused only where the
# SOURCE writes it in one expression, so every kernel that inlines EnvOps::step (scg_step, scg_step_sequence,
# scg_rollout_policy, scg_rollout_random) rounds identically — with the default (fast) the backend fused differently per
# inlining context and K x scg_step differed from scg_step_sequence by 1 ulp per step.  Instruction counts are unchanged.
CODEGEN_FLAGS = '-ffp-contract=on'


def source_hash():
    """64-bit digest of the kernel sources; compiled into every library (-DSCG_SRC_HASH) and compared at load time, so a
    library built from older sources is never used silently (file times do not survive a copy of the tree)."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS):
        with open(os.path.normpath(os.path.join(CSRC_DIR, name)), 'rb') as f:
            h.update(name.encode() + b'\0' + f.read())
    h.update(CODEGEN_FLAGS.encode())
    h.update(b'sched:q2=max-ilp,q3=max-ilp,q3dist=iterative-ilp')              # (sched_flags below: part of what a library was built with)
    return int.from_bytes(h.digest()[:8], 'little')


def _hipcc():
    return os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def _lib_source_hash(path):
    """SCG_SRC_HASH baked into a built library, read from the file's bytes (never dlopen here: a library loaded once
    stays mapped under its path, a rebuilt file would not be seen).  0 if absent."""
    import re
    try:
        with open(path, 'rb') as f:
            m = re.search(rb'SCG_SRC_HASH:0x([0-9a-f]{16})ULL', f.read())
    except OSError:
        return 0
    return int(m.group(1), 16) if m else 0


def build(force=False, verbose=False):
    """Compile libscg_hip.so for gfx950 in-tree with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC_DIR, s) for s in SOURCES]
    if not force and os.path.exists(LIB_PATH) and _lib_source_hash(LIB_PATH) == source_hash():
        return LIB_PATH
    hipcc = _hipcc()
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-ffp-contract=on', '-std=c++17', '-fPIC', '-shared', f'-DSCG_SRC_HASH=0x{source_hash():016x}ULL',
           '-o', LIB_PATH] + srcs
    if verbose:
        print(' '.join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise ScgError('hipcc failed:\n' + res.stdout + res.stderr)
    return LIB_PATH


_lib = None
_spec_libs = {}


def _bind(path):
    if not os.path.exists(path):
        raise ScgError(f'{path} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                       '(hipcc --offload-arch=gfx950).  There is no CPU fallback.')
    try:
        L = C.CDLL(path)
    except OSError as exc:
        raise ScgError(f'cannot load {path}: {exc}') from exc
    for name in EXPORTS:
        if not hasattr(L, name):
            raise ScgError(f'{path} does not export {name}')
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
    L.scg_step_range.argtypes = [c_vp, C.c_int, C.c_int, c_vp, c_vp, C.POINTER(StepOut), c_vp]
    L.scg_rollout_random.argtypes = [c_vp, C.c_int, C.POINTER(RolloutOut), c_vp]
    L.scg_rollout_policy.argtypes = [c_vp, C.POINTER(Policy), C.c_int, C.POINTER(PolicyRollout), c_vp]
    L.scg_step_sequence.argtypes = [c_vp, C.c_int, C.POINTER(Sequence), c_vp]
    for fn in (L.scg_set_state, L.scg_get_state, L.scg_set_params, L.scg_get_params):
        fn.argtypes = [c_vp, C.POINTER(c_f64), C.c_int, C.c_int, c_vp]
    L.scg_set_counters.argtypes = [c_vp, C.POINTER(c_i32), C.POINTER(C.c_uint32), C.c_int, C.c_int, c_vp]
    L.scg_get_counters.argtypes = [c_vp, C.POINTER(c_i32), C.POINTER(C.c_uint32), C.c_int, C.c_int, c_vp]
    L.scg_prior_model.argtypes = [c_vp, c_vp, c_vp, C.c_int, C.c_double, c_vp, c_vp, c_vp, c_vp, c_vp]
    L.scg_gae.argtypes = [C.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int,
                          c_f64, c_f64, C.c_int, c_vp]
    L.scg_spec_source.argtypes = [C.POINTER(Config), C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(c_u64)]
    L.scg_spec_hash.restype = c_u64
    L.scg_source_hash.restype = c_u64
    L.scg_set_seed.argtypes = [c_vp, c_u64]
    L.scg_set_step_launch.argtypes = [c_vp, C.c_int, C.c_int]
    L.scg_set_step_wsback.argtypes = [c_vp, C.c_int, C.c_int]
    return L


def lib():
    """The generic library (any config).  Raises ScgError (never falls back) if it is missing or inconsistent.
    A library compiled from other kernel sources than the ones in the tree is rebuilt when hipcc is present,
    refused otherwise."""
    global _lib
    if _lib is None:
        if os.path.exists(LIB_PATH) and _lib_source_hash(LIB_PATH) != source_hash():
            if os.path.exists(_hipcc()):
                build(force=True)
            else:
                raise ScgError(f'{LIB_PATH} was built from different kernel sources and hipcc is not available to rebuild it')
        _lib = _bind(LIB_PATH)
    return _lib


# ---------------------------------------------------------------------------- config-specialised builds
def spec_source(cfg):
    """(source text, hash) of the specialisation header for a scg_config (generated by the library itself)."""
    L = lib()
    n, h = C.c_size_t(0), c_u64(0)
    check(L.scg_spec_source(C.byref(cfg), None, 0, C.byref(n), C.byref(h)))
    buf = C.create_string_buffer(n.value)
    check(L.scg_spec_source(C.byref(cfg), buf, n.value, C.byref(n), C.byref(h)))
    return buf.value.decode(), int(h.value)


POLICY_ACTS = {'tanh': 0, 'relu': 1, 'leaky_relu': 2}


def spec_paths(hash_value, policy=None):
    """(header, library) of a specialisation.  policy=(hidden, activation): the variant that also carries the fused
    policy-in-the-loop rollout kernel (scg_rollout_policy) for that actor shape.  Development variants: SCG_SPEC_TAG=<name>
    selects / builds libscg_spec_<hash>_<name>.so (compiled with the extra flags in SCG_SPEC_FLAGS)."""
    tag = f'{hash_value:016x}'
    var = os.environ.get('SCG_SPEC_TAG', '')
    pol = f'_pol{policy[0]}_{policy[1]}' if policy else ''
    return (os.path.join(SPEC_DIR, f'scg_spec_{tag}.h'),
            os.path.join(SPEC_DIR, f'libscg_spec_{tag}{pol}{"_" + var if var else ""}.so'))


def policy_supported(obs_dim, hidden, act_dim, activation):
    return 1 <= obs_dim <= 32 and hidden in (32, 64, 96, 128) and 1 <= act_dim <= 4 and activation in POLICY_ACTS


def sched_flags(cfg):
    """Machine-scheduler strategy per kernel family, measured on MI355X at 65 536 envs (one wave per SIMD: the step is an
    in-order issue chain, so instruction ORDER is time; same-box A/B in tools/sessions/s39.sh / s40.sh, profiles/r03_summary.md):
    Quadrotor2D float kernels gain 3 % from LLVM's max-ILP strategy (5.38 -> 5.22 us), the disturbed Quadrotor3D kernels 2.4 % from
    the iterative-ILP one (11.12 -> 10.86 us; it crashes the compiler on the 2-D kernels, hence the fallback in build_spec);
    CartPole (-2.6 %) keeps the default.  The plain Quadrotor3D kernels: round 3 / 4 measured +-0.5 % / -1.2 % and kept the default; on round 5's
    sources (21-bit reset draws, EnvOps::step = advance + evaluate) the default schedule came out 4 % SLOWER than round 4's kernel on the same
    box (8.60 vs 8.27 us) and max-ilp 2 % faster than it (8.04-8.16 us; tools/sessions/s111.sh, profiles/r05_step_kernel_ab.md section 5):
    max-ilp since round 5.  A list of alternatives, first that compiles."""
    has_dist = any(int(n) > 0 for n in cfg.n_dist) or int(cfg.adversary_channel) >= 0       # (= SCG_SPEC_DIST, scg_kernels.hip)
    if int(cfg.dtype) != F32:
        return [[]]
    if int(cfg.system) == QUAD_2D:
        return [['-mllvm', '-amdgpu-sched-strategy=max-ilp'], []]
    if int(cfg.system) == QUAD_3D and has_dist:
        return [['-mllvm', '-amdgpu-sched-strategy=iterative-ilp'], []]
    if int(cfg.system) == QUAD_3D:
        return [['-mllvm', '-amdgpu-sched-strategy=max-ilp'], []]
    return [[]]


def build_spec(cfg, force=False, verbose=False, policy=None):
    """Compile libscg_spec_<hash>.so: the same sources with this task config as compile-time constants
    (policy=(hidden, activation): + the fused policy rollout kernel for that actor shape)."""
    src, h = spec_source(cfg)
    hdr, so = spec_paths(h, policy)
    os.makedirs(SPEC_DIR, exist_ok=True)
    srcs = [os.path.join(CSRC_DIR, s) for s in SOURCES]
    if not force and os.path.exists(so) and _lib_source_hash(so) == source_hash():
        return so
    with open(hdr, 'w') as f:
        f.write(src)
    hipcc = _hipcc()
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-ffp-contract=on', '-std=c++17', '-fPIC', '-shared', '-DSCG_SPEC', '-include', hdr,
           f'-DSCG_SRC_HASH=0x{source_hash():016x}ULL', '-o', so] + os.environ.get('SCG_SPEC_FLAGS', '').split()
    if policy:
        cmd += [f'-DSCG_POLICY_H={int(policy[0])}', f'-DSCG_POLICY_ACT={POLICY_ACTS[policy[1]]}']
    alts = [[]] if os.environ.get('SCG_SPEC_FLAGS') else sched_flags(cfg)       # (explicit development flags: nothing added)
    res = None
    for extra in alts:
        full = cmd + extra + srcs
        if verbose:
            print(' '.join(full))
        res = subprocess.run(full, capture_output=True, text=True)
        if res.returncode == 0:
            return so
    raise ScgError('hipcc failed (specialised build):\n' + res.stdout + res.stderr)


def lib_for(cfg, specialize='auto', policy=None):
    """Library to drive an env with this config: the matching specialised build when it exists in-tree
    (or, with specialize=True, after compiling it now), else the generic library.  policy=(hidden, activation) asks for
    the variant with the fused policy rollout; it is compiled on demand (it cannot come from the generic library)."""
    if specialize in (False, 'off', None) and not policy:
        return lib(), False
    _, h = spec_source(cfg)
    key = (h, os.environ.get('SCG_SPEC_TAG', ''), tuple(policy) if policy else None)
    if key in _spec_libs:
        return _spec_libs[key], True
    _, so = spec_paths(h, policy)
    stale = os.path.exists(so) and _lib_source_hash(so) != source_hash()
    if not os.path.exists(so) or stale:
        if policy or specialize is True or specialize == 'build' or (stale and os.path.exists(_hipcc())):
            build_spec(cfg, force=True, policy=policy)     # (stale: the kernel sources changed since it was compiled)
        else:
            return lib(), False             # never run kernels of older sources: the generic library was checked by lib()
    L = _bind(so)
    if int(L.scg_spec_hash()) != h:
        raise ScgError(f'{so} was built for another config')
    _spec_libs[key] = L
    return L, True


def check(rc, L=None):
    if rc != 0:
        L = L if L is not None else lib()
        raise ScgError(f'libscg_hip error {rc}: {L.scg_last_error().decode()}')

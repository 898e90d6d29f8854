"""ctypes binding + builder of libscg_learn_<obs>_<hidden>_<act_dim>_<activation>.so (include/scg_learn.h): the PPO learner's
MFMA kernels, compiled per network shape from csrc/scg_learn.hip (hipcc cross-compiles without a GPU; ~5 s).

No fallback lives here: callers that cannot get a library for their shape (hidden size not a multiple of 32, more than four
action dimensions, unknown activation) use the PyTorch update path explicitly (ppo.py decides, and says so)."""
import ctypes as C
import os
import subprocess

from safe_control_gym_amd import _lib as L

ACTS = {'tanh': 0, 'relu': 1, 'leaky_relu': 2}
SRC = os.path.join(L.CSRC_DIR, 'scg_learn.hip')
DEPS = [SRC, os.path.join(L.CSRC_DIR, 'scg_adam.h'), os.path.join(L.CSRC_DIR, 'scg_mlp.h'), os.path.join(L.CSRC_DIR, 'scg_once.h'), os.path.normpath(os.path.join(L.CSRC_DIR, '..', '..', 'include', 'scg_learn.h'))]


class MlpLayout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3')]


class PpoGradArgs(C.Structure):
    _fields_ = [('d_params', C.c_void_p), ('actor', MlpLayout), ('critic', MlpLayout), ('logstd_off', C.c_int32),
                ('n_params', C.c_int32), ('d_obs', C.c_void_p), ('d_act', C.c_void_p), ('d_logp_old', C.c_void_p),
                ('d_adv', C.c_void_p), ('d_ret', C.c_void_p), ('d_v_old', C.c_void_p), ('d_idx', C.c_void_p),
                ('batch', C.c_int32), ('clip_param', C.c_float), ('entropy_coef', C.c_float),
                ('use_clipped_value', C.c_int32), ('n_workgroups', C.c_int32), ('d_workspace', C.c_void_p),
                ('d_grad', C.c_void_p), ('d_stats', C.c_void_p)]


def supported(obs_dim, hidden, act_dim, activation):
    return (1 <= obs_dim <= 31 and hidden % 32 == 0 and 32 <= hidden <= 128 and act_dim in (1, 2, 4) and activation in ACTS)


def source_hash():
    import hashlib
    h = hashlib.sha256()
    for p in DEPS:
        with open(p, 'rb') as f:
            h.update(os.path.basename(p).encode() + b'\0' + f.read())
    return int.from_bytes(h.digest()[:8], 'little')


def lib_path(obs_dim, hidden, act_dim, activation):
    """$SCG_LEARN_TAG selects a tagged variant library (built with $SCG_LEARN_FLAGS next to the shipped one: same-box A/B runs)."""
    tag = os.environ.get('SCG_LEARN_TAG', '')
    return os.path.join(L.SPEC_DIR, f'libscg_learn_{obs_dim}_{hidden}_{act_dim}_{activation}{"_" + tag if tag else ""}.so')


def build(obs_dim, hidden, act_dim, activation, force=False):
    if not supported(obs_dim, hidden, act_dim, activation):
        raise L.ScgError(f'no MFMA learner for obs {obs_dim} hidden {hidden} act {act_dim} {activation}')
    so = lib_path(obs_dim, hidden, act_dim, activation)
    if not force and os.path.exists(so) and L._lib_source_hash(so) == source_hash():
        return so
    os.makedirs(L.SPEC_DIR, exist_ok=True)
    cmd = [L._hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', f'-DSCG_L_NIN={obs_dim}',
           f'-DSCG_L_H={hidden}', f'-DSCG_L_NU={act_dim}', f'-DSCG_L_ACT={ACTS[activation]}',
           f'-DSCG_SRC_HASH=0x{source_hash():016x}ULL', '-o', so, SRC] + os.environ.get('SCG_LEARN_FLAGS', '').split()
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise L.ScgError('hipcc failed (learner build):\n' + res.stdout + res.stderr)
    return so


_libs = {}


def lib(obs_dim, hidden, act_dim, activation):
    """The learner library of this shape: in-tree build when present and current, compiled now when hipcc is available."""
    key = (obs_dim, hidden, act_dim, activation)
    if key in _libs:
        return _libs[key]
    so = lib_path(*key)
    if not os.path.exists(so) or L._lib_source_hash(so) != source_hash():
        if not os.path.exists(L._hipcc()):
            raise L.ScgError(f'{so} is missing or stale and hipcc is not available to build it')
        build(*key, force=True)
    D = C.CDLL(so)
    D.scg_learn_last_error.restype = C.c_char_p
    D.scg_ppo_grad_workspace_bytes.restype = C.c_size_t
    D.scg_ppo_grad_workspace_bytes.argtypes = [C.c_int]
    D.scg_mlp_forward.argtypes = [C.c_void_p, C.POINTER(MlpLayout), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    D.scg_ppo_grad.argtypes = [C.POINTER(PpoGradArgs), C.c_void_p]
    D.scg_ppo_step.argtypes = [C.POINTER(PpoGradArgs), C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                               C.c_void_p]
    D.scg_adam_gated.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float,
                                 C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    D.scg_adam_gated_scaled.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float,
                                        C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    D.scg_random_permutation.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_void_p]
    D.scg_random_permutation_keyed.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
    D.scg_ppo_returns_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
    D.scg_ppo_returns_scratch_bytes.restype = C.c_size_t
    D.scg_ppo_returns_moments.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    D.scg_ppo_returns_normalise.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    shape = [C.c_int32() for _ in range(4)]
    D.scg_learn_shape(*[C.byref(v) for v in shape])
    if tuple(v.value for v in shape) != (obs_dim, hidden, act_dim, ACTS[activation]):
        raise L.ScgError(f'{so} was built for another network shape')
    _libs[key] = D
    return D


def check(D, rc):
    if rc != 0:
        raise L.ScgError(f'libscg_learn error {rc}: {D.scg_learn_last_error().decode()}')

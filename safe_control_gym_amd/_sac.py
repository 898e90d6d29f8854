"""ctypes binding + builder of libscg_sac_<obs>_<hidden>_<act_dim>_<activation>.so (include/scg_sac.h): one fused gradient step
of SACAgent.update on the matrix cores, compiled per network shape from csrc/scg_sac.hip (hipcc cross-compiles without a
GPU; ~8 s).  No fallback lives here: sac.py uses the PyTorch update, visibly, for shapes this library does not serve."""
import ctypes as C
import os
import subprocess

from safe_control_gym_amd import _lib as L
from safe_control_gym_amd._learn import ACTS, MlpLayout

SRC = os.path.join(L.CSRC_DIR, 'scg_sac.hip')
DEPS = [SRC, os.path.join(L.CSRC_DIR, 'scg_adam.h'), os.path.join(L.CSRC_DIR, 'scg_mlp.h'), os.path.join(L.CSRC_DIR, 'scg_once.h'), os.path.join(L.CSRC_DIR, 'scg_rng.h'),
        os.path.normpath(os.path.join(L.CSRC_DIR, '..', '..', 'include', 'scg_sac.h')),
        os.path.normpath(os.path.join(L.CSRC_DIR, '..', '..', 'include', 'scg_learn.h'))]


class SacArgs(C.Structure):
    _fields_ = [('d_params', C.c_void_p), ('d_target', C.c_void_p), ('d_grad', C.c_void_p), ('d_m', C.c_void_p), ('d_v', C.c_void_p),
                ('d_steps', C.c_void_p), ('actor', MlpLayout), ('q1', MlpLayout), ('q2', MlpLayout), ('n_actor', C.c_int32),
                ('n_params', C.c_int32), ('d_obs', C.c_void_p), ('d_act', C.c_void_p), ('d_rew', C.c_void_p), ('d_next_obs', C.c_void_p),
                ('d_mask', C.c_void_p), ('d_ring_size', C.c_void_p), ('batch', C.c_int32), ('gamma', C.c_float), ('tau', C.c_float),
                ('actor_lr', C.c_float), ('critic_lr', C.c_float), ('entropy_lr', C.c_float), ('use_entropy_tuning', C.c_int32),
                ('target_entropy', C.c_float), ('act_low', C.c_float * 4), ('act_high', C.c_float * 4), ('seed', C.c_uint64),
                ('d_counter', C.c_void_p), ('d_idx_in', C.c_void_p), ('d_eps_in', C.c_void_p), ('d_eps_next_in', C.c_void_p),
                ('d_workspace', C.c_void_p), ('d_stats', C.c_void_p), ('d_stats_acc', C.c_void_p), ('phases', C.c_int32)]


class SacRing(C.Structure):
    _fields_ = [('d_obs', C.c_void_p), ('d_act', C.c_void_p), ('d_rew', C.c_void_p), ('d_next_obs', C.c_void_p), ('d_mask', C.c_void_p),
                ('capacity', C.c_int32), ('d_pos', C.c_void_p), ('d_size_f', C.c_void_p), ('d_size_i32', C.c_void_p), ('d_counter', C.c_void_p)]


ACTOR_GRAD, CRITIC_GRAD, FINISH, ALL = 1, 2, 4, 7


def supported(obs_dim, hidden, act_dim, activation):
    return (1 <= act_dim <= 4 and obs_dim >= 1 and obs_dim + act_dim < 32 and hidden % 32 == 0 and 32 <= hidden <= 128
            and activation in ACTS)


def source_hash():
    import hashlib
    h = hashlib.sha256()
    for p in DEPS:
        with open(p, 'rb') as f:
            h.update(os.path.basename(p).encode() + b'\0' + f.read())
    return int.from_bytes(h.digest()[:8], 'little')


def lib_path(obs_dim, hidden, act_dim, activation):
    return os.path.join(L.SPEC_DIR, f'libscg_sac_{obs_dim}_{hidden}_{act_dim}_{activation}.so')


def build(obs_dim, hidden, act_dim, activation, force=False):
    if not supported(obs_dim, hidden, act_dim, activation):
        raise L.ScgError(f'no fused SAC update for obs {obs_dim} hidden {hidden} act {act_dim} {activation}')
    so = lib_path(obs_dim, hidden, act_dim, activation)
    if not force and os.path.exists(so) and L._lib_source_hash(so) == source_hash():
        return so
    os.makedirs(L.SPEC_DIR, exist_ok=True)
    cmd = [L._hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', f'-DSCG_S_NOBS={obs_dim}', f'-DSCG_S_H={hidden}',
           f'-DSCG_S_NU={act_dim}', f'-DSCG_S_ACT={ACTS[activation]}', f'-DSCG_SRC_HASH=0x{source_hash():016x}ULL', '-o', so, SRC] \
        + os.environ.get('SCG_SAC_FLAGS', '').split()
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise L.ScgError('hipcc failed (SAC build):\n' + res.stdout + res.stderr)
    return so


_libs = {}


def lib(obs_dim, hidden, act_dim, activation):
    key = (obs_dim, hidden, act_dim, activation)
    if key in _libs:
        return _libs[key]
    so = lib_path(*key)
    if not os.path.exists(so) or L._lib_source_hash(so) != source_hash():
        if not os.path.exists(L._hipcc()):
            raise L.ScgError(f'{so} is missing or stale and hipcc is not available to build it')
        build(*key, force=True)
    D = C.CDLL(so)
    D.scg_sac_last_error.restype = C.c_char_p
    D.scg_sac_workspace_bytes.restype = C.c_size_t
    D.scg_sac_workspace_bytes.argtypes = [C.c_int]
    D.scg_sac_update.argtypes = [C.POINTER(SacArgs), C.c_void_p]
    D.scg_sac_update_n.argtypes = [C.POINTER(SacArgs), C.c_int, C.c_void_p]
    D.scg_sac_act.argtypes = [C.c_void_p, C.POINTER(MlpLayout), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p, C.c_int, C.c_void_p,
                              C.c_void_p]
    D.scg_sac_sample.argtypes = [C.c_void_p, C.POINTER(MlpLayout), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p, C.c_int, C.c_uint64,
                                 C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    D.scg_sac_push.argtypes = [C.POINTER(SacRing), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    shape = [C.c_int32() for _ in range(4)]
    D.scg_sac_shape(*[C.byref(v) for v in shape])
    if tuple(v.value for v in shape) != (obs_dim, hidden, act_dim, ACTS[activation]):
        raise L.ScgError(f'{so} was built for another network shape')
    _libs[key] = D
    return D


def check(D, rc):
    if rc != 0:
        raise L.ScgError(f'libscg_sac error {rc}: {D.scg_sac_last_error().decode()}')

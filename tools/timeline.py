#!/usr/bin/env python3
"""In-kernel timeline of the specialised Q2-track step kernel (timing-only build with -DSCG_EXP_TIMELINE).

  python tools/timeline.py build   # here: compile the variant next to the shipped specialised library
  python tools/timeline.py run     # on the GPU box: run steps, read back the per-wave shader-clock marks
"""
import ctypes as C, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == 'run':
    import torch        # BEFORE anything that may load a HIP library: torch's bundled runtime must initialise first (s84 / s85: the run
    torch.cuda.init()   # died in hipSetDevice with "no ROCm-capable device is detected" when _lib was imported ahead of torch)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPEC = os.path.join(ROOT, 'safe_control_gym_amd', 'spec')


def spec_hash():
    from safe_control_gym_amd import _lib as L
    from safe_control_gym_amd.env_config import EnvSpec
    from safe_control_gym_amd.registration import load_task
    env_id, cfg = load_task('quadrotor_2D_track')
    c, _ = EnvSpec(env_id, cfg).to_c_config(64, L.F32, 1)
    return '%016x' % L.spec_source(c)[1]


H = spec_hash()
VARIANT = os.path.join(SPEC, 'exp', 'TIMELINE.so')
MARKS = ['entry', 'kernargs', 'state loaded', 'integrated', 'reward/done', 'constraints', 'obs+stats stored', 'end']

if sys.argv[1] == 'hash':
    print(H)
elif sys.argv[1] == 'build':
    os.makedirs(os.path.dirname(VARIANT), exist_ok=True)
    from safe_control_gym_amd import _lib as L          # the shipped build's flags: source-hash stamp (checked at load) and scheduler strategy
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-ffp-contract=on', '-std=c++17', '-fPIC', '-shared', '-w', '-DSCG_SPEC',
                           '-DSCG_EXP_TIMELINE', f'-DSCG_SRC_HASH=0x{L.source_hash():016x}ULL', '-mllvm', '-amdgpu-sched-strategy=max-ilp']
                          + sys.argv[2:] + ['-include', f'{SPEC}/scg_spec_{H}.h', '-o', VARIANT,
                                            f'{ROOT}/safe_control_gym_amd/csrc/scg_kernels.hip'])
else:
    import shutil, numpy as np, torch
    real = f'{SPEC}/libscg_spec_{H}.so'
    shutil.copy(real, '/tmp/keep.so'); shutil.copy(VARIANT, real)
    try:
        from safe_control_gym_amd.registration import load_task
        from safe_control_gym_amd.vec_env import HipVecEnv
        env_id, cfg = load_task('quadrotor_2D_track')
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
        env = HipVecEnv(env_id, n, seed=1, return_numpy=False, **cfg)
        env.bind_outputs(state=None, noisy_action=None)
        mode = sys.argv[3] if len(sys.argv) > 3 else 'default'          # default (the library's thresholds) | plain | split
        NEVER = 2 ** 31 - 1
        if mode != 'default':
            env.set_step_launch(*{'plain': (0, NEVER), 'split': (NEVER, NEVER)}[mode])
        split = mode == 'split' or (mode == 'default' and n <= 32768)
        pair = False
        env.reset_tensors()
        acts = [torch.rand(n, 2, device='cuda') * 2 - 1 for _ in range(16)]
        L = env._lib
        L.scg_exp_timeline.argtypes = [C.c_void_p, C.c_size_t]
        rows = []
        for it in range(300):
            env.step_tensors(acts[it % 16])
            if it >= 100:
                torch.cuda.synchronize()
                buf = np.zeros(4096 * 8, dtype=np.uint64)
                assert L.scg_exp_timeline(buf.ctypes.data, buf.size) == 0
                groups = max(1, n // 64)
                n_waves = 2 * ((groups + 7) // 8 * 8) if split else (2 * groups if pair else groups)   # split: workgroup b = role (b / 8) % 2
                t = buf.reshape(4096, 8)[:min(n_waves, 4096)].astype(np.int64)
                rows.append(t)
        t_all = np.stack(rows)                            # [iters][waves][marks]; XCD clocks are not mutually synchronised
        wave = np.arange(t_all.shape[1])
        role_state = (((wave >> 3) & 1) == 1) if split else (((wave >> 2) & 1) == 1)     # pair: waves 4-7 of each 8-wave workgroup
        roles = [('every output (one wave per 64 envs)', np.ones_like(wave, dtype=bool))] if not (split or pair) else \
            [('ROLE_SCORE (reward / done / constraint rows / statistics)' + (' + advance' if pair else ''), ~role_state),
             ('ROLE_STATE (observation / auto-reset / state)', role_state)]
        for name, sel in roles:
            t = t_all[:, sel, :]
            d = np.diff(t, axis=2).reshape(-1, 7)             # per-wave phase durations
            tot = (t[:, :, 7] - t[:, :, 0]).reshape(-1)
            print(f'{n} envs, {name}: {t.shape[1]} waves, {t.shape[0]} launches; shader-clock ticks per phase (per wave)')
            print(f'  {"phase":34s} {"mean":>7s} {"p10":>7s} {"p50":>7s} {"p90":>7s} {"share":>6s}')
            for k in range(7):
                col = d[:, k]
                print(f'  {MARKS[k] + " -> " + MARKS[k + 1]:34s} {col.mean():7.0f} {np.percentile(col, 10):7.0f} {np.median(col):7.0f} '
                      f'{np.percentile(col, 90):7.0f} {col.mean() / tot.mean():6.1%}')
            print(f'  {"entry -> end":34s} {tot.mean():7.0f} {np.percentile(tot, 10):7.0f} {np.median(tot):7.0f} {np.percentile(tot, 90):7.0f}')
        sys.stdout.flush(); shutil.copy('/tmp/keep.so', real); os._exit(0)      # (os._exit skips the finally block)
    finally:
        shutil.copy('/tmp/keep.so', real)

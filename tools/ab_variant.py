#!/usr/bin/env python3
"""Same-box A/B of a kernel variant against the shipped libraries in ~20 GPU-seconds (what tools/sessions/s79.sh did by hand).

  build (here, no GPU):   python tools/ab_variant.py build <tag> [--src DIR] [--flags=-DSCG_EXP_...] [--tasks t1,t2|all]
      compiles libscg_spec_<hash>_<tag>.so for the shipped task configs (float32) from the sources in DIR (default: the tree's
      csrc — pass a scratch copy to leave the product untouched) with the extra flags, stamped with the tree's source hash so that
      `SCG_SPEC_TAG=<tag>` loads it, and runs the store-hazard lint on it;
  run (GPU box):          python tools/ab_variant.py run <tag> [--tasks ...] [--rounds 2] [--envs 65536,262144] [--no-gate]
      alternates `bench.py --task T` (headline only) with SCG_SPEC_TAG=<tag> and without, prints the launch periods, then the float32
      one-step errors of the variant against the oracle (512 envs x 60 re-synchronised steps) — the quick gate before the full suite;
  clean:                  python tools/ab_variant.py clean <tag>
"""
import argparse
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TASKS = ('quadrotor_2D_track', 'cartpole_stab', 'quadrotor_3D_track', 'quadrotor_3D_track_disturbed')


def build(tag, src, flags, tasks, sched=None):
    from safe_control_gym_amd import _lib
    from safe_control_gym_amd.env_config import EnvSpec
    from safe_control_gym_amd.registration import load_task
    src = os.path.abspath(src or _lib.CSRC_DIR)
    for task in tasks:
        env_id, cfg = load_task(task)
        c, _ = EnvSpec(env_id, cfg).to_c_config(1, _lib.F32, 0)
        text, h = _lib.spec_source(c)
        hdr, so = _lib.spec_paths(h)
        os.makedirs(_lib.SPEC_DIR, exist_ok=True)
        with open(hdr, 'w') as f:
            f.write(text)
        out = so[:-3] + f'_{tag}.so'
        res = None
        for sched in ([[] if sched == 'default' else ['-mllvm', f'-amdgpu-sched-strategy={sched}']] if sched else _lib.sched_flags(c)):
            cmd = [_lib._hipcc(), '--offload-arch=gfx950', '-O3', '-ffp-contract=on', '-std=c++17', '-fPIC', '-shared', '-DSCG_SPEC', '-include', hdr,
                   f'-DSCG_SRC_HASH=0x{_lib.source_hash():016x}ULL', '-o', out] + flags.split() + sched + [os.path.join(src, s) for s in _lib.SOURCES]
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode == 0:
                break
        if res.returncode:
            sys.exit(f'{task}: hipcc failed\n{res.stderr[-2000:]}')
        subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'hazard_lint.py'), out], check=True)
        print(f'{task}: {os.path.basename(out)}')


def run(tag, tasks, rounds, envs=(65536,), gate=True):
    base = ['--steps', '4000', '--warmup', '500', '--no-secondary', '--no-cpu-baseline', '--ppo-seeds', '0', '--sac-seeds', '0']
    for task, n_envs in ((t, e) for t in tasks for e in envs):
        for _ in range(rounds):
            for t in (tag, ''):
                env = dict(os.environ, SCG_SPEC_TAG=t)
                out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--task', task, '--envs', str(n_envs)] + base, env=env,
                                     capture_output=True, text=True).stdout
                d = json.loads(out.strip().splitlines()[-1])
                print(f'{task:32s} {n_envs:8d} envs tag=[{t:8s}] {d["roofline"]["avg_launch_us"]:.4f} us  frac {d["roofline"]["frac"] or 0:.4f}  '
                      f'{d["config"]["kernel_build"]}  finite={d["config"]["finite_outputs"]}', flush=True)
    if not gate:
        return
    os.environ['SCG_SPEC_TAG'] = tag
    import numpy as np
    import torch
    from oracle.envs import make_oracle_env, make_rng
    from oracle.vec import OracleVecEnv
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    from tests.test_gpu_env_parity import _params, _raw_state
    for task in tasks:
        env_id, cfg = load_task(task)
        n, seed = 512, 3
        gpu = HipVecEnv(env_id, n, seed=seed, dtype=torch.float32, return_numpy=False, specialize=True, **cfg)
        o = make_oracle_env(env_id, n, make_rng('philox', n, seed), **cfg)
        ov = OracleVecEnv(o)
        ov.reset(); gpu.reset_tensors()
        rng = np.random.default_rng(0)
        worst, bad, worst_c, worst_o, worst_r = np.zeros(o.state_dim), 0, 0.0, 0.0, 0.0
        for _ in range(60):
            gpu.set_raw_state(_raw_state(o)); gpu.set_counters(o.ctrl_step_counter, o.episode)
            if o.RANDOMIZED_INERTIAL_PROP:
                gpu.set_params(_params(o))
            act = rng.uniform(-1, 1, (n, o.action_dim))
            obs_o, rew_o, done_o, info = ov.step(act)
            out = gpu.step_tensors(torch.as_tensor(act, dtype=torch.float32, device=gpu.device))
            same = out.done.cpu().numpy().astype(bool) == done_o
            bad += int((~same).sum())
            keep = same & ~done_o
            worst = np.maximum(worst, np.abs(out.state.cpu().numpy().T[keep] - o.state[keep]).max(axis=0))
            worst_o = max(worst_o, float(np.abs(out.obs.cpu().numpy()[keep] - obs_o[keep]).max()))
            worst_r = max(worst_r, float(np.abs(out.reward.cpu().numpy() - rew_o).max()))
            if 'constraint_values' in info and out.c_values is not None:      # every row of every env (also the finished episodes')
                worst_c = max(worst_c, float(np.abs(out.c_values.cpu().numpy().T - info['constraint_values']).max()))
        print(f'{task:32s} variant float32 one-step max |error|: state {worst.max():.2e}, obs {worst_o:.2e}, reward {worst_r:.2e}, '
              f'constraint values {worst_c:.2e} (tolerances of the parity tests: 2e-5 / 2e-5 / 3e-5 / 5e-5 x scale), done mismatches {bad}, '
              f'specialised {gpu.specialized}')
        gpu.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('cmd', choices=['build', 'run', 'clean'])
    ap.add_argument('tag')
    ap.add_argument('--src', default=None)
    ap.add_argument('--flags', default='')
    ap.add_argument('--tasks', default=TASKS[0])
    ap.add_argument('--sched', default=None, help="build: machine-scheduler strategy of the variant instead of _lib.sched_flags' choice ('default' = LLVM's)")
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--envs', default='65536', help='comma-separated env counts for the launch-period A/B')
    ap.add_argument('--no-gate', action='store_true', help='skip the one-step parity gate (e.g. on a second call for another env count)')
    a = ap.parse_args()
    tasks = TASKS if a.tasks == 'all' else tuple(a.tasks.split(','))
    if a.cmd == 'build':
        build(a.tag, a.src, a.flags, tasks, a.sched)
    elif a.cmd == 'run':
        run(a.tag, tasks, a.rounds, [int(e) for e in a.envs.split(',')], not a.no_gate)
    else:
        from safe_control_gym_amd import _lib
        for p in glob.glob(os.path.join(_lib.SPEC_DIR, f'*_{a.tag}.so')):
            os.remove(p)
            print('removed', os.path.basename(p))


if __name__ == '__main__':
    main()

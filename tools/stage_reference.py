#!/usr/bin/env python3
"""Stage the reference's Python package as UNTRACKED scratch for LOCAL checker runs (rounds 3-5 let it travel to the gpurun box; since
round 6 it does not: `.gpurunignore` lists oracle/_ref/reference/ — the reference's Python stays in this container).

    python tools/stage_reference.py [--reference /root/reference] [--dest oracle/_ref/reference]

`oracle/_ref/` is git-ignored (never in history); the built oracle library in it travels with the snapshot, this copy does not.
Only the checker side reads it — tests/test_gpu_dropin.py (the reference's own PPO / SAC classes on HipVecEnv),
tools/run_reference_ppo_on_hip.py, tests/golden/pybullet_probe.py — through tests/golden/ref_stubs.py::reference_root()
($SCG_REFERENCE_ROOT, /root/reference, oracle/_ref/reference in that order).  Nothing under safe_control_gym_amd/ or in
bench.py's timed region imports it (tests/test_capi_cpu.py::test_product_never_imports_the_oracle).
Copied: safe_control_gym/**/*.{py,yaml,urdf,obj,dae}, examples/rl/config_overrides/**, examples/lqr/**, examples/pid/** and examples/no_controller/** (scripts +
overrides), examples/rl/rl_experiment.py and the shipped checkpoints examples/rl/models/{ppo,sac,safe_explorer_ppo}/*.pt (21 MB) that
tools/run_reference_example.py (`rl`, `matrix`) evaluates; nothing else.
"""
import argparse
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = ('.py', '.yaml', '.urdf', '.obj', '.dae')


def stage(reference='/root/reference', dest=None, verbose=True):
    dest = dest or os.path.join(ROOT, 'oracle', '_ref', 'reference')
    if not os.path.isdir(os.path.join(reference, 'safe_control_gym')):
        if verbose:
            print(f'[stage_reference] no checkout at {reference}: nothing staged')
        return None
    if os.path.isdir(dest):
        shutil.rmtree(dest)
    n = 0
    for sub in ('safe_control_gym', os.path.join('examples', 'rl', 'config_overrides'), os.path.join('examples', 'lqr'), os.path.join('examples', 'pid'),
                os.path.join('examples', 'no_controller')):
        for d, _, files in os.walk(os.path.join(reference, sub)):
            for f in files:
                if not f.endswith(KEEP):
                    continue
                src = os.path.join(d, f)
                out = os.path.join(dest, os.path.relpath(src, reference))
                os.makedirs(os.path.dirname(out), exist_ok=True)
                shutil.copyfile(src, out)
                n += 1
    # the RL example script and the shipped checkpoints tools/run_reference_example.py evaluates on the GPU box
    rels = [os.path.join('examples', 'rl', 'rl_experiment.py')]
    for alg in ('ppo', 'sac', 'safe_explorer_ppo'):
        d = os.path.join(reference, 'examples', 'rl', 'models', alg)
        rels += [os.path.join('examples', 'rl', 'models', alg, f) for f in (sorted(os.listdir(d)) if os.path.isdir(d) else []) if f.endswith('.pt')]
    for rel in rels:
        src = os.path.join(reference, rel)
        if os.path.isfile(src):
            out = os.path.join(dest, rel)
            os.makedirs(os.path.dirname(out), exist_ok=True)
            shutil.copyfile(src, out)
            n += 1
    if verbose:
        print(f'[stage_reference] {n} files -> {dest}')
    return dest


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', default='/root/reference')
    ap.add_argument('--dest', default=None)
    a = ap.parse_args()
    sys.exit(0 if stage(a.reference, a.dest) else 1)

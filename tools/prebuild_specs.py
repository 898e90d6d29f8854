#!/usr/bin/env python3
"""Compile (here, on CPU — hipcc cross-compiles) the config-specialised libraries the GPU tests ask for, so that the
GPU box finds them in the snapshot instead of spending GPU-minutes in hipcc: every golden rollout config in both dtypes,
plus their `integrator: rk4` variants where the test uses them.  usage: prebuild_specs.py [jobs]"""
import glob
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def configs():
    import numpy as np
    out = []
    for p in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'rollout_*.npz'))):
        meta = json.loads(str(np.load(p)['meta_json']))
        cfg = dict(meta['config'])
        cfg.pop('seed', None)
        out.append((meta['task'], cfg))
    # variants of the shipped task YAMLs that tests/test_gpu_sequence.py and test_gpu_parity_scale.py step
    from safe_control_gym_amd.registration import load_task
    env_id, q2 = load_task('quadrotor_2D_track')
    out += [(env_id, dict(q2, obs_goal_horizon=3)), (env_id, dict(q2, episode_len_sec=0.3))]
    for task in ('quadrotor_2D_track', 'cartpole_stab', 'quadrotor_3D_track'):
        env_id, c = load_task(task)
        out.append((env_id, dict(c, randomized_init=True)))
    # tests/test_gpu_step_launch.py
    from tests.test_gpu_step_launch import CASES as SPLIT_CASES
    for task, over in SPLIT_CASES:
        env_id, c = load_task(task)
        out.append((env_id, dict(c, **over, _no_rk4=True)))
    # tests/test_gpu_config_fuzz.py: random configs whose specialised builds are tested (SPEC_SEEDS per system, both dtypes)
    from tests.config_fuzz import SYSTEMS, fuzz_config
    from tests.test_gpu_config_fuzz import SPEC_SEEDS
    for system in SYSTEMS:
        for seed in SPEC_SEEDS:
            env_id, c = fuzz_config(system, seed)
            out.append((env_id, dict(c, _no_rk4=True)))
    return out


def main():
    jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    from safe_control_gym_amd import _lib
    from safe_control_gym_amd.env_config import EnvSpec
    _lib.lib()
    todo = {}
    for task, cfg in configs():
        no_rk4 = cfg.pop('_no_rk4', False)
        variants = [cfg]
        if not no_rk4 and not cfg.get('disturbances') and not cfg.get('adversary_disturbance'):
            variants.append(dict(cfg, integrator='rk4'))
        for c in variants:
            for dt in (_lib.F32, _lib.F64):
                try:
                    cc, _ = EnvSpec(task, dict(c)).to_c_config(1, dt, 0)
                except Exception:                               # noqa: BLE001  (a variant the config layer rejects)
                    continue
                _, h = _lib.spec_source(cc)
                todo[h] = cc
    with ThreadPoolExecutor(jobs) as ex:
        for so in ex.map(_lib.build_spec, todo.values()):
            print(os.path.basename(so), flush=True)
    print(f'{len(todo)} specialisations present')


if __name__ == '__main__':
    main()

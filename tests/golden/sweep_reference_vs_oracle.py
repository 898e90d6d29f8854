#!/usr/bin/env python3
"""Live sweep (build container only: needs /root/reference): the REFERENCE's own Python against the oracle on tests/config_fuzz.py
configs beyond the committed rollout_fuzz_* fixtures — nothing is written, every case is generated into a temp dir, replayed by
tests/test_oracle_golden.py's checker and deleted.  Last runs (round 3): seeds 3..18 and 19..50 of the four systems = 192 configs, 191 reproduced to
1e-9; quadrotor_2D seed 3 makes the REFERENCE raise (AttributeError: 'Quadrotor' object has no attribute 'out_of_bounds' — its _get_info
reads the attribute _get_done only sets when it does not return early on goal_reached, quadrotor.py:871-892, here on an episode's first
step; the oracle and the kernels report False there).

    python tests/golden/sweep_reference_vs_oracle.py [first_seed last_seed]
"""
import os
import sys
import tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.golden import make_golden
from tests.config_fuzz import SYSTEMS, fuzz_config
import tests.test_oracle_golden as T
tmp = tempfile.mkdtemp()
make_golden.HERE = tmp
T.GOLDEN = tmp
bad = []
for system in SYSTEMS:
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 2 else 3, int(sys.argv[2]) + 1 if len(sys.argv) > 2 else 19):
        env_id, cfg = fuzz_config(system, seed)
        cfg = dict(cfg, seed=900 + seed)
        name = f'fuzz_{system}_{seed}'
        try:
            make_golden.rollout_case(name, env_id, cfg, n_envs=3, n_steps=70, seed=900 + seed, act_scale=0.7, act_seed=40 + seed,
                                     adversary=cfg.get('adversary_disturbance') is not None)
        except Exception as e:
            bad.append((name, 'REFERENCE raised', repr(e)[:200])); continue
        try:
            T.test_oracle_reproduces_reference_rollout(name)
        except Exception as e:
            bad.append((name, 'MISMATCH', str(e)[:600]))
        os.remove(os.path.join(tmp, f'rollout_{name}.npz'))
print('\n==== bad:', len(bad))
for b in bad: print(b)

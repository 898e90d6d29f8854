#!/usr/bin/env python3
"""More golden rollouts from the REFERENCE's own Python (protocol of make_golden.py), on configs drawn by tests/config_fuzz.py
instead of hand-picked ones: the reference's BenchmarkEnv / CartPole / Quadrotor / constraints / disturbances / DummyVecEnv /
VecRecordEpisodeStatistics run the random config, tests/test_oracle_golden.py makes the oracle reproduce the rollout, and every
GPU test that walks tests/golden/rollout_*.npz (free-running float64, float32 one-step, fixture replay; generic and specialised
libraries) picks the new cases up.

    python tests/golden/make_golden_fuzz.py        (build container only: needs /root/reference) -> rollout_fuzz_<system>_<seed>.npz
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden import make_golden  # noqa: E402  (installs the stubs, imports the reference)
from tests.config_fuzz import SYSTEMS, fuzz_config  # noqa: E402

SEEDS = (0, 1, 2)


def main():
    for system in SYSTEMS:
        for seed in SEEDS:
            env_id, cfg = fuzz_config(system, seed)
            cfg = dict(cfg, seed=900 + seed)
            make_golden.rollout_case(f'fuzz_{system}_{seed}', env_id, cfg, n_envs=3, n_steps=90, seed=900 + seed, act_scale=0.7,
                                     act_seed=40 + seed, adversary=cfg.get('adversary_disturbance') is not None)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Known answers of the reference's MetricExtractor (experiments/base_experiment.py:380-492 +
math_and_models/metrics/performance_metrics.py:6-31), produced by running the reference's own class.

    python tests/golden/make_metrics.py             (build container only: needs /root/reference)

Three synthetic evaluation runs (episodes of different lengths with per-step reward / mse / constraint_violation, stored as
BaseExperiment's trajectory dict stores them) go through MetricExtractor.compute_metrics; the per-episode totals this repo's
device-side accumulators hold (return, length, violation steps, sum of mse) and the reference's metrics go to
tests/golden/metrics.npz.  tests/test_metrics_golden.py makes safe_control_gym_amd.ppo.episode_metrics reproduce them.
"""
import os
import sys
from collections import defaultdict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden import ref_stubs  # noqa: E402

ref_stubs.install()

from safe_control_gym.experiments.base_experiment import MetricExtractor  # noqa: E402

KEYS = ('average_length', 'average_return', 'average_rmse', 'rmse_std', 'worst_case_rmse_at_0.5', 'failure_rate',
        'average_constraint_violation', 'constraint_violation_std')


def main():
    out = {}
    rng = np.random.default_rng(11)
    for case, n_ep in (('one', 1), ('three', 3), ('many', 40)):
        data = defaultdict(list)
        tot = np.zeros((n_ep, 4))
        for e in range(n_ep):
            L = int(rng.integers(5, 250))
            rew = rng.uniform(0.2, 1.0, L)
            mse = rng.uniform(0.0, 0.5, L) ** 2
            viol = (rng.uniform(size=L) < (0.1 if e % 3 == 0 else 0.0)).astype(int)
            # trajectory dict as BaseExperiment._execute_evaluations / RecordDataWrapper fill it (base_experiment.py:340-375):
            # per-episode arrays for reward / length, per-episode lists of step info dicts for mse / constraint_violation
            data['reward'].append(rew)
            data['length'].append(np.ones(L, dtype=int))
            data['info'].append([{}] + [{'mse': float(mse[t]), 'constraint_violation': int(viol[t])} for t in range(L)])
            tot[e] = rew.sum(), L, viol.sum(), mse.sum()
        m = MetricExtractor().compute_metrics(dict(data))
        out[f'{case}/totals'] = tot
        out[f'{case}/metrics'] = np.array([float(m[k]) for k in KEYS])
    out['keys'] = np.array(KEYS)
    np.savez(os.path.join(HERE, 'metrics.npz'), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()

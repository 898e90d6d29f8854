#!/usr/bin/env python3
"""BASELINE config #1 with the reference's own example script: examples/lqr/lqr_experiment.py — its ConfigFactory, YAML overrides
(config_overrides/cartpole/{cartpole_stab,lqr_cartpole_stab}.yaml), registry, `LQR` (or `iLQR`) controller, `BaseExperiment`,
`RecordDataWrapper`, `MetricExtractor` — run UNMODIFIED, with the ONE change INTEGRATION.md describes: the registry's `cartpole` id points
at this package's facade (`safe_control_gym_amd.benchmark_env:CartPole`) instead of the PyBullet env.

    python tools/run_reference_lqr_example.py [--algo lqr|ilqr] [--stub-handle]

Needs the reference checkout (build container: /root/reference; GPU box: the scratch copy tools/stage_reference.py stages) and runs it
under tests/golden/ref_stubs.py (stand-ins for gymnasium / casadi / pybullet / munch / dict_deep, all absent in this image).
--stub-handle: no GPU — the facade's batch-of-1 handle is the oracle-backed stand-in of the CPU suite (tests/test_facade_cpu.py);
without it the handle is the real HipVecEnv.  Prints the example's own "FINAL METRICS" line.
"""
import argparse
import importlib.util
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Munch(dict):
    """munch.Munch for ConfigFactory.merge (utils/configuration.py:92): attribute access on nested dicts."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    __setattr__ = dict.__setitem__


def munchify(x):
    if isinstance(x, dict):
        return Munch({k: munchify(v) for k, v in x.items()})
    return [munchify(v) for v in x] if isinstance(x, list) else x


def deep_set(d, key, value):                        # dict_deep.deep_set for --kv_overrides
    ks = key.split('.')
    for k in ks[:-1]:
        d = d.setdefault(k, {})
    d[ks[-1]] = value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--algo', default='lqr', choices=['lqr', 'ilqr'])
    ap.add_argument('--stub-handle', action='store_true')
    a = ap.parse_args()
    from tests.golden import ref_stubs
    ref = ref_stubs.reference_root()
    if ref is None:
        sys.exit('no reference checkout on this machine')
    ref_stubs.install()
    sys.modules['munch'].munchify, sys.modules['munch'].Munch = munchify, Munch
    sys.modules['dict_deep'].deep_set = deep_set
    import matplotlib
    matplotlib.use('Agg')
    import safe_control_gym.envs  # noqa: F401  (registers the reference's env ids)
    from safe_control_gym.utils.registration import register, registry
    # ---- the one-line change of INTEGRATION.md -------------------------------------------------------------------------
    registry.specs['cartpole'].entry_point = 'safe_control_gym_amd.benchmark_env:CartPole'
    # ----------------------------------------------------------------------------------------------------------------------
    cls = {'lqr': 'lqr:LQR', 'ilqr': 'ilqr:iLQR'}[a.algo]       # (controllers/__init__.py registers every controller, MPC's casadi / gpytorch included)
    register(idx=a.algo, entry_point=f'safe_control_gym.controllers.lqr.{cls}', config_entry_point=f'safe_control_gym.controllers.lqr:{a.algo}.yaml')
    if a.stub_handle:
        import safe_control_gym_amd.benchmark_env as B
        from tests.test_facade_cpu import _OracleBackedVec
        B.HipVecEnv = _OracleBackedVec
    ov = os.path.join(ref, 'examples', 'lqr', 'config_overrides', 'cartpole')
    sys.argv = ['lqr_experiment.py', '--algo', a.algo, '--task', 'cartpole', '--overrides', os.path.join(ov, 'cartpole_stab.yaml'),
                os.path.join(ov, f'{a.algo}_cartpole_stab.yaml')]
    spec = importlib.util.spec_from_file_location('lqr_experiment', os.path.join(ref, 'examples', 'lqr', 'lqr_experiment.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        mod.run(gui=False, plot=False, n_episodes=1, n_steps=None, save_data=False)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Live sweep (build container only: needs /root/reference): the reference's own `env.symbolic` — its CasADi expressions evaluated by
tests/golden/casadi_numeric.py — against this package's AnalyticModel on RANDOM model parameters: `prior_prop` (the prior model's own
mass / inertia / pole parameters, cartpole.py:390-401, quadrotor.py:476-483), `inertial_prop` overrides, control frequency (dt),
quad type; f, df/dx, df/du, the equilibrium, one rk_discrete step, at random (x, u).  Nothing is written.

    python tests/golden/sweep_symbolic.py [n_cases_per_system]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden import make_symbolic as S  # noqa: E402  (installs the stubs, imports the reference)

from safe_control_gym_amd.env_config import EnvSpec  # noqa: E402
from safe_control_gym_amd.symbolic import AnalyticModel  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(5)
    bad = []
    for name, (task, override, extra) in S.CASES.items():
        for case in range(n_cases):
            cfg = S.load_task_config(task, override, extra)
            cfg.pop('seed', None)
            cfg['cost'] = 'quadratic'
            ctrl, pyb = [(15, 750), (50, 1000), (25, 500), (60, 240), (100, 1000)][int(rng.integers(5))]
            cfg.update(ctrl_freq=ctrl, pyb_freq=pyb)
            if task == 'cartpole':
                prior = {'pole_length': float(rng.uniform(0.3, 0.8)), 'cart_mass': float(rng.uniform(0.5, 2.0)), 'pole_mass': float(rng.uniform(0.05, 0.3))}
                inertial = {'pole_length': float(rng.uniform(0.3, 0.8)), 'cart_mass': float(rng.uniform(0.5, 2.0)), 'pole_mass': float(rng.uniform(0.05, 0.3))}
            else:
                prior = {'M': float(rng.uniform(0.02, 0.05)), 'Iyy': float(rng.uniform(1e-5, 3e-5)), 'Ixx': float(rng.uniform(1e-5, 3e-5)),
                         'Izz': float(rng.uniform(1.5e-5, 4e-5))}
                inertial = {'M': float(rng.uniform(0.02, 0.05)), 'Iyy': float(rng.uniform(1e-5, 3e-5)), 'Ixx': float(rng.uniform(1e-5, 3e-5)),
                            'Izz': float(rng.uniform(1.5e-5, 4e-5))}
            which = int(rng.integers(3))
            if which == 0:
                cfg['prior_prop'] = prior
            elif which == 1:
                cfg['inertial_prop'] = inertial
            else:
                cfg['prior_prop'], cfg['inertial_prop'] = prior, inertial
            tag = f'{name} case {case} ({ctrl}/{pyb} Hz, ' + ('prior_prop' if which == 0 else 'inertial_prop' if which == 1 else 'both') + ')'
            try:
                env = {'cartpole': S.CartPole, 'quadrotor': S.Quadrotor}[task](**dict(cfg, output_dir='/tmp'))
                # upstream builds env.symbolic WITHOUT the config's prior_prop (quadrotor.py:326 / cartpole.py:236 call
                # `_setup_symbolic()` bare); controllers rebuild it through BaseController.get_prior -> env._setup_symbolic(prior_prop=...)
                at_construction = AnalyticModel(task, EnvSpec(task, dict(cfg)), {})
                np.testing.assert_allclose(at_construction.U_EQ, np.asarray(env.symbolic.U_EQ, dtype=float).reshape(-1), rtol=1e-13)
                x0, u0 = S.samples(name, env, np.random.default_rng(1))
                np.testing.assert_allclose(at_construction.f(x0[0], u0[0]), np.asarray(env.symbolic.fc_func(x0[0], u0[0])).reshape(-1), rtol=1e-12, atol=1e-12)
                if cfg.get('prior_prop'):
                    env._setup_symbolic(prior_prop=cfg['prior_prop'])
                sym = env.symbolic
                am = AnalyticModel(task, EnvSpec(task, dict(cfg)), cfg.get('prior_prop') or {})
                assert (am.nx, am.nu) == (sym.nx, sym.nu) and abs(am.dt - sym.dt) < 1e-15, 'dims / dt'
                np.testing.assert_allclose(am.X_EQ, np.asarray(env.X_EQ if hasattr(env, 'X_EQ') else sym.X_EQ, dtype=float).reshape(-1), atol=1e-15)
                np.testing.assert_allclose(am.U_EQ, np.asarray(sym.U_EQ, dtype=float).reshape(-1), rtol=1e-13)
                x, u = S.samples(name, env, rng)
                rk = S.rk_discrete(sym.fc_func, sym.nx, sym.nu, sym.dt)
                for i in range(6):
                    np.testing.assert_allclose(am.f(x[i], u[i]), np.asarray(sym.fc_func(x[i], u[i])).reshape(-1), rtol=1e-12, atol=1e-12)
                    A, B = am.df_func(x[i], u[i])
                    Ar, Br = sym.df_func(x[i], u[i])
                    np.testing.assert_allclose(A.toarray(), np.asarray(Ar), rtol=2e-6, atol=2e-6)
                    np.testing.assert_allclose(B.toarray(), np.asarray(Br), rtol=2e-6, atol=5e-5)
                    np.testing.assert_allclose(am.fd_func(x[i], u[i], substeps=1)['xf'].reshape(-1), np.asarray(rk(x[i], u[i])).reshape(-1),
                                               rtol=1e-12, atol=1e-12)
                print('ok  ', tag)
            except Exception as e:                              # noqa: BLE001
                bad.append((tag, repr(e)[:400]))
                print('BAD ', tag, repr(e)[:300])
    print('\n==== bad:', len(bad))
    for b in bad:
        print(b)


if __name__ == '__main__':
    main()

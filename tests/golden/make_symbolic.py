#!/usr/bin/env python3
"""Golden vectors of the reference's PRIOR MODEL, produced by evaluating the reference's own CasADi expressions.

    python tests/golden/make_symbolic.py            (build container only: needs /root/reference)

CasADi itself is not installable here; tests/golden/casadi_numeric.py implements the API slice the reference uses as an
expression graph over NumPy, so what is evaluated is the graph built by the reference's code —
    envs/gym_control/cartpole.py:390-437, envs/gym_pybullet_drones/quadrotor.py:468-604 (`_setup_symbolic`),
    math_and_models/symbolic_systems.py:68-121 (fc_func, df_func = cs.jacobian, loss + its gradients / Hessians, fd_func),
    controllers/mpc/mpc_utils.py:42-64 (rk_discrete: the classical RK4 step the `integrator: rk4` kernels mirror)
— not a restatement of those equations.  Output: tests/golden/symbolic.npz, per system (cartpole, quadrotor 1D/2D/3D)
random (x, u) samples with f, df/dx, df/du, one rk_discrete step of the control period, fd_func (the ODE integrated to
1e-12 by DOP853 in place of CVODES), the loss outputs for random (Xr, Ur) and the linearisation at (X_EQ, U_EQ).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden import ref_stubs  # noqa: E402

ref_stubs.install()

from safe_control_gym.controllers.mpc.mpc_utils import rk_discrete  # noqa: E402
from safe_control_gym.envs.gym_control.cartpole import CartPole  # noqa: E402
from safe_control_gym.envs.gym_pybullet_drones.quadrotor import Quadrotor  # noqa: E402

from tests.golden.make_golden import load_task_config  # noqa: E402

CASES = {
    'cartpole': ('cartpole', 'examples/rl/config_overrides/cartpole/cartpole_stab.yaml', {}),
    'quadrotor_1D': ('quadrotor', 'examples/rl/config_overrides/quadrotor_2D/quadrotor_2D_track.yaml',
                     {'quad_type': 1, 'init_state': {'init_x': 1.0, 'init_x_dot': 0.0}, 'rew_state_weight': [1, 0.1],
                      'rew_act_weight': 0.01, 'constraints': None,
                      'task_info': {'trajectory_type': 'circle', 'num_cycles': 1, 'trajectory_plane': 'zx',
                                    'trajectory_position_offset': [1.0, 0], 'trajectory_scale': 0.5}}),
    'quadrotor_2D': ('quadrotor', 'examples/rl/config_overrides/quadrotor_2D/quadrotor_2D_track.yaml', {}),
    'quadrotor_3D': ('quadrotor', 'examples/rl/config_overrides/quadrotor_3D/quadrotor_3D_track.yaml', {}),
}
N = 24


def samples(name, env, rng):
    nx, nu = env.state_dim, env.action_dim
    if name == 'cartpole':
        x = rng.uniform([-2, -2, -1.2, -3], [2, 2, 1.2, 3], size=(N, nx))
        u = rng.uniform(-10, 10, size=(N, nu))
    else:
        hover = env.GRAVITY_ACC * env.MASS / nu
        u = hover * rng.uniform(0.4, 1.8, size=(N, nu))
        if nx == 2:
            x = rng.uniform([-1, -2], [2, 2], size=(N, nx))
        elif nx == 6:
            x = rng.uniform([-2, -2, 0, -2, -1.0, -4], [2, 2, 2, 2, 1.0, 4], size=(N, nx))
        else:
            lo = [-2, -2, -2, -2, 0, -2, -0.8, -0.8, -1.5, -4, -4, -4]
            x = rng.uniform(lo, [2, 2, 2, 2, 2, 2, 0.8, 0.8, 1.5, 4, 4, 4], size=(N, nx))
    return x, u


def main():
    out, meta = {}, {}
    rng = np.random.default_rng(20260924)
    for name, (task, override, extra) in CASES.items():
        cfg = load_task_config(task, override, extra)
        cfg.pop('seed', None)
        cfg['output_dir'] = '/tmp'
        cfg['cost'] = 'quadratic'                  # makes the env carry Q / R (the model itself does not depend on it)
        env = {'cartpole': CartPole, 'quadrotor': Quadrotor}[task](**cfg)
        sym = env.symbolic
        nx, nu, dt = sym.nx, sym.nu, sym.dt
        x, u = samples(name, env, rng)
        rk = rk_discrete(sym.fc_func, nx, nu, dt)
        f = np.stack([np.asarray(sym.fc_func(x[i], u[i])).reshape(-1) for i in range(N)])
        A = np.stack([np.asarray(sym.df_func(x[i], u[i])[0]) for i in range(N)])
        B = np.stack([np.asarray(sym.df_func(x[i], u[i])[1]) for i in range(N)])
        x_rk4 = np.stack([np.asarray(rk(x[i], u[i])).reshape(-1) for i in range(N)])
        x_fd = np.stack([np.asarray(sym.fd_func(x0=x[i], p=u[i])['xf']).reshape(-1) for i in range(N)])
        Xr = rng.uniform(-1, 1, size=(N, nx))
        Ur = rng.uniform(0, 0.2, size=(N, nu))
        Q, R = np.asarray(env.Q, dtype=float), np.asarray(env.R, dtype=float)
        keys = ('l', 'l_x', 'l_xx', 'l_u', 'l_uu', 'l_xu')
        loss = {k: [] for k in keys}
        for i in range(N):
            res = sym.loss(x=x[i], u=u[i], Xr=Xr[i], Ur=Ur[i], Q=Q, R=R)
            for k in keys:
                loss[k].append(np.asarray(res[k]))
        A_eq, B_eq = (np.asarray(m) for m in sym.df_func(sym.X_EQ, sym.U_EQ))
        pre = name + '/'
        out.update({pre + 'x': x, pre + 'u': u, pre + 'f': f, pre + 'dfdx': A, pre + 'dfdu': B, pre + 'x_rk4': x_rk4,
                    pre + 'x_fd': x_fd, pre + 'Xr': Xr, pre + 'Ur': Ur, pre + 'Q': Q, pre + 'R': R,
                    pre + 'A_eq': A_eq, pre + 'B_eq': B_eq, pre + 'X_EQ': np.asarray(sym.X_EQ, dtype=float),
                    pre + 'U_EQ': np.asarray(sym.U_EQ, dtype=float)})
        for k in keys:
            out[pre + k] = np.stack(loss[k])
        cfg_json = {k: v for k, v in cfg.items() if k != 'output_dir'}
        meta[name] = {'task': task, 'config': json.loads(json.dumps(cfg_json, default=lambda o: np.asarray(o).tolist())),
                      'dt': dt, 'nx': nx, 'nu': nu}
        print(f'{name:14s} nx={nx} nu={nu} dt={dt:.4f}  max|rk4 - ode| = {np.max(np.abs(x_rk4 - x_fd)):.3e}')
        env.close()
    out['meta_json'] = json.dumps(meta)
    np.savez_compressed(os.path.join(HERE, 'symbolic.npz'), **out)
    print('symbolic.npz written')


if __name__ == '__main__':
    main()

"""Prior-model ("symbolic") dynamics of the reference, batched NumPy float64 — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference builds these right-hand sides as CasADi expressions and integrates them with `rk_discrete`
(/root/reference/safe_control_gym/controllers/mpc/mpc_utils.py:42-64, classical RK4) or cvodes:
  envs/gym_control/cartpole.py:400-414             cartpole  (x, x_dot, theta, theta_dot; u = force)
  envs/gym_pybullet_drones/quadrotor.py:485-491    1-D quadrotor (z, z_dot; u = T)
  envs/gym_pybullet_drones/quadrotor.py:497-510    2-D quadrotor (x, x_dot, z, z_dot, theta, theta_dot; u = T1, T2)
  envs/gym_pybullet_drones/quadrotor.py:520-563    3-D quadrotor (pos/vel interleaved, rpy, body rates; u = T1..T4)
This module backs the `integrator: rk4` mode of the HIP kernels (SCG_INT_RK4), an extension key: the reference's envs
always integrate with PyBullet.  Parity of this mode is pinned to the published equations only (casadi is not importable
in the build container), hence "parity unpinned" for the RK4 mode.
"""
import numpy as np


def f_cartpole(x, u, length, cart_mass, pole_mass, g):
    xd, th, thd = x[:, 1], x[:, 2], x[:, 3]
    Mm, ml = pole_mass + cart_mass, pole_mass * length
    sn, cs = np.sin(th), np.cos(th)
    tmp = (u[:, 0] + ml * thd * thd * sn) / Mm
    thdd = (g * sn - cs * tmp) / (length * (4.0 / 3.0 - pole_mass * cs * cs / Mm))
    return np.stack([xd, tmp - ml * thdd * cs / Mm, thd, thdd], axis=1)


def f_quad1d(x, u, mass, g):
    return np.stack([x[:, 1], u[:, 0] / mass - g], axis=1)


def f_quad2d(x, u, mass, iyy, arm, g):
    """arm = L / sqrt(2) (quadrotor.py:509)."""
    th = x[:, 4]
    T = u[:, 0] + u[:, 1]
    return np.stack([x[:, 1], np.sin(th) * T / mass, x[:, 3], np.cos(th) * T / mass - g, x[:, 5],
                     arm * (u[:, 1] - u[:, 0]) / iyy], axis=1)


def f_quad3d(x, u, mass, J, arm, gamma, g):
    """J [N, 3] diagonal inertia; arm = L / sqrt(2); gamma = KM / KF (quadrotor.py:552-562)."""
    phi, th, psi = x[:, 6], x[:, 7], x[:, 8]
    pb, qb, rb = x[:, 9], x[:, 10], x[:, 11]
    sphi, cphi, sth, cth, spsi, cpsi = np.sin(phi), np.cos(phi), np.sin(th), np.cos(th), np.sin(psi), np.cos(psi)
    thrust = u.sum(axis=1) / mass
    r02 = cpsi * sth * cphi + spsi * sphi            # third column of Rz Ry Rx
    r12 = spsi * sth * cphi - cpsi * sphi
    r22 = cth * cphi
    mb0 = arm * (u[:, 0] + u[:, 1] - u[:, 2] - u[:, 3])
    mb1 = arm * (-u[:, 0] + u[:, 1] + u[:, 2] - u[:, 3])
    mb2 = gamma * (-u[:, 0] + u[:, 1] - u[:, 2] + u[:, 3])
    jw0, jw1, jw2 = J[:, 0] * pb, J[:, 1] * qb, J[:, 2] * rb
    tth = sth / cth
    return np.stack([x[:, 1], r02 * thrust, x[:, 3], r12 * thrust, x[:, 5], r22 * thrust - g,
                     pb + sphi * tth * qb + cphi * tth * rb, cphi * qb - sphi * rb, (sphi * qb + cphi * rb) / cth,
                     (mb0 - (qb * jw2 - rb * jw1)) / J[:, 0], (mb1 - (rb * jw0 - pb * jw2)) / J[:, 1],
                     (mb2 - (pb * jw1 - qb * jw0)) / J[:, 2]], axis=1)


def rk4(f, x, u, h, n_steps):
    """n_steps classical RK4 steps of size h with zero-order-hold input (mpc_utils.py:42-64)."""
    for _ in range(int(n_steps)):
        k1 = f(x, u)
        k2 = f(x + 0.5 * h * k1, u)
        k3 = f(x + 0.5 * h * k2, u)
        k4 = f(x + h * k3, u)
        x = x + h / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
    return x

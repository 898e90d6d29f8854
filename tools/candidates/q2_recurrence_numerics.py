#!/usr/bin/env python3
"""NumPy float32 emulation of the planar quadrotor's substep loop as shipped (Taylor rotation per substep) and in the recurrence form
of r04_q2_recurrence_integrator.patch, against float64 with library sin / cos: max |error| of (th, w, vx, vz, x, z) after ONE control
step (20 substeps at 1 kHz) over 100 000 random states, thrusts and torques.  Last run:
    current ['4.48e-07', '4.90e-06', '4.79e-07', '4.82e-07', '4.43e-07', '1.09e-06']
    recur   ['7.90e-08', '1.61e-06', '5.20e-07', '9.70e-07', '4.43e-07', '1.09e-06']
(one-step tolerance of the GPU parity tests: 2e-5 absolute)."""
import numpy as np

f = np.float32
rng = np.random.default_rng(0)
n, nsub, h, g = 100000, 20, f(1 / 1000), 9.8
th0, w0 = rng.uniform(-0.5, 0.5, n), rng.uniform(-3, 3, n)
vx0, vz0, x0, z0 = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.uniform(0, 2, n)
tm, dw_rate = rng.uniform(5, 15, n), rng.uniform(-400, 400, n)            # thrust / mass, torque / inertia


def truth():
    th, w, vx, vz, x, z = (a.astype(np.float64) for a in (th0, w0, vx0, vz0, x0, z0))
    H = 1 / 1000
    for _ in range(nsub):
        w = w + H * dw_rate
        vx, vz = vx + H * (np.sin(th) * tm), vz + H * (np.cos(th) * tm - g)
        x, z, th = x + H * vx, z + H * vz, th + H * w
    return th, w, vx, vz, x, z


def ssc(d):
    d2 = d * d
    return ((d * (f(1) + d2 * (f(-1 / 6) + d2 * f(1 / 120)))).astype(f),
            (f(1) + d2 * (f(-0.5) + d2 * (f(1 / 24) + d2 * f(-1 / 720)))).astype(f))


def start():
    th, w, vx, vz, x, z = (a.astype(f) for a in (th0, w0, vx0, vz0, x0, z0))
    return th, w, vx, vz, x, z, np.sin(th).astype(f), np.cos(th).astype(f), tm.astype(f), (h * dw_rate.astype(f)).astype(f), f(-g)


def current():
    th, w, vx, vz, x, z, sn, cs, T, dwk, gz = start()
    for _ in range(nsub):
        w = (w + dwk).astype(f)
        d = (h * w).astype(f)
        sd, cd = ssc(d)
        vx, vz = (vx + h * (sn * T)).astype(f), (vz + h * (cs * T + gz)).astype(f)
        x, z, th = (x + h * vx).astype(f), (z + h * vz).astype(f), (th + d).astype(f)
        sn, cs = (sn * cd + cs * sd).astype(f), (cs * cd - sn * sd).astype(f)
    return th, w, vx, vz, x, z


def recur():
    th, w, vx, vz, x, z, sn, cs, T, dwk, gz = start()
    se, ce = ssc((h * dwk).astype(f))
    s1, c1 = ssc((h * (w + dwk)).astype(f))
    for _ in range(nsub):
        vx, vz = (vx + h * (sn * T)).astype(f), (vz + h * (cs * T + gz)).astype(f)
        x, z = (x + h * vx).astype(f), (z + h * vz).astype(f)
        sn, cs = (sn * c1 + cs * s1).astype(f), (cs * c1 - sn * s1).astype(f)
        s1, c1 = (s1 * ce + c1 * se).astype(f), (c1 * ce - s1 * se).astype(f)
    th = (f(nsub) * h * w + th + (h * dwk) * f(0.5 * nsub * (nsub + 1))).astype(f)
    w = (f(nsub) * dwk + w).astype(f)
    return th, w, vx, vz, x, z


if __name__ == '__main__':
    t = truth()
    for name, fn in (('current', current), ('recur', recur)):
        print(name, ['%.2e' % np.abs(a.astype(np.float64) - b).max() for a, b in zip(fn(), t)])

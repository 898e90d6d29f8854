"""Reference-trajectory tables (X_GOAL) built on the host once per environment batch.

Same contract as the reference's BenchmarkEnv._generate_trajectory
(/root/reference/safe_control_gym/envs/benchmark_env.py:504-713) and
math_and_models/transformations.py:54-125 (orthogonal projection onto a plane), expressed
as closed-form vector operations over the whole time grid.
"""
import numpy as np

AXIS = {'x': 0, 'y': 1, 'z': 2}


def _plane(plane):
    if len(plane) != 2 or plane[0] not in AXIS or plane[1] not in AXIS or plane[0] == plane[1]:
        raise ValueError('Trajectory plane should be in form of ab, where a and b can be {x, y, z}.')
    return AXIS[plane[0]], AXIS[plane[1]]


def _figure8(t, period, s):
    w = 2.0 * np.pi / period
    sn, cs = np.sin(w * t), np.cos(w * t)
    return s * sn, s * sn * cs, s * w * cs, s * w * (cs ** 2 - sn ** 2)


def _circle(t, period, s):
    w = 2.0 * np.pi / period
    sn, cs = np.sin(w * t), np.cos(w * t)
    return s * cs, s * sn, -s * w * sn, s * w * cs


def _square(t, period, s):
    seg = period / 4.0
    v = s / seg
    cyc = t % period
    along = v * (cyc % seg)
    k = np.floor(cyc / seg).astype(int)
    a = np.select([k == 0, k == 1, k == 2, k == 3], [0.0 * along, -along, -s + 0.0 * along, -s + along], 0.0)
    b = np.select([k == 0, k == 1, k == 2, k == 3], [along, s + 0.0 * along, s - along, 0.0 * along], 0.0)
    ad = np.select([k == 1, k == 3], [-v + 0.0 * along, v + 0.0 * along], 0.0)
    bd = np.select([k == 0, k == 2], [v + 0.0 * along, -v + 0.0 * along], 0.0)
    return a, b, ad, bd


SHAPES = {'figure8': _figure8, 'circle': _circle, 'square': _square}


def planar_reference(traj_type, traj_length, num_cycles, traj_plane, position_offset, scaling, sample_time):
    """(pos[T,3], vel[T,3]) sampled at t = 0, dt, ..., traj_length (one extra sample, benchmark_env.py:543)."""
    if traj_type not in SHAPES:
        raise ValueError('Trajectory type should be one of [circle, square, figure8].')
    ia, ib = _plane(traj_plane)
    t = np.arange(0, traj_length + sample_time, sample_time)
    a, b, ad, bd = SHAPES[traj_type](t, traj_length / num_cycles, scaling)
    pos = np.zeros((t.size, 3))
    vel = np.zeros((t.size, 3))
    pos[:, ia], pos[:, ib] = a + position_offset[0], b + position_offset[1]
    vel[:, ia], vel[:, ib] = ad, bd
    return pos, vel


def project_on_plane(pos, vel, point, normal):
    """Orthogonal projection onto the plane (point, normal) in homogeneous coordinates.

    The reference applies the SAME affine map to velocities (augmented with a 1), so the plane
    offset is added to them too (transformations.py:122-124); reproduced because it defines X_GOAL."""
    n = np.asarray(normal[:3], dtype=np.float64)
    n = n / np.sqrt(n @ n)
    A = np.identity(3) - np.outer(n, n)
    shift = (np.asarray(point[:3], dtype=np.float64) @ n) * n
    return pos @ A.T + shift, vel @ A.T + shift

"""Reference trajectories (X_GOAL tables).  ORACLE — test infrastructure only.

Follows benchmark_env.py:504-713 (_generate_trajectory / _get_coordinates / _figure8 /
_circle / _square) and math_and_models/transformations.py:54-125 (projection_matrix,
orthogonal branch only, and transform_trajectory).
"""
import numpy as np

_AXES = ('x', 'y', 'z')


def _plane_indices(traj_plane):
    # benchmark_env.py:534-541
    if (len(traj_plane) == 2 and traj_plane[0] in _AXES and traj_plane[1] in _AXES
            and traj_plane[0] != traj_plane[1]):
        return _AXES.index(traj_plane[0]), _AXES.index(traj_plane[1])
    raise ValueError('Trajectory plane should be in form of ab, where a and b can be {x, y, z}.')


def planar_coordinates(traj_type, t, period, scaling):
    """(a, b, a_dot, b_dot) at times ``t`` (vector) — benchmark_env.py:607-713."""
    t = np.asarray(t, dtype=np.float64)
    if traj_type == 'figure8':          # :626-631
        w = 2.0 * np.pi / period
        a = scaling * np.sin(w * t)
        b = scaling * np.sin(w * t) * np.cos(w * t)
        ad = scaling * w * np.cos(w * t)
        bd = scaling * w * (np.cos(w * t) ** 2 - np.sin(w * t) ** 2)
        return a, b, ad, bd
    if traj_type == 'circle':           # :652-657
        w = 2.0 * np.pi / period
        a = scaling * np.cos(w * t)
        b = scaling * np.sin(w * t)
        ad = -scaling * w * np.sin(w * t)
        bd = scaling * w * np.cos(w * t)
        return a, b, ad, bd
    if traj_type == 'square':           # :678-713
        seg_period = period / 4.0
        speed = scaling / seg_period
        cyc = t % period
        seg_t = cyc % seg_period
        seg_i = np.floor(cyc / seg_period).astype(int)
        seg_p = speed * seg_t
        a = np.zeros_like(t)
        b = np.zeros_like(t)
        ad = np.zeros_like(t)
        bd = np.zeros_like(t)
        m = seg_i == 0
        b[m], bd[m] = seg_p[m], speed
        m = seg_i == 1
        a[m], b[m], ad[m] = -seg_p[m], scaling, -speed
        m = seg_i == 2
        a[m], b[m], bd[m] = -scaling, scaling - seg_p[m], -speed
        m = seg_i == 3
        a[m], ad[m] = -scaling + seg_p[m], speed
        # (seg_i == 4 can only arise from round-off at cyc == period; the reference would
        #  raise UnboundLocalError there, we leave zeros.)
        return a, b, ad, bd
    raise ValueError('Trajectory type should be one of [circle, square, figure8].')


def generate_trajectory(traj_type='figure8', traj_length=10.0, num_cycles=1, traj_plane='xy',
                        position_offset=(0.0, 0.0), scaling=1.0, sample_time=0.01):
    """benchmark_env.py:504-558.  Returns (pos_ref (T,3), vel_ref (T,3), speed (T,1))."""
    period = traj_length / num_cycles
    ia, ib = _plane_indices(traj_plane)
    times = np.arange(0, traj_length + sample_time, sample_time)   # :543 (one step longer)
    a, b, ad, bd = planar_coordinates(traj_type, times, period, scaling)
    pos = np.zeros((len(times), 3))
    vel = np.zeros((len(times), 3))
    pos[:, ia] = a + position_offset[0]
    vel[:, ia] = ad
    pos[:, ib] = b + position_offset[1]
    vel[:, ib] = bd
    speed = np.linalg.norm(vel, axis=1, keepdims=True)
    return pos, vel, speed


def orthogonal_projection_matrix(point, normal):
    """transformations.py:86-107, orthogonal branch (no direction / perspective)."""
    M = np.identity(4)
    point = np.asarray(point[:3], dtype=np.float64)
    normal = np.asarray(normal[:3], dtype=np.float64)
    normal = normal / np.sqrt(np.dot(normal, normal))
    M[:3, :3] -= np.outer(normal, normal)
    M[:3, 3] = np.dot(point, normal) * normal
    return M


def transform_trajectory(pos, vel, point, normal):
    """transformations.py:110-125.  NOTE (reference quirk, replicated): velocities are
    augmented with a 1 as well, so the plane offset is added to them (:122-124)."""
    M = orthogonal_projection_matrix(point, normal)
    aug_pos = np.concatenate([pos, np.ones((pos.shape[0], 1))], -1)
    aug_vel = np.concatenate([vel, np.ones((vel.shape[0], 1))], -1)
    return (aug_pos @ M.T)[:, :3], (aug_vel @ M.T)[:, :3]

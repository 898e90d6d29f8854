"""Restated Bullet / PyBullet semantics used by the reference's two robots.

ORACLE — test infrastructure only (see oracle/__init__.py).

PyBullet (``pybullet ^3.2``, /root/reference/pyproject.toml:24) is a third-party
dependency that is NOT vendored under /root/reference and not installable here, so
this file restates the published algorithm of the calls the reference makes
(parity unpinned against a real Bullet build; everything above this layer is pinned
by tests/golden/).  Reference call sites:

* quadrotor: base_aviary.py:212-226 (load, zero damping), :364-384 (_physics: 4 prop
  forces in LINK_FRAME + yaw torque on the COM link), :271-279 (world-frame
  disturbance force at the COM), :282 (stepSimulation), :327-331 (state read-back);
  quadrotor.py:361-366 (changeDynamics mass / inertia), :380-384 (state reset), :795.
* cartpole: cartpole.py:301-322 (load without URDF_USE_INERTIA_FROM_FILE, zero
  damping, motors disabled, mass override), :331-346, :557-583 (tab force at the
  pole COM in WORLD_FRAME, TORQUE_CONTROL on the slider joint, stepSimulation).

Bullet algorithm restated (btMultiBodyDynamicsWorld::internalSingleStepSimulation,
double precision build used by the pybullet wheel):

1. forward dynamics at the CURRENT (q, qdot) with Featherstone's ABA
   (btMultiBody::computeAccelerationsArticulatedBodyAlgorithmMultiDof), gravity
   added as m*g per link, explicit gyroscopic term  w_b x (J w_b)  for the floating
   base (m_useGyroTerm = true), linear/angular damping zeroed by the reference;
2. qdot += h * qddot  (applyDeltaVeeMultiDof), every velocity coordinate clamped to
   +-m_maxCoordinateVelocity = 100;
3. q += h * qdot_new  (stepPositionsMultiDof): semi-implicit (symplectic) Euler;
   the floating-base orientation is advanced with the exponential map
   q <- normalize(exp(h/2 * w_world) (x) q), with Bullet's small-angle Taylor
   branch (|w| < 1e-3) and its ANGULAR_MOTION_THRESHOLD = pi/4 per step cap;
4. external forces / torques are cleared after every step.

All functions are batched: leading axis = environment.
"""
import numpy as np

MAX_COORDINATE_VELOCITY = 100.0          # btMultiBody::m_maxCoordinateVelocity
ANGULAR_MOTION_THRESHOLD = 0.25 * np.pi  # 0.5 * SIMD_HALF_PI


# --------------------------------------------------------------------------- #
# Quaternion helpers, PyBullet conventions: q = (x, y, z, w), body -> world.
# --------------------------------------------------------------------------- #
def quaternion_from_euler(rpy):
    """p.getQuaternionFromEuler (btQuaternion::setEulerZYX): q = qz(yaw) qy(pitch) qx(roll).

    Used at quadrotor.py:381 and base_aviary.py:223."""
    rpy = np.asarray(rpy, dtype=np.float64)
    hr, hp, hy = 0.5 * rpy[..., 0], 0.5 * rpy[..., 1], 0.5 * rpy[..., 2]
    cr, sr = np.cos(hr), np.sin(hr)
    cp, sp = np.cos(hp), np.sin(hp)
    cy, sy = np.cos(hy), np.sin(hy)
    x = sr * cp * cy - cr * sp * sy
    y = cr * sp * cy + sr * cp * sy
    z = cr * cp * sy - sr * sp * cy
    w = cr * cp * cy + sr * sp * sy
    return np.stack([x, y, z, w], axis=-1)


def euler_from_quaternion(q):
    """p.getEulerFromQuaternion (pybullet.c), incl. its gimbal-lock branches.

    Used at base_aviary.py:329."""
    q = np.asarray(q, dtype=np.float64)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    sqx, sqy, sqz, squ = x * x, y * y, z * z, w * w
    sarg = -2.0 * (x * z - w * y)
    roll = np.arctan2(2.0 * (y * z + w * x), squ - sqx - sqy + sqz)
    pitch = np.arcsin(np.clip(sarg, -1.0, 1.0))
    yaw = np.arctan2(2.0 * (x * y + w * z), squ + sqx - sqy - sqz)
    lo = sarg <= -0.99999
    hi = sarg >= 0.99999
    roll = np.where(lo | hi, 0.0, roll)
    pitch = np.where(lo, -0.5 * np.pi, np.where(hi, 0.5 * np.pi, pitch))
    yaw = np.where(lo, 2.0 * np.arctan2(x, -y), np.where(hi, 2.0 * np.arctan2(-x, y), yaw))
    return np.stack([roll, pitch, yaw], axis=-1)


def matrix_from_quaternion(q):
    """p.getMatrixFromQuaternion (btMatrix3x3::setRotation); returns (..., 3, 3), body -> world.

    Used at quadrotor.py:795."""
    q = np.asarray(q, dtype=np.float64)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    d = x * x + y * y + z * z + w * w
    s = 2.0 / d
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz = w * xs, w * ys, w * zs
    xx, xy, xz = x * xs, x * ys, x * zs
    yy, yz, zz = y * ys, y * zs, z * zs
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1.0 - (yy + zz)
    R[..., 0, 1] = xy - wz
    R[..., 0, 2] = xz + wy
    R[..., 1, 0] = xy + wz
    R[..., 1, 1] = 1.0 - (xx + zz)
    R[..., 1, 2] = yz - wx
    R[..., 2, 0] = xz - wy
    R[..., 2, 1] = yz + wx
    R[..., 2, 2] = 1.0 - (xx + yy)
    return R


def _quat_mul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
        aw * bw - ax * bx - ay * by - az * bz,
    ], axis=-1)


def integrate_base_orientation(quat, omega_world, h):
    """btMultiBody::stepPositionsMultiDof's pQuatUpdateFun for the base body.

    Bullet stores the base quaternion as world->base and right-multiplies it by
    Q(-axis, cos); for the body->world quaternion exposed by pybullet this is
    q <- normalize(Q(axis, cos(|w| h / 2)) (x) q) with axis = w * sin(|w| h/2)/|w|."""
    w = np.asarray(omega_world, dtype=np.float64)
    ang = np.sqrt(np.sum(w * w, axis=-1))
    ang = np.where(ang * h > ANGULAR_MOTION_THRESHOLD, 0.5 * (0.5 * np.pi) / h, ang)
    small = ang < 0.001
    safe = np.where(small, 1.0, ang)
    k_small = 0.5 * h - (h * h * h) * 0.020833333333 * ang * ang
    k_big = np.sin(0.5 * ang * h) / safe
    k = np.where(small, k_small, k_big)
    dq = np.concatenate([w * k[..., None], np.cos(0.5 * ang * h)[..., None]], axis=-1)
    qn = _quat_mul(dq, quat)
    return qn / np.sqrt(np.sum(qn * qn, axis=-1, keepdims=True))


# --------------------------------------------------------------------------- #
# Free-floating rigid body with 4 thrust links (cf2x.urdf) — one Bullet step.
# --------------------------------------------------------------------------- #
def quadrotor_substep(pos, quat, vel, omega, prop_forces, yaw_torque, dist_force,
                      mass, inertia_diag, arm, gravity, h, dist_point=None):
    """One ``p.stepSimulation`` for the quadrotor multibody (base_aviary.py:364-384,:271-282).

    Args (batched over axis 0):
        pos (N,3), quat (N,4 xyzw body->world), vel (N,3 world), omega (N,3 world).
        prop_forces (N,4): KF*rpm_i^2 applied along link z at the prop link origins
            (+d,+d,0), (-d,+d,0), (-d,-d,0), (+d,-d,0), d = ``arm`` (0.028, cf2x.urdf:42-78).
        yaw_torque (N,): KM*(-r0^2 + r1^2 - r2^2 + r3^2) about body z.
        dist_force (N,3) or None: world-frame force on the COM link.
        dist_point (N,3) or None: world position the force is applied at.  pybullet's
            applyExternalForce(WORLD_FRAME) adds the torque (posObj - link_origin) x F; the
            reference passes the position CACHED AT THE START OF THE CONTROL STEP
            (base_aviary.py:272 reads self.pos, refreshed only at :286), so from the second
            substep on the force acts off-centre and produces a small torque.  None = at the COM.
        mass (N,), inertia_diag (N,3).
    Returns new (pos, quat, vel, omega).
    """
    R = matrix_from_quaternion(quat)
    f0, f1, f2, f3 = (prop_forces[:, i] for i in range(4))
    thrust = f0 + f1 + f2 + f3
    tau_b = np.stack([arm * (f0 + f1 - f2 - f3),
                      arm * (-f0 + f1 + f2 - f3),
                      yaw_torque], axis=-1)
    if dist_force is not None and dist_point is not None:
        tau_w = np.cross(dist_point - pos, dist_force)
        tau_b = tau_b + np.einsum('nji,nj->ni', R, tau_w)
    w_b = np.einsum('nji,nj->ni', R, omega)          # R^T w
    Jw = inertia_diag * w_b
    wdot_b = (tau_b - np.cross(w_b, Jw)) / inertia_diag
    wdot = np.einsum('nij,nj->ni', R, wdot_b)
    acc = R[:, :, 2] * (thrust / mass)[:, None]
    acc[:, 2] -= gravity
    if dist_force is not None:
        acc = acc + dist_force / mass[:, None]
    omega = np.clip(omega + h * wdot, -MAX_COORDINATE_VELOCITY, MAX_COORDINATE_VELOCITY)
    vel = np.clip(vel + h * acc, -MAX_COORDINATE_VELOCITY, MAX_COORDINATE_VELOCITY)
    pos = pos + h * vel
    quat = integrate_base_orientation(quat, omega, h)
    return pos, quat, vel, omega


# --------------------------------------------------------------------------- #
# Cart-pole multibody (cartpole_template.urdf) — one Bullet step.
# --------------------------------------------------------------------------- #
POLE_BOX_WIDTH = 0.05   # cartpole_template.urdf:63 (collision box 0.05 x 0.05 x 2l)


def pole_inertia(pole_mass, pole_half_length, mode='box'):
    """Pole inertia about its COM around the hinge axis (y).

    'box': what Bullet ends up with.  The reference loads the rewritten URDF WITHOUT
    URDF_USE_INERTIA_FROM_FILE (cartpole.py:301-304) and then calls
    changeDynamics(mass=...) on the pole (cartpole.py:318-322); both code paths make
    Bullet recompute the local inertia from the link's collision shape
    (btCompoundShape/btBoxShape::calculateLocalInertia -> m/12 (lx^2 + lz^2) of the
    0.05 x 0.05 x 2l box).
    'rod': the slender-rod value the reference writes into the URDF
    (cartpole.py:296, m (2l)^2 / 12), which Bullet ignores in this configuration; it
    equals the CasADi prior model (cartpole.py:412-414).
    """
    if mode == 'box':
        return pole_mass * (POLE_BOX_WIDTH ** 2 + (2.0 * pole_half_length) ** 2) / 12.0
    if mode == 'rod':
        return pole_mass * (2.0 * pole_half_length) ** 2 / 12.0
    raise ValueError(mode)


def cartpole_substep(x, x_dot, theta, theta_dot, force, tab_force,
                     cart_mass, pole_mass, pole_half_length, pole_inertia_com, gravity, h):
    """One ``p.stepSimulation`` for the cart-pole (cartpole.py:552-583).

    Fixed base ``slideBar``; prismatic joint along +x (cart, mass M); revolute joint
    about +y at the cart origin; pole COM at height l (cartpole_template.urdf:41-76).
    With generalised coordinates (x, theta) the ABA forward dynamics equal

        [ M+m        m l cos(th) ] [x_dd ]   [ F + m l th_d^2 sin(th) + f_x              ]
        [ m l cos(th) I_p + m l^2] [th_dd] = [ m g l sin(th) + l (f_x cos(th) - f_z sin(th)) ]

    where f = (f_x, f_z) is the world-frame "tab" force applied at the pole COM.
    Then semi-implicit Euler with the +-100 coordinate-velocity clamp.
    """
    s, c = np.sin(theta), np.cos(theta)
    m, M, l = pole_mass, cart_mass, pole_half_length
    a11 = M + m
    a12 = m * l * c
    a22 = pole_inertia_com + m * l * l
    b1 = force + m * l * theta_dot * theta_dot * s
    b2 = m * gravity * l * s
    if tab_force is not None:
        b1 = b1 + tab_force[:, 0]
        b2 = b2 + l * (tab_force[:, 0] * c - tab_force[:, 1] * s)
    det = a11 * a22 - a12 * a12
    x_dd = (a22 * b1 - a12 * b2) / det
    th_dd = (a11 * b2 - a12 * b1) / det
    x_dot = np.clip(x_dot + h * x_dd, -MAX_COORDINATE_VELOCITY, MAX_COORDINATE_VELOCITY)
    theta_dot = np.clip(theta_dot + h * th_dd, -MAX_COORDINATE_VELOCITY, MAX_COORDINATE_VELOCITY)
    x = x + h * x_dot
    theta = theta + h * theta_dot
    return x, x_dot, theta, theta_dot

"""An INDEPENDENT anchor for the restated Bullet step (oracle/bullet.py): a generic, URDF-driven Featherstone articulated-body
algorithm (spatial vectors, three passes; fixed base with prismatic / revolute joints, floating base, fixed joints folded into
the parent) fed by the reference's own robot descriptions —

    /root/reference/safe_control_gym/envs/gym_control/assets/cartpole_template.urdf   (slideBar -> cart -> pole)
    /root/reference/safe_control_gym/envs/gym_pybullet_drones/assets/cf2x.urdf        (base_link + 4 massless prop links + COM link)

— and compared at 1e-12 with the CLOSED FORMS oracle/bullet.py integrates (cartpole_substep's 2x2 mass-matrix solve,
quadrotor_substep's Newton-Euler step).  What this catches and first-order convergence to the symbolic ODE
(tests/test_bullet_convergence.py) cannot: reduction errors — which inertia enters the hinge equation, the COM offset of the
pole, where on the body a LINK_FRAME prop force acts (the prop link's inertial origin, cf2x.urdf:42-78), the sign / frame of the
gyroscopic term and of the yaw torque, the lever arm of a world-frame force applied at the pole's COM.

The algorithm is btMultiBody::computeAccelerationsArticulatedBodyAlgorithmMultiDof's (Featherstone, "Rigid Body Dynamics
Algorithms", 2008, table 7.1 + section 9.4 for the floating base), written from the book, not from the oracle: no shared code.
The URDF numbers are read from the reference's files when a checkout / staged copy is present; otherwise from the minimal
dynamics-only restatement below (asserted equal to the real files whenever those are readable)."""
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from oracle import bullet

G = 9.8

# dynamics-relevant content of the two URDFs (links: inertial origin / mass / inertia / collision box; joints: type, axis, origin)
CARTPOLE_URDF = """<robot name="physics">
  <link name="slideBar"><inertial><mass value="0"/><inertia ixx="1.0" iyy="1.0" izz="1.0"/></inertial></link>
  <link name="cart"><collision><geometry><box size="0.5 0.5 0.2"/></geometry><origin xyz="0 0 0"/></collision>
    <inertial><mass value="1"/><inertia ixx="1.0" iyy="1.0" izz="1.0"/></inertial></link>
  <joint name="slider_to_cart" type="prismatic"><axis xyz="1 0 0"/><origin xyz="0.0 0.0 0.0"/><parent link="slideBar"/><child link="cart"/></joint>
  <link name="pole"><inertial><origin xyz="0 0 0.5"/><mass value="0.1"/><inertia ixx="1.0" iyy="1.0" izz="1.0"/></inertial>
    <collision><geometry><box size="0.05 0.05 1.0"/></geometry><origin rpy="0 0 0" xyz="0 0 0.5"/></collision></link>
  <joint name="cart_to_pole" type="continuous"><axis xyz="0 1 0"/><origin xyz="0.0 0.0 0"/><parent link="cart"/><child link="pole"/></joint>
</robot>"""
CF2X_URDF = """<robot name="cf2">
  <link name="base_link"><inertial><origin rpy="0 0 0" xyz="0 0 0"/><mass value="0.027"/>
    <inertia ixx="1.4e-5" ixy="0.0" ixz="0.0" iyy="1.4e-5" iyz="0.0" izz="2.17e-5"/></inertial></link>
  <link name="prop0_link"><inertial><origin rpy="0 0 0" xyz="0.028 0.028 0"/><mass value="0"/><inertia ixx="0" iyy="0" izz="0"/></inertial></link>
  <joint name="prop0_joint" type="fixed"><parent link="base_link"/><child link="prop0_link"/></joint>
  <link name="prop1_link"><inertial><origin rpy="0 0 0" xyz="-0.028 0.028 0"/><mass value="0"/><inertia ixx="0" iyy="0" izz="0"/></inertial></link>
  <joint name="prop1_joint" type="fixed"><parent link="base_link"/><child link="prop1_link"/></joint>
  <link name="prop2_link"><inertial><origin rpy="0 0 0" xyz="-0.028 -0.028 0"/><mass value="0"/><inertia ixx="0" iyy="0" izz="0"/></inertial></link>
  <joint name="prop2_joint" type="fixed"><parent link="base_link"/><child link="prop2_link"/></joint>
  <link name="prop3_link"><inertial><origin rpy="0 0 0" xyz="0.028 -0.028 0"/><mass value="0"/><inertia ixx="0" iyy="0" izz="0"/></inertial></link>
  <joint name="prop3_joint" type="fixed"><parent link="base_link"/><child link="prop3_link"/></joint>
  <link name="center_of_mass_link"><inertial><origin rpy="0 0 0" xyz="0 0 0"/><mass value="0"/><inertia ixx="0" iyy="0" izz="0"/></inertial></link>
  <joint name="center_of_mass_joint" type="fixed"><parent link="base_link"/><child link="center_of_mass_link"/></joint>
</robot>"""


# ------------------------------------------------------------------------------------------------ spatial algebra (Featherstone ch. 2)
def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def rot_rpy(rpy):
    r, p, y = rpy
    Rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    Ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
    Rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1.0]])
    return Rz @ Ry @ Rx                     # URDF: fixed-axis roll, pitch, yaw


def xform(E, r):
    """Plücker motion transform A -> B for frame B at position r (in A) with B's axes = columns of E^T ... X = [E 0; -E r^x E]."""
    X = np.zeros((6, 6))
    X[:3, :3] = E
    X[3:, 3:] = E
    X[3:, :3] = -E @ skew(r)
    return X


def crm(v):
    """Motion cross product matrix v x ."""
    M = np.zeros((6, 6))
    M[:3, :3] = skew(v[:3])
    M[3:, 3:] = skew(v[:3])
    M[3:, :3] = skew(v[3:])
    return M


def crf(v):
    return -crm(v).T


def spatial_inertia(m, c, Ic):
    """Rigid-body inertia about the link origin: mass m, COM c (link coordinates), rotational inertia Ic about the COM."""
    C = skew(c)
    I = np.zeros((6, 6))
    I[:3, :3] = Ic + m * C @ C.T
    I[:3, 3:] = m * C
    I[3:, :3] = m * C.T
    I[3:, 3:] = m * np.eye(3)
    return I


# ------------------------------------------------------------------------------------------------ URDF -> model
def _vec(s, n=3):
    return np.array([float(x) for x in s.split()]) if s else np.zeros(n)


def parse_urdf(text, inertia_from_collision=False):
    """links: name -> (mass, com, R_inertial, Ic); joints: child -> (parent, type, axis, xyz, rpy).  inertia_from_collision: what
    Bullet does when the URDF is loaded WITHOUT URDF_USE_INERTIA_FROM_FILE / after changeDynamics(mass=): the rotational inertia of
    the link's collision box, m/12 (ly^2+lz^2, lx^2+lz^2, lx^2+ly^2) (btBoxShape::calculateLocalInertia)."""
    root = ET.fromstring(text)
    links, joints = {}, {}
    for ln in root.findall('link'):
        ine = ln.find('inertial')
        m, com, rpy, Ic = 0.0, np.zeros(3), np.zeros(3), np.zeros((3, 3))
        if ine is not None:
            m = float(ine.find('mass').get('value'))
            o = ine.find('origin')
            if o is not None:
                com, rpy = _vec(o.get('xyz')), _vec(o.get('rpy'))
            it = ine.find('inertia')
            g = lambda k: float(it.get(k, 0.0))                                  # noqa: E731
            Ic = np.array([[g('ixx'), g('ixy'), g('ixz')], [g('ixy'), g('iyy'), g('iyz')], [g('ixz'), g('iyz'), g('izz')]])
        col = ln.find('collision')
        if inertia_from_collision and col is not None and col.find('geometry/box') is not None:
            lx, ly, lz = _vec(col.find('geometry/box').get('size'))
            Ic = m / 12.0 * np.diag([ly * ly + lz * lz, lx * lx + lz * lz, lx * lx + ly * ly])
        links[ln.get('name')] = (m, com, rot_rpy(rpy), Ic)
    for j in root.findall('joint'):
        o = j.find('origin')
        xyz, rpy = (_vec(o.get('xyz')), _vec(o.get('rpy'))) if o is not None else (np.zeros(3), np.zeros(3))
        ax = j.find('axis')
        joints[j.find('child').get('link')] = (j.find('parent').get('link'), j.get('type'), _vec(ax.get('xyz')) if ax is not None else np.zeros(3),
                                               xyz, rpy)
    return links, joints


def build_tree(links, joints):
    """Movable bodies in topological order; links behind FIXED joints are folded into their parent body (inertia transformed to the
    parent's frame) and remembered with their placement (so that forces can be applied to them).  Returns (bodies, placement):
    bodies[i] = dict(name, parent index, joint type, axis, XT (parent body frame -> joint frame), I); placement[link] = (body, E, r)."""
    root = next(n for n in links if n not in joints)
    bodies, placement = [], {}

    def add(name, body, E, r):                      # link `name` rides on `body` at rotation E^T / offset r (body coordinates)
        m, com, Rin, Ic = links[name]
        Ib = spatial_inertia(m, r + E.T @ com, E.T @ Rin @ Ic @ Rin.T @ E)
        bodies[body]['I'] = bodies[body]['I'] + Ib
        placement[name] = (body, E, r)
        for ch, (par, typ, axis, xyz, rpy) in joints.items():
            if par != name:
                continue
            Ej = rot_rpy(rpy).T @ E                 # child axes expressed from the body frame
            rj = r + E.T @ xyz
            if typ == 'fixed':
                add(ch, body, Ej, rj)
            else:
                bodies.append({'name': ch, 'parent': body, 'type': typ, 'axis': axis, 'XT': xform(Ej, rj), 'I': np.zeros((6, 6))})
                add(ch, len(bodies) - 1, np.eye(3), np.zeros(3))
    bodies.append({'name': root, 'parent': -1, 'type': 'base', 'axis': np.zeros(3), 'XT': np.eye(6), 'I': np.zeros((6, 6))})
    add(root, 0, np.eye(3), np.zeros(3))
    return bodies, placement


def joint(typ, axis, q):
    """(XJ, S): revolute = rotation by q about `axis`, prismatic = translation by q along it (axis-aligned axes suffice here)."""
    S = np.zeros(6)
    if typ in ('revolute', 'continuous'):
        S[:3] = axis
        K = skew(axis)
        E = (np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * K @ K).T          # Rodrigues, transposed: coordinates parent -> child
        return xform(E, np.zeros(3)), S
    S[3:] = axis
    return xform(np.eye(3), axis * q), S


def aba(bodies, q, qd, tau, f_ext, floating, v_base=None, gravity=np.array([0.0, 0.0, -G]), R_base=np.eye(3)):
    """Featherstone table 7.1 (fixed base; gravity as a fictitious base acceleration) / section 9.4 (floating base: 6-D base
    acceleration from the articulated inertia of the whole tree).  f_ext[i]: spatial force on body i in ITS coordinates.
    Returns (qdd of the joints, base spatial acceleration in base coordinates [floating only])."""
    n = len(bodies)
    Xup, S, v, c, IA, pA = [None] * n, [None] * n, [None] * n, [None] * n, [None] * n, [None] * n
    v[0] = v_base if floating else np.zeros(6)
    c[0] = np.zeros(6)
    IA[0] = bodies[0]['I'].copy()
    pA[0] = crf(v[0]) @ IA[0] @ v[0] - f_ext[0]
    for i in range(1, n):
        b = bodies[i]
        XJ, S[i] = joint(b['type'], b['axis'], q[i - 1])
        Xup[i] = XJ @ b['XT']
        vJ = S[i] * qd[i - 1]
        v[i] = Xup[i] @ v[b['parent']] + vJ
        c[i] = crm(v[i]) @ vJ
        IA[i] = b['I'].copy()
        pA[i] = crf(v[i]) @ IA[i] @ v[i] - f_ext[i]
    U, d, u = [None] * n, [None] * n, [None] * n
    for i in range(n - 1, 0, -1):
        U[i] = IA[i] @ S[i]
        d[i] = S[i] @ U[i]
        u[i] = tau[i - 1] - S[i] @ pA[i]
        Ia = IA[i] - np.outer(U[i], U[i]) / d[i]
        pa = pA[i] + Ia @ c[i] + U[i] * u[i] / d[i]
        p = bodies[i]['parent']
        IA[p] = IA[p] + Xup[i].T @ Ia @ Xup[i]
        pA[p] = pA[p] + Xup[i].T @ pa
    a = [None] * n
    g6 = np.concatenate([np.zeros(3), R_base.T @ gravity])
    if floating:
        a[0] = -np.linalg.solve(IA[0], pA[0]) + g6         # (gravity as a uniform acceleration field: exact for every body)
    else:
        a[0] = -g6
    qdd = np.zeros(n - 1)
    for i in range(1, n):
        ap = Xup[i] @ a[bodies[i]['parent']] + c[i]
        qdd[i - 1] = (u[i] - U[i] @ ap) / d[i]
        a[i] = ap + S[i] * qdd[i - 1]
    return qdd, a[0]


def point_force(F, r):
    """Spatial force of a linear force F acting at point r (both in the body's coordinates)."""
    return np.concatenate([np.cross(r, F), F])


def _reference_urdf(rel, fallback):
    from tests.golden.ref_stubs import reference_root
    ref = reference_root()
    path = os.path.join(ref, rel) if ref else None
    return (open(path).read(), True) if path and os.path.isfile(path) else (fallback, False)


CP_REL = 'safe_control_gym/envs/gym_control/assets/cartpole_template.urdf'
CF_REL = 'safe_control_gym/envs/gym_pybullet_drones/assets/cf2x.urdf'


def _same_model(a, b):
    (la, ja), (lb, jb) = a, b
    assert set(la) == set(lb) and set(ja) == set(jb)
    for k in la:
        for x, y in zip(la[k], lb[k]):
            np.testing.assert_allclose(x, y, rtol=0, atol=0, err_msg=k)
    for k in ja:
        assert ja[k][:2] == jb[k][:2]
        for x, y in zip(ja[k][2:], jb[k][2:]):
            np.testing.assert_allclose(x, y, rtol=0, atol=0, err_msg=k)


def test_embedded_urdf_restatements_equal_the_reference_files():
    found = 0
    for rel, text in ((CP_REL, CARTPOLE_URDF), (CF_REL, CF2X_URDF)):
        real, ok = _reference_urdf(rel, None)
        if ok:
            _same_model(parse_urdf(real, True), parse_urdf(text, True))
            _same_model(parse_urdf(real, False), parse_urdf(text, False))
            found += 1
    if not found:
        pytest.skip('no reference checkout / staged copy: the embedded restatements are what the other tests use')


# ------------------------------------------------------------------------------------------------ cart-pole
@pytest.mark.parametrize('inertia', ['box', 'rod'])
def test_cartpole_closed_form_equals_generic_aba_on_the_reference_urdf(inertia):
    text, _ = _reference_urdf(CP_REL, CARTPOLE_URDF)
    links, joints = parse_urdf(text, inertia_from_collision=(inertia == 'box'))
    m, com, _, Ic = links['pole']
    M = links['cart'][0]
    l = com[2]
    if inertia == 'rod':                    # what cartpole.py:296 writes into the URDF (and the CasADi prior assumes): m (2l)^2 / 12
        links['pole'] = (m, com, np.eye(3), np.diag([m * (2 * l) ** 2 / 12.0] * 2 + [0.0]))
    bodies, place = build_tree(links, joints)
    assert [b['name'] for b in bodies] == ['slideBar', 'cart', 'pole'] and (M, m, l) == (1.0, 0.1, 0.5)
    Ip = bullet.pole_inertia(np.array([m]), np.array([l]), inertia)
    rng = np.random.default_rng(0)
    h = 1.0                                  # one unit step from rest-of-state: (new - old) / h IS the acceleration
    worst = 0.0
    for _ in range(200):
        x, xd, th, thd = rng.uniform([-2, -3, -3.1, -6], [2, 3, 3.1, 6])
        F = rng.uniform(-10, 10)
        tab = rng.uniform(-1, 1, 2) if rng.random() < 0.5 else None
        # generic: tab force (f_x, 0, f_z) in WORLD axes at the pole COM -> pole coordinates (pole frame = world rotated by th about y)
        fext = [np.zeros(6)] * 3
        if tab is not None:
            K = skew(np.array([0, 1.0, 0]))
            Rw = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K           # pole axes in world coordinates
            fext = [np.zeros(6), np.zeros(6), point_force(Rw.T @ np.array([tab[0], 0.0, tab[1]]), com)]
        qdd, _ = aba(bodies, [x, th], [xd, thd], [F, 0.0], fext, floating=False)
        got = bullet.cartpole_substep(np.array([x]), np.array([xd]), np.array([th]), np.array([thd]), np.array([F]),
                                      None if tab is None else tab[None], np.array([M]), np.array([m]), np.array([l]), Ip, G, h)
        ref = np.array([got[1][0] - xd, got[3][0] - thd])
        worst = max(worst, np.abs(qdd - ref).max() / max(1.0, np.abs(ref).max()))
    assert worst < 1e-12, worst


# ------------------------------------------------------------------------------------------------ quadrotor
def test_quadrotor_closed_form_equals_generic_aba_on_the_reference_urdf():
    text, _ = _reference_urdf(CF_REL, CF2X_URDF)
    links, joints = parse_urdf(text)
    bodies, place = build_tree(links, joints)
    assert len(bodies) == 1 and len(place) == 6                                  # one composite floating body: base + five folded links
    mass = links['base_link'][0]
    J = np.diag(links['base_link'][3])
    assert mass == 0.027 and np.allclose(J, [1.4e-5, 1.4e-5, 2.17e-5])
    arm = place['prop0_link'][2][0] + links['prop0_link'][1][0]                  # the prop link's INERTIAL origin: where LINK_FRAME forces act
    assert arm == 0.028
    rng = np.random.default_rng(1)
    worst = 0.0
    for _ in range(200):
        rpy = rng.uniform(-1.2, 1.2, 3)
        quat = bullet.quaternion_from_euler(rpy[None])
        R = bullet.matrix_from_quaternion(quat)[0]
        vel, omega = rng.uniform(-3, 3, 3), rng.uniform(-8, 8, 3)
        f = rng.uniform(0.0, 0.15, 4)
        tz = rng.uniform(-1e-3, 1e-3)
        dist = rng.uniform(-0.05, 0.05, 3) if rng.random() < 0.5 else None
        # generic: each prop force along ITS link's z at its inertial origin, the yaw torque on the COM link, the disturbance at the COM
        fx = np.zeros(6)
        for i in range(4):
            _, E, r = place[f'prop{i}_link']
            fx += point_force(E.T @ np.array([0, 0, f[i]]), r + E.T @ links[f'prop{i}_link'][1])
        fx[:3] += np.array([0, 0, tz])
        if dist is not None:
            fx += point_force(R.T @ dist, np.zeros(3))
        vb = np.concatenate([R.T @ omega, R.T @ vel])
        _, a0 = aba(bodies, [], [], [], [fx], floating=True, v_base=vb, R_base=R)
        wdot = R @ a0[:3]
        acc = R @ (a0[3:] + np.cross(vb[:3], vb[3:]))                            # classical acceleration of the origin (= COM)
        h = 2.0 ** -10                      # (new - old) / h IS the acceleration, to ~1e-13 (a power of two: the division is exact)
        p2, q2, v2, w2 = bullet.quadrotor_substep(np.zeros((1, 3)), quat, vel[None].copy(), omega[None].copy(), f[None], np.array([tz]),
                                                  None if dist is None else dist[None], np.array([mass]), J[None], arm, G, h)
        ref_w, ref_a = (w2[0] - omega) / h, (v2[0] - vel) / h
        assert np.abs(w2).max() < 100 and np.abs(v2).max() < 100                 # (the +-100 clamp is not in play)
        worst = max(worst, np.abs(wdot - ref_w).max() / max(1.0, np.abs(ref_w).max()), np.abs(acc - ref_a).max() / max(1.0, np.abs(ref_a).max()))
    assert worst < 1e-12, worst


def test_generic_aba_itself_against_a_pendulum_known_answer():
    """The checker's own check: a point-mass pendulum (mass 2 at distance 0.7 below a revolute y joint) swings with
    qdd = -(g / L) sin(q); a prismatic joint under gravity along its axis accelerates with -g."""
    urdf = """<robot name="p"><link name="w"/><link name="b"><inertial><origin xyz="0 0 -0.7"/><mass value="2"/><inertia ixx="0" iyy="0" izz="0"/></inertial></link>
      <joint name="j" type="revolute"><axis xyz="0 1 0"/><parent link="w"/><child link="b"/></joint></robot>"""
    bodies, _ = build_tree(*parse_urdf(urdf))
    for q in (0.3, -1.1, 2.0):
        qdd, _ = aba(bodies, [q], [0.7], [0.0], [np.zeros(6)] * 2, floating=False)
        assert qdd[0] == pytest.approx(-(G / 0.7) * np.sin(q), rel=1e-13)
    urdf = urdf.replace('revolute', 'prismatic').replace('0 1 0', '0 0 1')
    bodies, _ = build_tree(*parse_urdf(urdf))
    qdd, _ = aba(bodies, [0.2], [0.0], [0.0], [np.zeros(6)] * 2, floating=False)
    assert qdd[0] == pytest.approx(-G, rel=1e-13)

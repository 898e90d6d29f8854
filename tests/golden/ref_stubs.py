"""Stand-in modules that let the REFERENCE's own Python run in this container.

Used only by tests/golden/make_golden.py (fixture generation, needs /root/reference) —
never at test time, never by the product.

The reference's hot path is pure Python on top of three wheels that are not installed
here and cannot be installed (no network): gymnasium, casadi, pybullet.  To generate
golden vectors from the reference's real code we provide the thinnest possible
stand-ins:

* ``gymnasium``: Env / Wrapper base classes, spaces.Box, utils.seeding.np_random
  (= Generator(PCG64(SeedSequence(seed))), as gymnasium implements it).
* ``casadi``: tests/golden/casadi_numeric.py, a numeric implementation of the API slice the
  reference uses (MX expression graph over NumPy, symbolic jacobian, Function, integrator):
  the reference's OWN prior-model and cost expressions evaluate (fc / df / loss / rk_discrete).
* ``pybullet`` / ``pybullet_data``: the ~25 API calls the two robots make, backed by the
  restated Bullet semantics in oracle/bullet.py (one body per client).  Geometry and
  inertial data are parsed from the URDF files the reference passes to loadURDF, so e.g.
  the 0.028 m prop offsets come from the reference's cf2x.urdf, not from a constant here.
* ``munch``, ``imageio``, ``termcolor``, ``dict_deep``: empty placeholders for imports
  that the hot path never touches.
* ``safe_control_gym``: registered as a namespace package pointing at
  /root/reference/safe_control_gym so that sub-modules import WITHOUT executing the
  package __init__ (which pulls in every controller, gpytorch, cvxpy, ...).
"""
import sys
import types
import xml.etree.ElementTree as etxml

import numpy as np

import os
import tempfile

# The reference checkout is read-only input: CPython would drop __pycache__/*.pyc next to every module imported from it.  From the
# moment anything of this repo can import the reference, bytecode goes to a scratch prefix instead (sys.pycache_prefix, PEP 3147 extension).
if sys.pycache_prefix is None:
    sys.pycache_prefix = os.path.join(tempfile.gettempdir(), 'scg_pycache')


def reference_root():
    """Where the reference checkout is on THIS machine: $SCG_REFERENCE_ROOT, the build container's /root/reference, or the
    untracked scratch copy tools/stage_reference.py makes under oracle/_ref/reference (local checker runs only: never in git,
    never on the gpurun box — .gpurunignore).  None when there is none."""
    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for cand in (os.environ.get('SCG_REFERENCE_ROOT'), '/root/reference', os.path.join(here, 'oracle', '_ref', 'reference')):
        if cand and os.path.isdir(os.path.join(cand, 'safe_control_gym')):
            return cand
    return None


REFERENCE_ROOT = reference_root() or '/root/reference'
REAL_PYBULLET = False       # set by install(): fixtures come from a real pybullet wheel instead of oracle/bullet.py


# --------------------------------------------------------------------------- #
# gymnasium
# --------------------------------------------------------------------------- #
def _np_random(seed=None):
    if seed is not None and not (isinstance(seed, (int, np.integer)) and 0 <= seed):
        raise ValueError(f'Seed must be a non-negative integer or omitted, not {seed}')
    seed_seq = np.random.SeedSequence(seed)
    return np.random.Generator(np.random.PCG64(seed_seq)), seed_seq.entropy


class _Box:
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.shape(low) if np.ndim(low) else np.shape(high)
        self._shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self._shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self._shape).copy()
        self._np_random = None
        if seed is not None:
            self.seed(seed)

    @property
    def shape(self):
        return self._shape

    def seed(self, seed=None):
        self._np_random, seed = _np_random(seed)
        return [seed]

    @property
    def np_random(self):
        if self._np_random is None:
            self.seed()
        return self._np_random

    def sample(self):
        return self.np_random.uniform(self.low, self.high, size=self._shape).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return bool(x.shape == self._shape and np.all(x >= self.low) and np.all(x <= self.high))


class _Env:
    metadata = {}

    def close(self):
        pass


class _Wrapper(_Env):
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return getattr(self.env, 'unwrapped', self.env)


def _make_gymnasium():
    gym = types.ModuleType('gymnasium')
    spaces = types.ModuleType('gymnasium.spaces')
    utils = types.ModuleType('gymnasium.utils')
    seeding = types.ModuleType('gymnasium.utils.seeding')
    spaces.Box = _Box
    seeding.np_random = _np_random
    utils.seeding = seeding
    gym.spaces, gym.utils = spaces, utils
    gym.Env, gym.Wrapper = _Env, _Wrapper
    return {'gymnasium': gym, 'gymnasium.spaces': spaces, 'gymnasium.utils': utils,
            'gymnasium.utils.seeding': seeding}


# --------------------------------------------------------------------------- #
# casadi: tests/golden/casadi_numeric.py — a NUMERIC stand-in (expression graph over NumPy with symbolic
# differentiation), so the reference's own prior-model / cost expressions (quadrotor.py:468-604, cartpole.py:390-437,
# symbolic_systems.py:68-121) evaluate instead of being absorbed.
# --------------------------------------------------------------------------- #
def _make_casadi():
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import casadi_numeric
    mods = {}
    casadi_numeric.install(mods)
    return mods


# --------------------------------------------------------------------------- #
# pybullet (restated semantics from oracle/bullet.py)
# --------------------------------------------------------------------------- #
class _QuadBody:
    kind = 'quad'

    def __init__(self, urdf, pos, quat):
        from oracle import bullet
        root = etxml.parse(urdf).getroot()
        links = root.findall('link')
        base = links[0].find('inertial')
        self.mass = float(base.find('mass').attrib['value'])
        ine = base.find('inertia').attrib
        self.J = np.array([float(ine['ixx']), float(ine['iyy']), float(ine['izz'])])
        offs = []
        for lk in links[1:5]:
            xyz = [float(v) for v in lk.find('inertial').find('origin').attrib['xyz'].split()]
            offs.append(xyz)
        offs = np.array(offs)
        # the restated step assumes the X layout (+d,+d) (-d,+d) (-d,-d) (+d,-d)
        d = offs[0, 0]
        assert np.allclose(offs, [[d, d, 0], [-d, d, 0], [-d, -d, 0], [d, -d, 0]])
        self.arm = float(d)
        self.pos = np.array(pos, dtype=float)
        self.quat = np.array(quat, dtype=float)
        self.vel, self.omega = np.zeros(3), np.zeros(3)
        self.clear()
        self.bullet = bullet

    def clear(self):
        self.prop = np.zeros(4)
        self.yaw_torque = 0.0
        self.dist = None
        self.dist_point = None

    def step(self, g, h):
        p, q, v, w = self.bullet.quadrotor_substep(
            self.pos[None], self.quat[None], self.vel[None], self.omega[None], self.prop[None],
            np.array([self.yaw_torque]), None if self.dist is None else self.dist[None],
            np.array([self.mass]), self.J[None], self.arm, g, h,
            None if self.dist_point is None else self.dist_point[None])
        self.pos, self.quat, self.vel, self.omega = p[0], q[0], v[0], w[0]
        self.clear()


class _CartPoleBody:
    kind = 'cartpole'

    def __init__(self, urdf, use_inertia_from_file):
        from oracle import bullet
        root = etxml.parse(urdf).getroot()
        links = {lk.attrib['name']: lk for lk in root.findall('link')}
        self.cart_mass = float(links['cart'].find('inertial').find('mass').attrib['value'])
        pole = links['pole']
        self.pole_mass = float(pole.find('inertial').find('mass').attrib['value'])
        self.l = float(pole.find('inertial').find('origin').attrib['xyz'].split()[-1])
        box = [float(v) for v in pole.find('collision').find('geometry').find('box').attrib['size'].split()]
        self.box = box
        self.urdf_iyy = float(pole.find('inertial').find('inertia').attrib['iyy'])
        self.use_file = use_inertia_from_file
        self._recompute_inertia()
        self.q = np.zeros(2)       # x, theta
        self.qd = np.zeros(2)
        self.bullet = bullet
        self.clear()

    def _recompute_inertia(self):
        # Bullet: without URDF_USE_INERTIA_FROM_FILE, and again on changeDynamics(mass=...),
        # the local inertia is recomputed from the collision shape (box): m/12 (lx^2 + lz^2).
        if self.use_file:
            self.ip = self.urdf_iyy
        else:
            self.ip = self.pole_mass / 12.0 * (self.box[0] ** 2 + self.box[2] ** 2)

    def clear(self):
        self.joint_force = 0.0
        self.tab = None

    def pole_com(self):
        x, th = self.q
        return (x + self.l * np.sin(th), 0.0, self.l * np.cos(th))

    def step(self, g, h):
        x, xd, th, thd = self.bullet.cartpole_substep(
            np.array([self.q[0]]), np.array([self.qd[0]]), np.array([self.q[1]]), np.array([self.qd[1]]),
            np.array([self.joint_force]), None if self.tab is None else self.tab[None],
            np.array([self.cart_mass]), np.array([self.pole_mass]), np.array([self.l]),
            np.array([self.ip]), g, h)
        self.q = np.array([x[0], th[0]])
        self.qd = np.array([xd[0], thd[0]])
        self.clear()


class _Client:
    def __init__(self):
        self.reset()

    def reset(self):
        self.bodies = {}
        self.gravity = 0.0
        self.dt = 1.0 / 240.0


def _make_pybullet():
    from oracle import bullet
    p = types.ModuleType('pybullet')
    p.GUI, p.DIRECT = 1, 2
    p.LINK_FRAME, p.WORLD_FRAME = 1, 2
    p.URDF_USE_INERTIA_FROM_FILE = 2
    p.VELOCITY_CONTROL, p.TORQUE_CONTROL, p.POSITION_CONTROL = 0, 1, 2
    p.ER_TINY_RENDERER, p.ER_BULLET_HARDWARE_OPENGL = 0, 1
    p.ER_SEGMENTATION_MASK_OBJECT_AND_LINKINDEX = 1
    p.STATE_LOGGING_VIDEO_MP4 = 3
    clients = {}

    def _c(physicsClientId=0, **_):
        return clients[physicsClientId]

    def connect(mode, **k):
        cid = min(set(range(len(clients) + 1)) - set(clients))    # Bullet hands out the lowest free id (len() would reuse a LIVE id after a disconnect)
        clients[cid] = _Client()
        return cid

    def disconnect(physicsClientId=0):
        clients.pop(physicsClientId, None)

    def resetSimulation(physicsClientId=0):
        clients[physicsClientId].reset()

    def setGravity(x, y, z, physicsClientId=0):
        clients[physicsClientId].gravity = -z

    def setTimeStep(dt, physicsClientId=0):
        clients[physicsClientId].dt = dt

    def loadURDF(fileName, basePosition=(0, 0, 0), baseOrientation=(0, 0, 0, 1), flags=0,
                 physicsClientId=0, **k):
        c = clients[physicsClientId]
        bid = len(c.bodies)
        if fileName.endswith('plane.urdf'):
            c.bodies[bid] = None
        elif 'cartpole' in fileName:
            c.bodies[bid] = _CartPoleBody(fileName, bool(flags & p.URDF_USE_INERTIA_FROM_FILE))
        else:
            c.bodies[bid] = _QuadBody(fileName, basePosition, baseOrientation)
        return bid

    def changeDynamics(bodyUniqueId, linkIndex, mass=None, localInertiaDiagonal=None,
                       linearDamping=None, angularDamping=None, physicsClientId=0, **k):
        b = clients[physicsClientId].bodies[bodyUniqueId]
        if b.kind == 'quad':
            assert linkIndex == -1
            if mass is not None:
                b.mass = float(mass)
            if localInertiaDiagonal is not None:
                b.J = np.array(localInertiaDiagonal, dtype=float)
        else:
            if mass is not None:
                if linkIndex == 0:
                    b.cart_mass = float(mass)
                elif linkIndex == 1:
                    b.pole_mass = float(mass)
                    b.use_file = False          # changeDynamics(mass) recomputes from the collision shape
                    b._recompute_inertia()

    def getQuaternionFromEuler(rpy, **k):
        return tuple(bullet.quaternion_from_euler(np.asarray(rpy, dtype=float)))

    def getEulerFromQuaternion(q, **k):
        return tuple(bullet.euler_from_quaternion(np.asarray(q, dtype=float)))

    def getMatrixFromQuaternion(q, **k):
        return tuple(bullet.matrix_from_quaternion(np.asarray(q, dtype=float)).reshape(-1))

    def resetBasePositionAndOrientation(bid, pos, quat, physicsClientId=0):
        b = clients[physicsClientId].bodies[bid]
        b.pos, b.quat = np.array(pos, dtype=float), np.array(quat, dtype=float)

    def resetBaseVelocity(bid, linearVelocity=None, angularVelocity=None, physicsClientId=0):
        b = clients[physicsClientId].bodies[bid]
        b.vel, b.omega = np.array(linearVelocity, dtype=float), np.array(angularVelocity, dtype=float)

    def getBasePositionAndOrientation(bid, physicsClientId=0):
        b = clients[physicsClientId].bodies[bid]
        return tuple(b.pos), tuple(b.quat)

    def getBaseVelocity(bid, physicsClientId=0):
        b = clients[physicsClientId].bodies[bid]
        return tuple(b.vel), tuple(b.omega)

    def applyExternalForce(objectUniqueId, linkIndex, forceObj, posObj, flags, physicsClientId=0):
        b = clients[physicsClientId].bodies[objectUniqueId]
        f = np.array(forceObj, dtype=float)
        if b.kind == 'quad':
            if flags == p.LINK_FRAME:
                assert 0 <= linkIndex < 4 and f[0] == 0 and f[1] == 0 and not np.any(np.asarray(posObj))
                b.prop[linkIndex] += f[2]
            else:
                # world-frame force at an arbitrary world point: Bullet adds (posObj - origin) x F
                assert linkIndex == 4 and b.dist is None
                b.dist = f
                b.dist_point = np.array(posObj, dtype=float)
        else:
            assert flags == p.WORLD_FRAME and linkIndex == 1 and np.allclose(posObj, b.pole_com())
            t = np.array([f[0], f[2]])
            b.tab = t if b.tab is None else b.tab + t

    def applyExternalTorque(objectUniqueId, linkIndex, torqueObj, flags, physicsClientId=0):
        b = clients[physicsClientId].bodies[objectUniqueId]
        assert b.kind == 'quad' and flags == p.LINK_FRAME and torqueObj[0] == 0 and torqueObj[1] == 0
        b.yaw_torque += torqueObj[2]

    def setJointMotorControl2(bodyUniqueId, jointIndex, controlMode, force=0.0, physicsClientId=0, **k):
        b = clients[physicsClientId].bodies[bodyUniqueId]
        if controlMode == p.TORQUE_CONTROL:
            assert jointIndex == 0
            b.joint_force += float(force)
        else:
            assert force == 0        # default velocity motors switched off (cartpole.py:310-311)

    def resetJointState(bodyUniqueId, jointIndex, targetValue, targetVelocity=0.0, physicsClientId=0):
        b = clients[physicsClientId].bodies[bodyUniqueId]
        b.q[jointIndex], b.qd[jointIndex] = targetValue, targetVelocity

    def getJointState(bodyUniqueId, jointIndex, physicsClientId=0):
        b = clients[physicsClientId].bodies[bodyUniqueId]
        return (b.q[jointIndex], b.qd[jointIndex], (0.0,) * 6, 0.0)

    def getLinkState(bodyUniqueId, linkIndex, physicsClientId=0, **k):
        b = clients[physicsClientId].bodies[bodyUniqueId]
        assert linkIndex == 1
        return (b.pole_com(),)

    def stepSimulation(physicsClientId=0):
        c = clients[physicsClientId]
        for b in c.bodies.values():
            if b is not None:
                b.step(c.gravity, c.dt)

    def _noop(*a, **k):
        return None

    for name, fn in list(locals().items()):
        if callable(fn) and not name.startswith('_') and name not in ('p', 'bullet', 'clients'):
            setattr(p, name, fn)
    for name in ('setRealTimeSimulation', 'setAdditionalSearchPath', 'setPhysicsEngineParameter',
                 'resetDebugVisualizerCamera', 'computeViewMatrixFromYawPitchRoll',
                 'computeProjectionMatrixFOV', 'addUserDebugLine', 'stopStateLogging', 'startStateLogging'):
        setattr(p, name, _noop)
    pd = types.ModuleType('pybullet_data')
    pd.getDataPath = lambda: ''
    return {'pybullet': p, 'pybullet_data': pd}


def install():
    """Install all stand-ins into sys.modules and make ``safe_control_gym`` importable lazily."""
    mods = {}
    mods.update(_make_gymnasium())
    mods.update(_make_casadi())
    try:                                        # a machine that HAS the wheel generates the fixtures from real Bullet
        import pybullet  # noqa: F401
        import pybullet_data  # noqa: F401
        global REAL_PYBULLET
        REAL_PYBULLET = True
    except ImportError:
        mods.update(_make_pybullet())
    for name in ('munch', 'imageio', 'termcolor', 'dict_deep'):
        mods[name] = types.ModuleType(name)
    mods['munch'].munchify = lambda x: x
    mods['munch'].Munch = dict
    mods['termcolor'].colored = lambda s, *a, **k: s
    for k, v in mods.items():
        sys.modules.setdefault(k, v)
    pkg = types.ModuleType('safe_control_gym')
    pkg.__path__ = [REFERENCE_ROOT + '/safe_control_gym']
    sys.modules.setdefault('safe_control_gym', pkg)
    for sub in ('controllers', 'envs/env_wrappers'):
        name = 'safe_control_gym.' + sub.replace('/', '.')
        m = types.ModuleType(name)
        m.__path__ = [REFERENCE_ROOT + '/safe_control_gym/' + sub]
        sys.modules.setdefault(name, m)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

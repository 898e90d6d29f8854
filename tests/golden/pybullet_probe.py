#!/usr/bin/env python3
"""PyBullet probes: settle every "recalled" Bullet statement behind oracle/bullet.py.

Run on the first machine that has the `pybullet` wheel AND the reference checkout:

    python tests/golden/pybullet_probe.py [--reference /root/reference] [--write-trace]

Without pybullet the script prints "SKIP: pybullet not importable" and exits 0 (this container: no wheel, no network).
Each probe drives REAL Bullet through the same calls the reference makes (base_aviary.py:212-226,273-282,364-384;
cartpole.py:281-342,557-583), compares with what oracle/bullet.py predicts and prints PASS / FAIL with the numbers, so a
FAIL names the assumption that is wrong.  `--write-trace` additionally stores a short real-Bullet trajectory of both
robots in tests/golden/pybullet_trace.npz; tests/test_oracle_golden.py::test_real_bullet_trace (auto-skipped while the
file is absent) replays it through the oracle.  `make_golden.py` itself switches to the real module when it is importable
(ref_stubs.install), so regenerating the fixtures on such a machine re-pins everything above `p.stepSimulation` too.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

RESULTS = []


def report(name, ok, detail):
    RESULTS.append(ok)
    print(f'[{"PASS" if ok else "FAIL"}] {name}: {detail}')


def load_drone(p, ref, client, pos=(0, 0, 1.0), rpy=(0, 0, 0)):
    urdf = os.path.join(ref, 'safe_control_gym/envs/gym_pybullet_drones/assets/cf2x.urdf')
    p.setGravity(0, 0, -9.8, physicsClientId=client)
    p.setRealTimeSimulation(0, physicsClientId=client)
    bid = p.loadURDF(urdf, pos, p.getQuaternionFromEuler(rpy), physicsClientId=client)      # base_aviary.py:221-225
    p.changeDynamics(bid, -1, linearDamping=0, angularDamping=0, physicsClientId=client)     # :226
    return bid


def probe_integrator(p, ref):
    """1. drop with zero rpm: semi-implicit Euler z_n = z_0 - g h^2 n (n + 1) / 2 (explicit: n (n - 1) / 2)."""
    c = p.connect(p.DIRECT)
    h, n = 1e-3, 50
    p.setTimeStep(h, physicsClientId=c)
    d = load_drone(p, ref, c)
    for _ in range(n):
        p.stepSimulation(physicsClientId=c)
    z = p.getBasePositionAndOrientation(d, physicsClientId=c)[0][2]
    semi, expl = 1.0 - 9.8 * h * h * n * (n + 1) / 2, 1.0 - 9.8 * h * h * n * (n - 1) / 2
    report('1 integrator is semi-implicit Euler', abs(z - semi) < 1e-12, f'z = {z:.15f}, semi-implicit {semi:.15f}, explicit {expl:.15f}')
    p.disconnect(c)


def probe_damping(p, ref):
    """2. constant world force on the COM link, zero damping: v_n = n h F / m exactly (no clamp below 100)."""
    c = p.connect(p.DIRECT)
    h, n, F, m = 1e-3, 100, 0.05, 0.027
    p.setTimeStep(h, physicsClientId=c)
    p.setGravity(0, 0, 0, physicsClientId=c)
    urdf = os.path.join(ref, 'safe_control_gym/envs/gym_pybullet_drones/assets/cf2x.urdf')
    d = p.loadURDF(urdf, (0, 0, 1), (0, 0, 0, 1), physicsClientId=c)
    p.changeDynamics(d, -1, linearDamping=0, angularDamping=0, physicsClientId=c)
    for _ in range(n):
        pos = p.getBasePositionAndOrientation(d, physicsClientId=c)[0]
        p.applyExternalForce(d, 4, forceObj=[F, 0, 0], posObj=pos, flags=p.WORLD_FRAME, physicsClientId=c)
        p.stepSimulation(physicsClientId=c)
    v = p.getBaseVelocity(d, physicsClientId=c)[0][0]
    report('2 no hidden damping, forces cleared each step', abs(v - n * h * F / m) < 1e-10, f'v = {v:.12f}, expected {n * h * F / m:.12f}')
    p.disconnect(c)


def probe_force_point(p, ref):
    """3. one prop force for one substep: d(omega_body) = h J^-1 [d F, -d F, -KM r^2], d = 0.028 (cf2x.urdf:42)."""
    from oracle import bullet
    c = p.connect(p.DIRECT)
    h = 1e-3
    p.setTimeStep(h, physicsClientId=c)
    d = load_drone(p, ref, c)
    KF, KM, rpm = 3.16e-10, 7.94e-12, 14000.0
    p.applyExternalForce(d, 0, forceObj=[0, 0, KF * rpm ** 2], posObj=[0, 0, 0], flags=p.LINK_FRAME, physicsClientId=c)
    p.applyExternalTorque(d, 4, torqueObj=[0, 0, -KM * rpm ** 2], flags=p.LINK_FRAME, physicsClientId=c)
    p.stepSimulation(physicsClientId=c)
    w = np.array(p.getBaseVelocity(d, physicsClientId=c)[1])
    pos, quat, vel, om = bullet.quadrotor_substep(np.array([[0, 0, 1.0]]), np.array([[0, 0, 0, 1.0]]), np.zeros((1, 3)), np.zeros((1, 3)),
                                                  np.array([[KF * rpm ** 2, 0, 0, 0]]), np.array([-KM * rpm ** 2]), None,
                                                  np.array([0.027]), np.array([[1.4e-5, 1.4e-5, 2.17e-5]]), 0.028, 9.8, h)
    report('3 prop force applied at the link inertial origin (d = 0.028)', np.allclose(w, om[0], rtol=1e-9, atol=1e-12),
           f'omega real {w}, oracle {om[0]}')
    p.disconnect(c)


def probe_gyro(p, ref):
    """4. free spin: d(omega_body) = -h J^-1 (w x J w), explicit."""
    from oracle import bullet
    c = p.connect(p.DIRECT)
    h = 1e-3
    p.setTimeStep(h, physicsClientId=c)
    p.setGravity(0, 0, 0, physicsClientId=c)
    d = load_drone(p, ref, c)
    p.setGravity(0, 0, 0, physicsClientId=c)
    w0 = np.array([3.0, -2.0, 5.0])
    p.resetBaseVelocity(d, [0, 0, 0], list(w0), physicsClientId=c)
    p.stepSimulation(physicsClientId=c)
    w = np.array(p.getBaseVelocity(d, physicsClientId=c)[1])
    _, _, _, om = bullet.quadrotor_substep(np.array([[0, 0, 1.0]]), np.array([[0, 0, 0, 1.0]]), np.zeros((1, 3)), w0[None],
                                           np.zeros((1, 4)), np.zeros(1), None, np.array([0.027]),
                                           np.array([[1.4e-5, 1.4e-5, 2.17e-5]]), 0.028, 0.0, h)
    report('4 explicit gyroscopic term', np.allclose(w, om[0], rtol=1e-9, atol=1e-12), f'omega real {w}, oracle {om[0]}, no-gyro {w0}')
    p.disconnect(c)


def probe_pole_inertia(ref):
    """5. pole inertia right after CartPole.reset(): collision box m (0.05^2 + (2l)^2) / 12 or URDF rod m (2l)^2 / 12."""
    from tests.golden import ref_stubs
    ref_stubs.install()
    import pybullet as p
    from safe_control_gym.envs.gym_control.cartpole import CartPole
    env = CartPole(output_dir='/tmp')
    env.reset()
    info = p.getDynamicsInfo(env.CARTPOLE_ID, 1, physicsClientId=env.PYB_CLIENT)
    iyy = info[2][1]
    m, l = env.POLE_MASS, env.EFFECTIVE_POLE_LENGTH
    box, rod = m * (0.05 ** 2 + (2 * l) ** 2) / 12, m * (2 * l) ** 2 / 12
    which = 'box' if abs(iyy - box) < abs(iyy - rod) else 'rod'
    report("5 pole inertia is the collision box's (oracle default pole_inertia='box')", which == 'box',
           f'Iyy = {iyy:.9g}, box {box:.9g}, rod {rod:.9g} -> {which}')
    env.close()


def probe_ground(p, ref):
    """6. ground plane at z = -0.05 (base_aviary.py:107,219): no contact force for z >= 0."""
    import pybullet_data
    c = p.connect(p.DIRECT)
    p.setTimeStep(1e-3, physicsClientId=c)
    p.setAdditionalSearchPath(pybullet_data.getDataPath(), physicsClientId=c)
    p.loadURDF('plane.urdf', [0, 0, -0.05], physicsClientId=c)
    d = load_drone(p, ref, c, pos=(0, 0, 0.0))
    for _ in range(3):
        p.stepSimulation(physicsClientId=c)
    vz = p.getBaseVelocity(d, physicsClientId=c)[0][2]
    report('6 no ground contact at z = 0', abs(vz + 3 * 9.8e-3) < 1e-9, f'v_z after 3 free-fall steps = {vz:.9f} (free fall {-3 * 9.8e-3:.9f})')
    z_touch = None
    for _ in range(400):
        p.stepSimulation(physicsClientId=c)
        z, vz = p.getBasePositionAndOrientation(d, physicsClientId=c)[0][2], p.getBaseVelocity(d, physicsClientId=c)[0][2]
        if vz > -1e-3 and z_touch is None:
            z_touch = z
    print(f'      contact arrests the fall at base z ~ {z_touch} (outside the parity envelope: oracle models no ground)')
    p.disconnect(c)


def probe_end_to_end(ref, write_trace):
    """7. the reference's own envs on real Bullet vs the oracle on the same seed (same PCG64 initial states) and actions."""
    from tests.golden import ref_stubs
    ref_stubs.install()
    from oracle.envs import make_oracle_env, make_rng
    from safe_control_gym.envs.gym_control.cartpole import CartPole
    from safe_control_gym.envs.gym_pybullet_drones.quadrotor import Quadrotor
    from tests.golden.make_golden import load_task_config
    out = {}
    for name, cls, task, over in (('cartpole', CartPole, 'cartpole', 'examples/rl/config_overrides/cartpole/cartpole_stab.yaml'),
                                  ('quadrotor_3D', Quadrotor, 'quadrotor', 'examples/rl/config_overrides/quadrotor_3D/quadrotor_3D_track.yaml')):
        cfg = load_task_config(task, over)
        cfg.pop('seed', None)
        cfg.update(output_dir='/tmp', done_on_out_of_bound=False)
        env = cls(seed=5, **cfg)
        env.reset()
        ocfg = {k: v for k, v in cfg.items() if k != 'output_dir'}
        orc = make_oracle_env(task, 1, make_rng('numpy', 1, 5), **ocfg)     # the reference's own PCG64 stream, same seed
        orc.reset()
        assert np.allclose(orc.state[0], env.state, rtol=1e-12, atol=1e-12), 'initial states differ (seeding)'
        rng = np.random.default_rng(0)
        states, ostates, acts = [], [], []
        for _ in range(300):
            a = rng.uniform(-1, 1, size=env.action_dim)
            env.step(a)
            orc.step(a[None])
            states.append(np.asarray(env.state).copy()); ostates.append(orc.state[0].copy()); acts.append(a)
        states, ostates = np.array(states), np.array(ostates)
        rel = np.max(np.abs(states - ostates), axis=0) / np.maximum(np.max(np.abs(states), axis=0), 1e-9)
        report(f'7 {name}: 300 control steps, real Bullet vs oracle (north_star bar 1e-4 relative per state dim)', bool(np.all(rel < 1e-4)),
               f'max relative deviation per dim {np.array2string(rel, precision=2)}')
        out[name + '/states'], out[name + '/actions'], out[name + '/x0'] = states, np.array(acts), states[0]
        import yaml
        out[name + '/config_yaml'] = yaml.safe_dump(ocfg)
        env.close()
    if write_trace:
        np.savez_compressed(os.path.join(HERE, 'pybullet_trace.npz'), **out)
        print('tests/golden/pybullet_trace.npz written')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', default='/root/reference')
    ap.add_argument('--write-trace', action='store_true')
    args = ap.parse_args()
    try:
        import pybullet as p
    except ImportError:
        print('SKIP: pybullet not importable (no wheel in this image, no network) — nothing probed')
        return 0
    if not os.path.isdir(args.reference):
        print(f'SKIP: reference checkout {args.reference} not found')
        return 0
    probe_integrator(p, args.reference)
    probe_damping(p, args.reference)
    probe_force_point(p, args.reference)
    probe_gyro(p, args.reference)
    probe_pole_inertia(args.reference)
    probe_ground(p, args.reference)
    probe_end_to_end(args.reference, args.write_trace)
    print(f'{sum(RESULTS)} / {len(RESULTS)} probes passed')
    return 0 if all(RESULTS) else 1


if __name__ == '__main__':
    sys.exit(main())

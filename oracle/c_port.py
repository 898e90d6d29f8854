"""ctypes driver of oracle/scg_oracle.c (OpenMP CPU port of the control step).  ORACLE — test infrastructure only.

`build()` compiles oracle/_ref/libscg_oracle.so with gcc (called from __graft_entry__.build()); `CPort` configures it
from a NumPy oracle env (constants, box-constraint rows, X_GOAL) so both run the very same task config.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'scg_oracle.c')
OUT_DIR = os.path.join(HERE, '_ref')
LIB = os.path.join(OUT_DIR, 'libscg_oracle.so')

MAXS, MAXA, MAXR = 12, 4, 64
i32, f64 = C.c_int32, C.c_double


class OcCfg(C.Structure):
    _fields_ = [('system', i32), ('n', i32), ('nx', i32), ('nu', i32), ('ns', i32), ('nobs', i32),
                ('substeps', i32), ('ctrl_steps', i32), ('tracking', i32), ('goal_rows', i32), ('goal_horizon', i32),
                ('normalized', i32), ('done_on_oob', i32), ('randomized_init', i32), ('env_id_offset', i32),
                ('n_rows', i32), ('pad', i32), ('seed', C.c_uint64),
                ('pyb_dt', f64), ('act_scale', f64), ('hover', f64), ('goal_tolerance', f64),
                ('act_low', f64 * MAXA), ('act_high', f64 * MAXA),
                ('kf', f64), ('km', f64), ('pwm_scale', f64), ('pwm_const', f64), ('pwm_min', f64), ('pwm_max', f64),
                ('g', f64), ('arm', f64), ('vmax', f64), ('pole_box_w', f64), ('param', f64 * 4),
                ('rew_sw', f64 * MAXS), ('rew_aw', f64 * MAXA), ('u_goal', f64 * MAXA), ('mse_w', f64 * MAXS),
                ('state_low', f64 * MAXS), ('state_high', f64 * MAXS), ('x_thr', f64), ('th_thr', f64),
                ('init_state', f64 * MAXS), ('init_lo', f64 * MAXS), ('init_hi', f64 * MAXS), ('init_rand', i32 * MAXS),
                ('row_var', i32 * MAXR), ('row_idx', i32 * MAXR), ('row_sign', f64 * MAXR), ('row_b', f64 * MAXR),
                ('x_goal', C.POINTER(f64))]


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    # no -march=native: the .so is built in the build container and travels to the GPU box (different host CPU)
    cmd = ['gcc', '-O3', '-fopenmp', '-fPIC', '-shared', '-o', LIB, SRC, '-lm']
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError('gcc failed:\n' + res.stderr)
    return LIB


class CPort:
    def __init__(self, oracle_env, seed, env_id_offset=0):
        e = oracle_env
        self.lib = C.CDLL(build())
        assert self.lib.oc_sizeof_cfg() == C.sizeof(OcCfg)
        assert not e.disturbances and e.COST == 'rl_reward' and e.rew_exponential and not e.RANDOMIZED_INERTIAL_PROP
        c = OcCfg()
        cart = e.NAME == 'cartpole'
        c.system = 0 if cart else int(e.QUAD_TYPE)
        assert c.system in (0, 2, 3)
        c.n, c.nx, c.nu, c.nobs = e.num_envs, e.state_dim, e.action_dim, e.obs_dim
        c.ns = {0: 4, 2: 6, 3: 13}[c.system]
        c.substeps, c.ctrl_steps = e.PYB_STEPS_PER_CTRL, int(np.ceil(e.CTRL_STEPS))      # counter >= float CTRL_STEPS (benchmark_env.py:499)
        c.tracking = int(e.TASK == 'traj_tracking')
        xg = np.ascontiguousarray(np.atleast_2d(e.X_GOAL), dtype=np.float64)
        self._xg = xg
        c.goal_rows, c.x_goal = xg.shape[0], xg.ctypes.data_as(C.POINTER(f64))
        c.goal_horizon = (e.obs_dim // e.state_dim - 1)
        c.normalized, c.done_on_oob = int(e.NORMALIZED_RL_ACTION_SPACE), int(e.done_on_out_of_bound)
        c.randomized_init, c.env_id_offset, c.seed = int(e.RANDOMIZED_INIT), env_id_offset, int(seed)
        c.pyb_dt = e.PYB_TIMESTEP
        c.goal_tolerance = float(e.TASK_INFO.get('stabilization_goal_tolerance', 0.0)) if not c.tracking else 0.0
        for j in range(c.nu):
            c.act_low[j], c.act_high[j] = float(e.physical_action_bounds[0][j]), float(e.physical_action_bounds[1][j])
            c.u_goal[j] = float(e.U_GOAL[j])
        raw = np.broadcast_to(e.rew_act_weight, (c.nu,)) if e.rew_act_weight.size == 1 else e.rew_act_weight
        rsw = np.broadcast_to(e.rew_state_weight, (c.nx,)) if e.rew_state_weight.size == 1 else e.rew_state_weight
        for j in range(c.nu):
            c.rew_aw[j] = float(raw[j])
        for k in range(c.nx):
            c.rew_sw[k], c.mse_w[k] = float(rsw[k]), float(e.info_mse_metric_state_weight[k])
            c.state_low[k], c.state_high[k] = float(e.state_space_low[k]), float(e.state_space_high[k])
        c.g, c.vmax = e.GRAVITY_ACC, 100.0
        if cart:
            c.act_scale, c.hover = float(e.action_scale), 0.0
            c.param[0], c.param[1], c.param[2] = e.EFFECTIVE_POLE_LENGTH, e.CART_MASS, e.POLE_MASS
            c.pole_box_w = 0.05 if e.pole_inertia_mode == 'box' else 0.0
            c.x_thr, c.th_thr = e.x_threshold, e.theta_threshold_radians
            labels, info = e.INIT_NAMES, e.INIT_STATE_RAND_INFO
        else:
            c.act_scale, c.hover = float(e.norm_act_scale), float(getattr(e, 'hover_thrust', 0.0))
            c.param[0], c.param[1], c.param[2], c.param[3] = e.MASS, e.J[0], e.J[1], e.J[2]
            c.kf, c.km, c.pwm_scale, c.pwm_const = e.KF, e.KM, e.PWM2RPM_SCALE, e.PWM2RPM_CONST
            c.pwm_min, c.pwm_max, c.arm = e.MIN_PWM, e.MAX_PWM, e.PROP_OFFSET
            labels, info = e.INIT_STATE_LABELS[e.QUAD_TYPE], e.INIT_STATE_RAND_INFO
        for k, name in enumerate(labels):
            c.init_state[k] = float(e.init_values[name])
            if name in info:
                assert info[name]['distrib'] == 'uniform'
                c.init_rand[k], c.init_lo[k], c.init_hi[k] = 1, float(info[name]['low']), float(info[name]['high'])
        rows = []
        if e.constraints is not None:
            for con in e.constraints.constraints:
                A = con.A @ con.constraint_filter           # box rows only: exactly one +-1 per row
                for r in range(A.shape[0]):
                    nz = np.nonzero(A[r])[0]
                    assert len(nz) == 1 and abs(A[r, nz[0]]) == 1.0 and not con.strict and con.decimals == 8
                    rows.append((0 if con.constrained_variable == 'state' else 1, int(nz[0]), float(A[r, nz[0]]), float(con.b[r])))
        c.n_rows = len(rows)
        for q, (v, ix, sg, b) in enumerate(rows):
            c.row_var[q], c.row_idx[q], c.row_sign[q], c.row_b[q] = v, ix, sg, b
        self.cfg = c
        n = c.n
        self.state = np.empty((n, c.ns))
        self.step_ctr = np.zeros(n, dtype=np.int32)
        self.episode = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
        self.obs = np.empty((n, c.nobs))
        self.rew = np.zeros(n)
        self.done = np.zeros(n, dtype=np.uint8)
        self.flags = np.zeros(n, dtype=np.uint8)
        self.cvals = np.empty((n, max(c.n_rows, 1)))
        self.mse = np.zeros(n)
        self.term_obs = np.empty((n, c.nobs))
        # first touch by the OpenMP threads that will own the rows (NUMA placement on multi-socket hosts)
        self.lib.oc_touch(C.byref(self.cfg), self._p(self.state, f64), self._p(self.obs, f64), self._p(self.cvals, f64),
                          self._p(self.term_obs, f64))

    @staticmethod
    def _p(a, t):
        return a.ctypes.data_as(C.POINTER(t))

    def reset(self):
        self.lib.oc_reset(C.byref(self.cfg), self._p(self.state, f64), self._p(self.step_ctr, i32),
                          self._p(self.episode, C.c_uint32), self._p(self.obs, f64))
        return self.obs

    def step(self, action):
        a = np.ascontiguousarray(action, dtype=np.float64)
        self.lib.oc_step(C.byref(self.cfg), self._p(self.state, f64), self._p(self.step_ctr, i32),
                         self._p(self.episode, C.c_uint32), self._p(a, f64), self._p(self.obs, f64), self._p(self.rew, f64),
                         self._p(self.done, C.c_uint8), self._p(self.flags, C.c_uint8), self._p(self.cvals, f64),
                         self._p(self.mse, f64), self._p(self.term_obs, f64))
        return self.obs, self.rew, self.done.astype(bool)

    def run(self, actions, k_steps):
        """k_steps control steps with the action ring `actions` [ring, n, nu]; no barrier between steps (envs are
        independent), outputs hold the last step.  This is the loop bench.py times as the CPU baseline."""
        a = np.ascontiguousarray(actions, dtype=np.float64)
        self.lib.oc_run(C.byref(self.cfg), self._p(self.state, f64), self._p(self.step_ctr, i32),
                        self._p(self.episode, C.c_uint32), self._p(a, f64), int(a.shape[0]), int(k_steps), self._p(self.obs, f64),
                        self._p(self.rew, f64), self._p(self.done, C.c_uint8), self._p(self.flags, C.c_uint8),
                        self._p(self.cvals, f64), self._p(self.mse, f64), self._p(self.term_obs, f64))
        return self.obs, self.rew, self.done.astype(bool)

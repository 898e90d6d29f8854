/* scg_oracle.c — plain-C (double precision, OpenMP over environments) restatement of the control step.
 *
 * ORACLE — test infrastructure only (see oracle/__init__.py).  It exists to time the CPU path on all host cores
 * (bench.py `cpu_baseline`, kind "port") and is itself validated against the NumPy oracle (tests/test_oracle_c_port.py),
 * which is pinned to the reference's own Python by tests/golden/.  Feature scope = what the shipped RL task configs use
 * (examples/rl/config_overrides/{cartpole/cartpole_stab, quadrotor_2D/quadrotor_2D_track, quadrotor_3D/quadrotor_3D_track}.yaml):
 * normalised actions, rl_reward (exponential), stabilisation / trajectory tracking with a 0/1-row goal horizon, box
 * constraints, done on out-of-bounds, time-limit truncation, auto-reset with uniform initial-state randomisation.
 * Disturbances, randomised inertia, quadratic cost, adversary: NumPy oracle only.
 *
 * Follows (paths relative to /root/reference/safe_control_gym/envs):
 *   gym_pybullet_drones/quadrotor.py:722-775 (_preprocess_control), quadrotor_utils.py:16-60 (cmd2pwm / pwm2rpm),
 *   base_aviary.py:232-286,364-384 (_advance_simulation / _physics) with Bullet's step as restated in oracle/bullet.py,
 *   quadrotor.py:777-923 (obs / reward / done / info), gym_control/cartpole.py:479-696, benchmark_env.py:422-502
 *   (extend_obs, after_step), constraints.py:97-131,320-322 (box rows, float32 bounds, round to 8 decimals),
 *   env_wrappers/vectorized_env/dummy_vec_env.py:29-41 (auto-reset), quadrotor.py:328-392 / cartpole.py:266-352 (reset).
 * Random draws: Philox4x32-10 with the addressing of oracle/rng.py.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <string.h>

#define OC_MAX_STATE 12
#define OC_MAX_ACTION 4
#define OC_MAX_ROWS 64

typedef struct {
    int32_t system;            /* 0 cartpole, 2 quadrotor 2D, 3 quadrotor 3D */
    int32_t n, nx, nu, ns, nobs;
    int32_t substeps, ctrl_steps, tracking, goal_rows, goal_horizon;
    int32_t normalized, done_on_oob, randomized_init, env_id_offset;
    int32_t n_rows, pad;
    uint64_t seed;
    double pyb_dt, act_scale, hover, goal_tolerance;
    double act_low[OC_MAX_ACTION], act_high[OC_MAX_ACTION];
    double kf, km, pwm_scale, pwm_const, pwm_min, pwm_max, g, arm, vmax, pole_box_w;
    double param[4];           /* cartpole: l, M, m | quadrotor: M, Ixx, Iyy, Izz */
    double rew_sw[OC_MAX_STATE], rew_aw[OC_MAX_ACTION], u_goal[OC_MAX_ACTION], mse_w[OC_MAX_STATE];
    double state_low[OC_MAX_STATE], state_high[OC_MAX_STATE], x_thr, th_thr;
    double init_state[OC_MAX_STATE], init_lo[OC_MAX_STATE], init_hi[OC_MAX_STATE];
    int32_t init_rand[OC_MAX_STATE];
    int32_t row_var[OC_MAX_ROWS], row_idx[OC_MAX_ROWS];
    double row_sign[OC_MAX_ROWS], row_b[OC_MAX_ROWS];
    const double* x_goal;      /* [goal_rows][nx] */
} oc_cfg;

/* ---- Philox4x32-10 (oracle/rng.py) ---- */
static void philox(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
static double u01(uint32_t w) { return ((double)(w >> 8) + 0.5) * (1.0 / 16777216.0); }

static double normalize_angle(double x) {      /* math_and_models/normalization.py:8-10 */
    double y = x + M_PI;
    y = y - 2.0 * M_PI * floor(y / (2.0 * M_PI));
    return y - M_PI;
}
static double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

static void quat_to_mat(const double* q, double R[3][3]) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double d = x * x + y * y + z * z + w * w, s = 2.0 / d;
    double xs = x * s, ys = y * s, zs = z * s, wx = w * xs, wy = w * ys, wz = w * zs;
    double xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
    R[0][0] = 1.0 - (yy + zz); R[0][1] = xy - wz; R[0][2] = xz + wy;
    R[1][0] = xy + wz; R[1][1] = 1.0 - (xx + zz); R[1][2] = yz - wx;
    R[2][0] = xz - wy; R[2][1] = yz + wx; R[2][2] = 1.0 - (xx + yy);
}
static void quat_to_euler(const double* q, double* rpy) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double sarg = -2.0 * (x * z - w * y);
    if (sarg <= -0.99999) { rpy[0] = 0; rpy[1] = -0.5 * M_PI; rpy[2] = 2 * atan2(x, -y); }
    else if (sarg >= 0.99999) { rpy[0] = 0; rpy[1] = 0.5 * M_PI; rpy[2] = 2 * atan2(-x, y); }
    else {
        rpy[0] = atan2(2 * (y * z + w * x), w * w - x * x - y * y + z * z);
        rpy[1] = asin(sarg);
        rpy[2] = atan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z);
    }
}

/* env.state from the raw simulator state (quadrotor.py:784-802) */
static void state_vector(const oc_cfg* c, const double* s, double* st) {
    if (c->system == 0) { memcpy(st, s, 4 * sizeof(double)); return; }
    if (c->system == 2) {
        for (int k = 0; k < 6; ++k) st[k] = s[k];
        double th = s[4];
        if (fabs(th) >= 1.5663) {
            double sa = sin(th);
            st[4] = sa <= -0.99999 ? -0.5 * M_PI : (sa >= 0.99999 ? 0.5 * M_PI : asin(sa));
        }
        return;
    }
    double R[3][3], rpy[3];
    quat_to_mat(s + 3, R);
    quat_to_euler(s + 3, rpy);
    st[0] = s[0]; st[1] = s[7]; st[2] = s[1]; st[3] = s[8]; st[4] = s[2]; st[5] = s[9];
    st[6] = rpy[0]; st[7] = rpy[1]; st[8] = rpy[2];
    for (int j = 0; j < 3; ++j) st[9 + j] = R[0][j] * s[10] + R[1][j] * s[11] + R[2][j] * s[12];
}

static void write_obs(const oc_cfg* c, const double* st, int next_index, double* dst) {
    for (int k = 0; k < c->nx; ++k) dst[k] = st[k];
    if (c->goal_horizon > 0) {
        if (c->tracking) {
            for (int r = 0; r < c->goal_horizon; ++r) {
                int row = next_index + r; if (row > c->goal_rows - 1) row = c->goal_rows - 1;
                for (int k = 0; k < c->nx; ++k) dst[c->nx * (1 + r) + k] = c->x_goal[row * c->nx + k];
            }
        } else {
            for (int k = 0; k < c->nx; ++k) dst[c->nx + k] = c->x_goal[k];
        }
    }
}

static void reset_env(const oc_cfg* c, int i, double* s, int32_t* step, uint32_t* episode) {
    *episode += 1u;
    *step = 0;
    double iv[OC_MAX_STATE];
    for (int k = 0; k < c->nx; ++k) iv[k] = c->init_state[k];
    if (c->randomized_init) {
        /* uniform draws only: the compact layout of scg_rng.h (variable j = 21-bit field j % 6 of block j / 6, left-aligned
           in a word: fields 0-3 the top 21 bits of the four words, 4 and 5 their 11 / 10 low bits paired up) */
        for (int b = 0; b < (c->nx + 5) / 6; ++b) {
            uint32_t ctr[4] = {(uint32_t)(c->env_id_offset + i), *episode, 0u, (0u << 16) | (0u << 8) | (uint32_t)b};
            philox(ctr, (uint32_t)(c->seed & 0xffffffffu), (uint32_t)(c->seed >> 32));
            for (int k = 0; k < 6 && 6 * b + k < c->nx; ++k) {
                const int j = 6 * b + k;
                uint32_t w;
                if (k < 4) w = ctr[k] & 0xfffff800u;
                else w = (ctr[k == 4 ? 0 : 2] << 21) | ((ctr[k == 4 ? 1 : 3] & 0x3ffu) << 11);
                if (c->init_rand[j]) iv[j] += c->init_lo[j] + (c->init_hi[j] - c->init_lo[j]) * u01(w);
            }
        }
    }
    if (c->system == 3) {
        s[0] = iv[0]; s[1] = iv[2]; s[2] = iv[4];
        double hr = 0.5 * iv[6], hp = 0.5 * iv[7], hy = 0.5 * iv[8];
        double cr = cos(hr), sr = sin(hr), cp = cos(hp), sp = sin(hp), cy = cos(hy), sy = sin(hy);
        s[3] = sr * cp * cy - cr * sp * sy; s[4] = cr * sp * cy + sr * cp * sy;
        s[5] = cr * cp * sy - sr * sp * cy; s[6] = cr * cp * cy + sr * sp * sy;
        s[7] = iv[1]; s[8] = iv[3]; s[9] = iv[5];
        s[10] = iv[9]; s[11] = iv[10]; s[12] = iv[11];
    } else {
        for (int k = 0; k < c->ns; ++k) s[k] = iv[k];
    }
}

void oc_reset(const oc_cfg* c, double* state, int32_t* step, uint32_t* episode, double* obs) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < c->n; ++i) {
        double st[OC_MAX_STATE];
        reset_env(c, i, state + (size_t)i * c->ns, step + i, episode + i);
        state_vector(c, state + (size_t)i * c->ns, st);
        write_obs(c, st, 1, obs + (size_t)i * c->nobs);
    }
}

/* One vectorised control step with auto-reset.  flags: bit0 truncated, bit1 violation, bit2 out_of_bounds. */
static inline void step_env(const oc_cfg* c, int i, double* state, int32_t* step, uint32_t* episode, const double* action,
                            double* obs, double* rew, unsigned char* done, unsigned char* flags, double* cvals,
                            double* mse_out, double* term_obs) {
    {
        double* s = state + (size_t)i * c->ns;
        const int c0 = step[i];
        double noisy[OC_MAX_ACTION], clipped[OC_MAX_ACTION];
        for (int j = 0; j < c->nu; ++j) {
            double a = action[(size_t)i * c->nu + j];
            if (c->normalized) a = c->system == 0 ? c->act_scale * a : (1.0 + c->act_scale * a) * c->hover;
            noisy[j] = a;
            clipped[j] = clampd(a, c->act_low[j], c->act_high[j]);
        }
        const double h = c->pyb_dt, vmax = c->vmax;
        if (c->system == 0) {
            const double l = c->param[0], M = c->param[1], m = c->param[2];
            const double ip = m * (c->pole_box_w * c->pole_box_w + 4.0 * l * l) / 12.0;
            double x = s[0], xd = s[1], th = s[2], thd = s[3];
            for (int k = 0; k < c->substeps; ++k) {
                double sn = sin(th), cs = cos(th);
                double a11 = M + m, a12 = m * l * cs, a22 = ip + m * l * l;
                double b1 = clipped[0] + m * l * thd * thd * sn, b2 = m * c->g * l * sn;
                double det = a11 * a22 - a12 * a12;
                double xdd = (a22 * b1 - a12 * b2) / det, thdd = (a11 * b2 - a12 * b1) / det;
                xd = clampd(xd + h * xdd, -vmax, vmax); thd = clampd(thd + h * thdd, -vmax, vmax);
                x += h * xd; th += h * thd;
            }
            s[0] = x; s[1] = xd; s[2] = th; s[3] = thd;
        } else {
            double pwm[4], f[4], tq[4];
            const int n_motor = 4 / c->nu;
            for (int j = 0; j < c->nu; ++j) {
                double thr = clipped[j] > 0 ? clipped[j] : 0.0;
                pwm[j] = (sqrt(thr / n_motor / c->kf) - c->pwm_const) / c->pwm_scale;
            }
            if (c->nu == 2) { pwm[2] = pwm[1]; pwm[3] = pwm[0]; }
            for (int j = 0; j < 4; ++j) {
                double p = clampd(pwm[j], c->pwm_min, c->pwm_max), rpm = c->pwm_scale * p + c->pwm_const;
                f[j] = rpm * rpm * c->kf; tq[j] = rpm * rpm * c->km;
            }
            const double thrust = f[0] + f[1] + f[2] + f[3], mass = c->param[0];
            if (c->system == 2) {
                const double tau = c->arm * (-f[0] + f[1] + f[2] - f[3]), iyy = c->param[2];
                double x = s[0], vx = s[1], z = s[2], vz = s[3], th = s[4], w = s[5];
                for (int k = 0; k < c->substeps; ++k) {
                    double sn = sin(th), cs = cos(th);
                    w = clampd(w + h * (tau / iyy), -vmax, vmax);
                    vx = clampd(vx + h * (sn * thrust / mass), -vmax, vmax);
                    vz = clampd(vz + h * (cs * thrust / mass - c->g), -vmax, vmax);
                    x += h * vx; z += h * vz; th += h * w;
                }
                s[0] = x; s[1] = vx; s[2] = z; s[3] = vz; s[4] = th; s[5] = w;
            } else {
                const double J[3] = {c->param[1], c->param[2], c->param[3]};
                const double tb[3] = {c->arm * (f[0] + f[1] - f[2] - f[3]), c->arm * (-f[0] + f[1] + f[2] - f[3]),
                                      -tq[0] + tq[1] - tq[2] + tq[3]};
                double* p = s; double* q = s + 3; double* v = s + 7; double* w = s + 10;
                for (int k = 0; k < c->substeps; ++k) {
                    double R[3][3], wb[3], wd[3], jw[3];
                    quat_to_mat(q, R);
                    for (int j = 0; j < 3; ++j) wb[j] = R[0][j] * w[0] + R[1][j] * w[1] + R[2][j] * w[2];
                    for (int j = 0; j < 3; ++j) jw[j] = J[j] * wb[j];
                    wd[0] = (tb[0] - (wb[1] * jw[2] - wb[2] * jw[1])) / J[0];
                    wd[1] = (tb[1] - (wb[2] * jw[0] - wb[0] * jw[2])) / J[1];
                    wd[2] = (tb[2] - (wb[0] * jw[1] - wb[1] * jw[0])) / J[2];
                    double acc[3] = {R[0][2] * thrust / mass, R[1][2] * thrust / mass, R[2][2] * thrust / mass - c->g};
                    for (int j = 0; j < 3; ++j)
                        w[j] = clampd(w[j] + h * (R[j][0] * wd[0] + R[j][1] * wd[1] + R[j][2] * wd[2]), -vmax, vmax);
                    for (int j = 0; j < 3; ++j) v[j] = clampd(v[j] + h * acc[j], -vmax, vmax);
                    for (int j = 0; j < 3; ++j) p[j] += h * v[j];
                    double ang = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
                    if (ang * h > 0.25 * M_PI) ang = 0.25 * M_PI / h;
                    double kk = ang < 0.001 ? 0.5 * h - h * h * h * 0.020833333333 * ang * ang : sin(0.5 * ang * h) / ang;
                    double dx = w[0] * kk, dy = w[1] * kk, dz = w[2] * kk, dw = cos(0.5 * ang * h);
                    double qx = dw * q[0] + dx * q[3] + dy * q[2] - dz * q[1];
                    double qy = dw * q[1] - dx * q[2] + dy * q[3] + dz * q[0];
                    double qz = dw * q[2] + dx * q[1] - dy * q[0] + dz * q[3];
                    double qw = dw * q[3] - dx * q[0] - dy * q[1] - dz * q[2];
                    double inv = 1.0 / sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
                    q[0] = qx * inv; q[1] = qy * inv; q[2] = qz * inv; q[3] = qw * inv;
                }
            }
        }
        double st[OC_MAX_STATE], ref[OC_MAX_STATE];
        state_vector(c, s, st);
        int row = 0;
        if (c->tracking) { row = c0 + 1; if (row > c->goal_rows - 1) row = c->goal_rows - 1; }
        for (int k = 0; k < c->nx; ++k) ref[k] = c->x_goal[row * c->nx + k];
        /* reward (quadrotor.py:826-845, cartpole.py:617-635) */
        double dist = 0.0;
        for (int k = 0; k < c->nx; ++k) {
            double sv = st[k];
            if (c->system == 0 && k == 2) sv = normalize_angle(sv);
            dist += c->rew_sw[k] * (sv - ref[k]) * (sv - ref[k]);
        }
        for (int j = 0; j < c->nu; ++j) dist += c->rew_aw[j] * (noisy[j] - c->u_goal[j]) * (noisy[j] - c->u_goal[j]);
        double r = exp(-dist);
        /* done (quadrotor.py:864-894, cartpole.py:654-672) */
        int dn = 0; unsigned char fl = 0;
        int goal = 0;
        if (!c->tracking) {
            double n2 = 0; for (int k = 0; k < c->nx; ++k) n2 += (st[k] - ref[k]) * (st[k] - ref[k]);
            goal = sqrt(n2) < c->goal_tolerance; dn = goal;
        }
        if (c->done_on_oob && !goal) {
            int oob = 0;
            if (c->system == 0) oob = st[0] < -c->x_thr || st[0] > c->x_thr || st[2] < -c->th_thr || st[2] > c->th_thr;
            else for (int k = 0; k < c->nx; ++k) {
                int masked = c->system == 3 ? ((k < 6 && (k & 1) == 0) || (k >= 6 && k < 9)) : ((k & 1) == 0);
                if (masked && (st[k] < c->state_low[k] || st[k] > c->state_high[k])) oob = 1;
            }
            if (oob) { fl |= 4; dn = 1; }
        }
        /* mse (quadrotor.py:907-922) */
        double mse = 0.0;
        for (int k = 0; k < c->nx; ++k) {
            double sv = st[k];
            if (c->tracking && ((c->system == 0 && k == 2) || (c->system == 2 && k == 4) || (c->system == 3 && k >= 6 && k < 9)))
                sv = normalize_angle(sv);
            double e = (sv - ref[k]) * c->mse_w[k];
            mse += e * e;
        }
        /* constraints (constraints.py:97-131) */
        int viol = 0;
        for (int q = 0; q < c->n_rows; ++q) {
            double vv = c->row_var[q] == 0 ? st[c->row_idx[q]] : noisy[c->row_idx[q]];
            double cv = rint((c->row_sign[q] * vv - c->row_b[q]) * 1e8) / 1e8;
            if (cv > 0.0) viol = 1;
            if (cvals) cvals[(size_t)i * c->n_rows + q] = cv;
        }
        if (viol) fl |= 2;
        step[i] = c0 + 1;
        if (step[i] >= c->ctrl_steps) { if (!dn) fl |= 1; dn = 1; }
        rew[i] = r; done[i] = (unsigned char)dn; flags[i] = fl; if (mse_out) mse_out[i] = mse;
        if (dn) {
            if (term_obs) write_obs(c, st, c0 + 2, term_obs + (size_t)i * c->nobs);
            reset_env(c, i, s, step + i, episode + i);
            state_vector(c, s, st);
            write_obs(c, st, 1, obs + (size_t)i * c->nobs);
        } else {
            write_obs(c, st, c0 + 2, obs + (size_t)i * c->nobs);
        }
    }
}

void oc_step(const oc_cfg* c, double* state, int32_t* step, uint32_t* episode, const double* action, double* obs,
             double* rew, unsigned char* done, unsigned char* flags, double* cvals, double* mse_out, double* term_obs) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < c->n; ++i)
        step_env(c, i, state, step, episode, action, obs, rew, done, flags, cvals, mse_out, term_obs);
}

/* k_steps control steps of every env with the actions of a ring of `ring` pre-generated batches ([ring][n][nu]); the envs are
 * independent, so each thread advances its own envs through all the steps without a barrier per step (the output arrays
 * hold the last step's values, like the GPU benchmark that reuses its output buffers).  This is the CPU baseline loop. */
void oc_run(const oc_cfg* c, double* state, int32_t* step, uint32_t* episode, const double* actions, int ring, int k_steps,
            double* obs, double* rew, unsigned char* done, unsigned char* flags, double* cvals, double* mse_out, double* term_obs) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < c->n; ++i)
        for (int t = 0; t < k_steps; ++t)
            step_env(c, i, state, step, episode, actions + (size_t)(t % ring) * c->n * c->nu, obs, rew, done, flags, cvals,
                     mse_out, term_obs);
}

/* first-touch initialisation of the per-env arrays by the threads that will own them (NUMA placement) */
void oc_touch(const oc_cfg* c, double* state, double* obs, double* cvals, double* term_obs) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < c->n; ++i) {
        for (int k = 0; k < c->ns; ++k) state[(size_t)i * c->ns + k] = 0.0;
        for (int k = 0; k < c->nobs; ++k) { obs[(size_t)i * c->nobs + k] = 0.0; term_obs[(size_t)i * c->nobs + k] = 0.0; }
        for (int k = 0; k < c->n_rows; ++k) cvals[(size_t)i * c->n_rows + k] = 0.0;
    }
}

int oc_sizeof_cfg(void) { return (int)sizeof(oc_cfg); }

/* OMP_NUM_THREADS is read once per process (the host program may have initialised the OpenMP runtime long before) */
void oc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int oc_get_threads(void) { return omp_get_max_threads(); }

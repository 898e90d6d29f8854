/* scg_hip.h — C ABI of libscg_hip.so: MI355X (gfx950) batched simulator + rollout kernels for the
 * safe-control-gym CartPole / Quadrotor environments.
 *
 * This is the drop-in boundary.  The reference has no native code: its hot path is
 * Python calling PyBullet once per environment.  Each entry point below replaces the reference
 * interface cited next to it (paths relative to /root/reference/safe_control_gym); the ctypes
 * binding a maintainer adds on the reference side is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; every `d_*` pointer is a DEVICE pointer owned by the caller (e.g. a torch
 *     tensor's data_ptr()); host pointers are named `h_*`.
 *   - element type of all floating-point device buffers = cfg.dtype (SCG_F32 float, SCG_F64 double).
 *   - all kernels are enqueued on the caller's hipStream_t (`stream`, may be NULL = default stream)
 *     and return without synchronising, except the h_* accessors which synchronise that stream.
 *   - return 0 on success, a negative scg_status otherwise; scg_last_error() returns a thread-local
 *     message.  The library never aborts and never falls back to a CPU path.
 *   - a handle is bound to one device and is not thread-safe.
 */
#ifndef SCG_HIP_H
#define SCG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCG_ABI_VERSION 1

#define SCG_MAX_STATE 12        /* quadrotor 3D: x xd y yd z zd phi theta psi p q r */
#define SCG_MAX_ACTION 4
#define SCG_MAX_GOAL_HORIZON 4
#define SCG_MAX_OBS (SCG_MAX_STATE * (1 + SCG_MAX_GOAL_HORIZON))
#define SCG_MAX_CON_ROWS 64     /* scalar constraint rows, all constraints stacked */
#define SCG_MAX_QUAD_CON 2
#define SCG_MAX_DISTURB 4       /* disturbances per channel */
#define SCG_MAX_PARAM 4         /* per-env inertial parameters */
#define SCG_MAX_CHOICE 8

typedef enum {
    SCG_OK = 0,
    SCG_ERR_INVALID = -1,       /* bad argument / unsupported configuration */
    SCG_ERR_HIP = -2,           /* a HIP runtime call failed */
    SCG_ERR_NOMEM = -3,
    SCG_ERR_STATE = -4          /* call order violated (e.g. step before reset) */
} scg_status;

typedef enum { SCG_F32 = 0, SCG_F64 = 1 } scg_dtype;

/* envs/__init__.py:5-11 ids 'cartpole' / 'quadrotor' + quadrotor_utils.py:8-13 QuadType. */
typedef enum { SCG_CARTPOLE = 0, SCG_QUAD_1D = 1, SCG_QUAD_2D = 2, SCG_QUAD_3D = 3 } scg_system;
/* benchmark_env.py:21-32 */
typedef enum { SCG_TASK_STABILIZATION = 0, SCG_TASK_TRAJ_TRACKING = 1 } scg_task;
typedef enum { SCG_COST_RL_REWARD = 0, SCG_COST_QUADRATIC = 1 } scg_cost;
/* integrator: PYB_EULER reproduces p.stepSimulation at pyb_freq (base_aviary.py:282, cartpole.py:583);
 * RK4 integrates the symbolic prior model once per control step (controllers/mpc/mpc_utils.py:42-64). */
typedef enum { SCG_INT_PYB_EULER = 0, SCG_INT_RK4 = 1 } scg_integrator;

/* disturbances.py:277-283 */
typedef enum { SCG_DIST_NONE = 0, SCG_DIST_IMPULSE = 1, SCG_DIST_STEP = 2, SCG_DIST_UNIFORM = 3,
               SCG_DIST_WHITE = 4, SCG_DIST_PERIODIC = 5 } scg_dist_kind;
typedef enum { SCG_CH_ACTION = 0, SCG_CH_DYNAMICS = 1, SCG_CH_OBSERVATION = 2 } scg_channel;

/* One entry of a `disturbances: {channel: [...]}` list (disturbances.py:70-259).  Per-dimension
 * vectors already include the mask, exactly as the reference would evaluate `noise * mask`. */
typedef struct {
    int32_t kind;               /* scg_dist_kind */
    int32_t dim;
    int32_t step_offset;        /* impulse/step: fixed offset, or -1 = integers(max_step) at reset */
    int32_t max_step;           /* int(EPISODE_LEN_SEC / CTRL_TIMESTEP) */
    double duration;            /* impulse */
    double decay_rate;          /* impulse */
    double frequency;           /* periodic */
    double a[SCG_MAX_STATE];    /* impulse/step: magnitude*mask | uniform: low | white: std | periodic: scale */
    double b[SCG_MAX_STATE];    /* uniform: high */
    double mask[SCG_MAX_STATE]; /* uniform/white: multiplicative mask (1 when absent) */
} scg_disturbance;

/* benchmark_env.py:237-268 `{distrib: ..., low/high | loc/scale | args}`; value += sample. */
typedef enum { SCG_RAND_NONE = 0, SCG_RAND_UNIFORM = 1, SCG_RAND_NORMAL = 2, SCG_RAND_CHOICE = 3 } scg_rand_kind;
typedef struct {
    int32_t kind;               /* scg_rand_kind */
    int32_t n_choice;
    double p0, p1;              /* uniform: low, high | normal: loc, scale */
    double choices[SCG_MAX_CHOICE];
} scg_rand;

/* One scalar row of the stacked constraint vector (constraints.py:97-109,273):
 *   SPARSE: c = sign * v[index] - b      (bounded / default / abs_bound rows)
 *   DENSE : c = coef . v - b             (linear_constraint rows, coef = A @ filter)
 * v = state (var 0) or the NOISY UNCLIPPED physical action (var 1), constraints.py:155-174.
 * ABS   : c = |v[index]| - b             (SymmetricStateConstraint.get_value, :445-447) */
typedef enum { SCG_ROW_SPARSE = 0, SCG_ROW_DENSE = 1, SCG_ROW_ABS = 2, SCG_ROW_QUADRATIC = 3 } scg_row_kind;
typedef struct {
    int32_t kind;               /* scg_row_kind */
    int32_t var;                /* 0 state, 1 input */
    int32_t index;              /* SPARSE/ABS: variable index; QUADRATIC: index into quad_P */
    int32_t strict;             /* violated also when c == 0 (constraints.py:127-130) */
    double sign;
    double b;
    double round_scale;         /* 10^decimals (np.round, constraints.py:109); 0 = no rounding */
    double coef[SCG_MAX_STATE]; /* DENSE */
} scg_con_row;

typedef struct {
    int32_t abi_version;        /* SCG_ABI_VERSION */
    int32_t system;             /* scg_system */
    int32_t dtype;              /* scg_dtype */
    int32_t integrator;         /* scg_integrator */
    int32_t num_envs;
    int32_t env_id_offset;      /* global env id = env_id_offset + local index (rank * N for env shards) */
    uint64_t seed;              /* Philox key */

    /* timing — benchmark_env.py:139-148 */
    int32_t substeps;           /* PYB_FREQ / CTRL_FREQ */
    int32_t ctrl_steps;         /* EPISODE_LEN_SEC * CTRL_FREQ */
    double pyb_dt;              /* 1 / PYB_FREQ */
    double ctrl_dt;             /* 1 / CTRL_FREQ */

    /* task / cost — quadrotor.py:150-201,261-323; cartpole.py:152-167,215-233 */
    int32_t task;               /* scg_task */
    int32_t cost;               /* scg_cost */
    int32_t obs_goal_horizon;
    int32_t goal_rows;          /* rows of X_GOAL passed to scg_create (1 for stabilisation) */
    int32_t rew_exponential;
    int32_t done_on_out_of_bound;
    int32_t done_on_violation;
    int32_t use_constraint_penalty;
    int32_t obs_wrap_angle;     /* cartpole.py:598-599 */
    int32_t normalized_action;  /* normalized_rl_action_space */
    int32_t info_goal_reached;  /* stabilisation + quadratic cost: expose goal_reached */
    int32_t auto_reset;         /* 1: DummyVecEnv semantics (reset inside the step where done);
                                   0: single-env semantics, the caller resets (BenchmarkEnv.step) */
    double goal_tolerance;      /* TASK_INFO['stabilization_goal_tolerance'] */
    double constraint_penalty;
    double rew_state_weight[SCG_MAX_STATE];
    double rew_act_weight[SCG_MAX_ACTION];
    double q_diag[SCG_MAX_STATE];           /* quadratic cost: diag(Q), diag(R) (lqr_utils.py:77-99) */
    double r_diag[SCG_MAX_ACTION];
    double mse_weight[SCG_MAX_STATE];       /* info_mse_metric_state_weight */
    double u_goal[SCG_MAX_ACTION];
    double state_low[SCG_MAX_STATE];        /* state_space bounds (float32-rounded), quadrotor.py:697 */
    double state_high[SCG_MAX_STATE];
    double x_threshold;                     /* cartpole.py:452-454 */
    double theta_threshold;

    /* action pre-processing — quadrotor.py:722-775, quadrotor_utils.py:16-60, cartpole.py:479-530 */
    double act_scale;           /* quadrotor: norm_act_scale; cartpole: action_scale (10) */
    double hover_thrust;        /* quadrotor: g * URDF_mass / action_dim (quadrotor.py:630) */
    double act_low[SCG_MAX_ACTION];         /* physical_action_bounds (float32-rounded) */
    double act_high[SCG_MAX_ACTION];
    double kf, km, pwm2rpm_scale, pwm2rpm_const, pwm_min, pwm_max;   /* cf2x.urdf:5 */

    /* physics — cf2x.urdf / cartpole_template.urdf */
    double gravity;             /* 9.8 */
    double arm;                 /* prop offset used by the engine: 0.028 (PyBullet) or L/sqrt(2) (prior) */
    double max_coordinate_velocity;         /* Bullet clamps every velocity coordinate to +-100 */
    double pole_box_width;      /* cartpole: collision box width for Bullet's recomputed pole inertia;
                                   0 => slender rod m (2l)^2 / 12 (cartpole.py:296) */
    /* inertial parameters: quadrotor M, Ixx, Iyy, Izz | cartpole pole_length, cart_mass, pole_mass */
    double base_param[SCG_MAX_PARAM];
    int32_t randomized_inertial_prop;
    int32_t randomized_init;
    scg_rand param_rand[SCG_MAX_PARAM];
    /* initial state in INIT_STATE_LABELS order (quadrotor.py:209-214; cartpole.py:324) */
    double init_state[SCG_MAX_STATE];
    scg_rand init_rand[SCG_MAX_STATE];

    /* disturbances / adversary — benchmark_env.py:216-228,279-295 */
    int32_t n_dist[3];          /* per scg_channel */
    int32_t adversary_channel;  /* -1 none, else scg_channel (action or dynamics) */
    scg_disturbance dist[3][SCG_MAX_DISTURB];
    double adversary_scale, adversary_offset;

    /* constraints — constraints.py */
    int32_t n_con_rows;
    int32_t n_state_con_rows;   /* rows whose var == state (evaluated at reset, benchmark_env.py:356-357) */
    scg_con_row con[SCG_MAX_CON_ROWS];
    double quad_P[SCG_MAX_QUAD_CON][SCG_MAX_STATE * SCG_MAX_STATE];  /* filter' P filter, row-major nx*nx */
} scg_config;

/* Outputs of one vectorised control step (dummy_vec_env.py:29-41 `step_wait` +
 * record_episode_statistics.py:139-166).  d_obs, d_reward, d_done, d_flags are always written; any other pointer
 * may be NULL = output not wanted.  Row-major arrays; T = the env's dtype.  Per-env rows ([N][k]) must be 16-byte
 * aligned.  The step kernel is fastest when all bound arrays lie within one 4 GiB window (e.g. carved from one
 * allocation): it then addresses them through a single buffer resource. */
typedef struct {
    void* d_obs;            /* [N][obs_dim]  observation AFTER auto-reset (post-reset obs where done) */
    void* d_reward;         /* [N] */
    uint8_t* d_done;        /* [N] 0/1 */
    uint8_t* d_flags;       /* [N] bit0 TimeLimit.truncated, bit1 constraint_violation, bit2 out_of_bounds,
                                   bit3 goal_reached, bit4 ground_contact (quadrotors: z at / below the reference world's
                                   ground plane, base_aviary.py:107,219-220 — Bullet's contact response is not modelled, the
                                   flag marks steps from which the trajectory is no longer the reference's) */
    void* d_c_values;       /* [n_con_rows][N]  info['constraint_values'] of the step (pre-reset), one row per
                                   constraint so that every wave stores contiguous 256-byte segments;
                                   at reset: the state rows, densely in rows 0..n_state_con_rows-1 */
    void* d_mse;            /* [N] info['mse'] */
    void* d_terminal_obs;   /* [N][obs_dim]  written only where done: info['terminal_observation'] */
    void* d_state;          /* [state_dim][N] env.state after the step and auto-reset */
    void* d_noisy_action;   /* [action_dim][N] current_noisy_physical_action (pre-clip) */
    /* columnar VecRecordEpisodeStatistics, one 4-vector per env: (return, length, sum of constraint_violation,
     * sum of mse), all of type T.  d_ep_stats is read-modify-written every step (zeroed where done);
     * d_fin_stats receives the finished episode's totals where done. */
    void* d_ep_stats;       /* [N][4] running totals */
    void* d_fin_stats;      /* [N][4] */
} scg_step_out;

/* Totals of a fused K-step random-action rollout (config #2 of BASELINE.json). */
typedef struct {
    void* d_reward_sum;     /* [N] sum of rewards over the K steps */
    int32_t* d_done_count;  /* [N] number of episodes finished */
    int32_t* d_violation_count; /* [N] */
    void* d_last_obs;       /* [N][obs_dim] */
} scg_rollout_out;

typedef struct scg_env scg_env;

/* Shapes derived from a config (BenchmarkEnv.state_dim / action_dim / obs_dim, benchmark_env.py:170-175). */
int scg_dims(const scg_config* cfg, int32_t* state_dim, int32_t* action_dim, int32_t* obs_dim,
             int32_t* n_state_arrays, int32_t* n_params);

/* Bytes of caller-owned device workspace that hold the SoA simulator state (body state, per-env
 * inertial parameters, disturbance offsets, step/episode counters). */
int scg_workspace_bytes(const scg_config* cfg, size_t* bytes);

/* Replaces env construction: utils/registration.py:123-125 make('cartpole'|'quadrotor', **task_config)
 * x N inside vectorized_env/__init__.py:42-66 make_vec_envs.  `h_x_goal` is the X_GOAL table
 * (goal_rows x state_dim, row-major doubles; benchmark_env.py:504-558, quadrotor.py:261-323). */
int scg_create(const scg_config* cfg, const double* h_x_goal, int device, void* d_workspace,
               size_t workspace_bytes, scg_env** out);
int scg_destroy(scg_env* env);

/* Replaces VecEnv.reset() (dummy_vec_env.py:43-48 -> Quadrotor.reset quadrotor.py:328-392 /
 * CartPole.reset cartpole.py:266-352).  d_mask: NULL = reset every env, else [N] bytes, non-zero =
 * reset that env.  Writes out->d_obs (+ d_c_values for state rows, d_state) for the reset envs. */
int scg_reset(scg_env* env, const uint8_t* d_mask, const scg_step_out* out, void* stream);

/* Replaces VecEnv.step_async/step_wait (dummy_vec_env.py:24-41 -> Quadrotor.step quadrotor.py:394-445 /
 * CartPole.step cartpole.py:238-264, incl. auto-reset).  d_action: [N][action_dim];
 * d_adv_action: [N][adv_dim] already scaled by set_adversary_control (benchmark_env.py:216-228) or NULL. */
int scg_step(scg_env* env, const void* d_action, const void* d_adv_action, const scg_step_out* out,
             void* stream);

/* The same control step for the env range [first_env, first_env + n_envs) only (first_env a multiple of 64).  All
 * array arguments are the FULL [N]-sized arrays; only the rows of that range are read and written.  Envs are
 * independent (dummy_vec_env.py:29-41 is a serial loop over them), so disjoint ranges of one handle may be advanced by
 * separate calls, in any order, from any streams.  (Measured on ROCm 7.2 / MI355X: sub-shard launches on 2-4 streams do NOT
 * overlap — the queues are drained one kernel at a time and the launch floor is paid per range; tools/README.md.  The
 * entry point is for callers that own only part of a handle's envs, not a throughput device.) */
int scg_step_range(scg_env* env, int first_env, int n_envs, const void* d_action, const void* d_adv_action,
                   const scg_step_out* out, void* stream);

/* K control steps per launch with in-kernel actions ~ U(-1, 1) (Philox channel 4); same per-step
 * semantics as scg_step, state kept in registers between steps. */
int scg_rollout_random(scg_env* env, int k_steps, const scg_rollout_out* out, void* stream);

/* K control steps per launch with CALLER-SUPPLIED action sequences: exactly
 *     for t in range(K): obs[t], rew[t], done[t], info[t] = vec_env.step(actions[t])
 * (dummy_vec_env.py:24-41 called K times: same per-step semantics as scg_step incl. auto-reset, terminal observation and
 * episode statistics) with the state kept in registers between steps.  For open-loop consumers: sampling-based MPC scoring
 * candidate action sequences, replay of logged actions, system identification.  Every per-step output is [K]-stacked. */
typedef struct {
    const void* d_actions;      /* [K][N][action_dim] */
    const void* d_adv_actions;  /* [K][N][adv_dim] already scaled (benchmark_env.py:216-228), or NULL */
    void* d_obs;                /* [K][N][obs_dim]  observation returned by step t (post auto-reset) */
    void* d_reward;             /* [K][N] */
    uint8_t* d_done;            /* [K][N] */
    uint8_t* d_flags;           /* [K][N] bits as in scg_step_out */
    void* d_terminal_obs;       /* [K][N][obs_dim] written where done, or NULL */
    void* d_mse;                /* [K][N] or NULL */
    void* d_c_values;           /* [K][n_con_rows][N] or NULL */
    void* d_ep_stats;           /* [N][4] running episode totals (read at launch, written back at the end) or NULL */
    void* d_fin_stats;          /* [K][N][4] totals of the episodes that finished at step t, or NULL */
    void* d_state;              /* [K][state_dim][N] env.state after step t and auto-reset, or NULL */
    void* d_noisy_action;       /* [K][action_dim][N] current_noisy_physical_action of step t, or NULL */
} scg_sequence;
int scg_step_sequence(scg_env* env, int k_steps, const scg_sequence* seq, void* stream);

/* K control steps per launch with the POLICY in the loop — the rollout collector of PPO.train_step
 * (controllers/ppo/ppo.py:266-284) and the evaluation loop of PPO.run (:210-257) as ONE launch: per step the actor MLP
 * obs -> H -> H -> act_dim (ppo_utils.py:149-199; exact float32 on the matrix cores), action = mean + exp(logstd) N(0,1)
 * (Philox channel 5 of the env's stream) or the mean, log-probability, then the same control step as scg_step.
 * Only libraries specialised for a task config AND a policy shape implement it (libscg_spec_<hash>_pol<H>_<act>.so, see
 * safe_control_gym_amd/_lib.py); others return SCG_ERR_INVALID.  float32 envs, single-row observations. */
typedef struct {
    const float* d_params;          /* flat float32 parameter vector */
    int32_t W1, b1, W2, b2, W3, b3; /* offsets (floats) of the actor's pi_net.fcs.{0,1,2}.{weight,bias} (nn.Linear layout) */
    int32_t logstd_off;             /* offset of actor.logstd [act_dim] */
    int32_t hidden, activation;     /* must equal the library's compiled policy shape (0 tanh | 1 relu | 2 leaky_relu) */
    int32_t deterministic;          /* 1: action = mean (evaluation) */
} scg_policy;
typedef struct {
    void* d_obs;                    /* [k + 1][N][obs_dim]: row 0 = observation of the current state, row t + 1 = after step t */
    void* d_act;                    /* [k][N][act_dim] */
    void* d_logp;                   /* [k][N] */
    void* d_reward;                 /* [k][N] */
    uint8_t* d_done;                /* [k][N] */
    uint8_t* d_flags;               /* [k][N] bits as in scg_step_out */
    void* d_terminal_obs;           /* [k][N][obs_dim], written where done (nullable) */
    void* d_ep_stats;               /* [N][4] running episode totals, shared with scg_step's accumulator (nullable) */
    void* d_episode_acc;            /* [N][8] += per finished episode: count, return, length, violation steps, mse sum, 0, 0, 0 (nullable) */
    int32_t max_episodes;           /* > 0: an env stops adding to d_episode_acc after that many episodes (evaluation) */
} scg_policy_rollout;
int scg_rollout_policy(scg_env* env, const scg_policy* policy, int k_steps, const scg_policy_rollout* out, void* stream);

/* Parity-test / checkpoint accessors (host side, synchronise `stream`).
 * Raw simulator state, n_state_arrays doubles per env:
 *   cartpole  x, x_dot, theta, theta_dot
 *   quad 1D   z, z_dot             quad 2D  x, x_dot, z, z_dot, theta(unwrapped), theta_dot
 *   quad 3D   pos[3], quat[4] (x,y,z,w body->world), vel[3], ang_vel[3] (world) */
int scg_set_state(scg_env* env, const double* h_state, int first_env, int n, void* stream);
int scg_get_state(scg_env* env, double* h_state, int first_env, int n, void* stream);
int scg_set_params(scg_env* env, const double* h_params, int first_env, int n, void* stream);
int scg_get_params(scg_env* env, double* h_params, int first_env, int n, void* stream);
/* Tuning knobs (no reference counterpart; results are identical bit for bit whichever launch runs).  Config-specialised libraries pick
 * the geometry of scg_step's launch by the size of the shard:
 *   >= wide_min_envs    256-thread workgroups (the largest shards; default 8 388 608 or env SCG_WIDE_MIN_ENVS);
 *   wsback_min_envs .. wsback_max_envs (Quadrotor systems; default 131 072 .. 524 288 or env SCG_WSBACK_MIN_ENVS / SCG_WSBACK_MAX_ENVS,
 *                       scg_set_step_wsback): one-wave workgroups with the handle's WORKSPACE arrays stored write-back — the next
 *                       launch's wave of the same env group reads them from the same XCD's L2 (-3 .. -9 % per launch there);
 *   otherwise, and the generic library always: one wave per 64 envs in one-wave workgroups.
 * A negative argument leaves a threshold unchanged; INT_MAX switches the wide launch off; scg_set_step_wsback(env, 1, 0) (an empty
 * range) the write-back launch.  split_max_envs is accepted and IGNORED: the two-waves-per-64-envs launch of rounds 5-6 was removed
 * (correctly ordered it is slower than the one-wave launch: csrc/scg_env_kernels.h). */
int scg_set_step_launch(scg_env* env, int split_max_envs, int wide_min_envs);
int scg_set_step_wsback(scg_env* env, int wsback_min_envs, int wsback_max_envs);
/* Version of the Philox WORD LAYOUT of the reset draws (which bits of which block feed which initial-state / parameter / offset
 * variable).  Seeded runs and checkpoints reproduce only within one version:
 *   1  rounds 1-4: one 32-bit word per one-word draw, four per block;
 *   2  round 5 on: six 21-bit fields per block for the compact (uniform / one-word) draws (csrc/scg_rng.h, oracle/rng.py).
 * Checkpoints carry it (HipVecEnv.get_env_random_state) and a load under another version warns. */
#define SCG_RNG_LAYOUT_VERSION 2
int scg_rng_layout_version(void);
/* BenchmarkEnv.seed (benchmark_env.py:193-214): new Philox key for subsequent draws. */
int scg_set_seed(scg_env* env, uint64_t seed);
/* ctrl_step_counter / episode index per env (benchmark_env.py:329-330). */
int scg_set_counters(scg_env* env, const int32_t* h_step, const uint32_t* h_episode, int first_env, int n, void* stream);
int scg_get_counters(scg_env* env, int32_t* h_step, uint32_t* h_episode, int first_env, int n, void* stream);

/* Batched prior-model services for model-based controllers (math_and_models/symbolic_systems.py:68-121 fc_func / df_func /
 * fd_func; controllers/lqr/lqr_utils.py, controllers/mpc/mpc_utils.py:42-64 rk_discrete): for n samples (x [n][state_dim],
 * u [n][action_dim], env dtype, device pointers) writes, where the pointer is not NULL, f(x,u) [n][nx], the continuous-time
 * Jacobians A = df/dx [n][nx][nx] and B = df/du [n][nx][nu] (central differences with step eps, as df_func does
 * numerically here since CasADi is not available) and the state after one RK4 step of the control period [n][nx].
 * Equations and inertial parameters: the env's config (prior-model arm L/sqrt(2) when integrator = SCG_INT_RK4 or
 * engine_arm = symbolic). */
int scg_prior_model(scg_env* env, const void* d_x, const void* d_u, int n, double eps, void* d_f, void* d_A, void* d_B,
                    void* d_xnext, void* stream);

/* Replaces controllers/ppo/ppo_utils.py:374-400 compute_returns_and_advantages on [T][N] buffers:
 *   rew += gamma * terminal_v;  ret_t = rew_t + gamma m_t ret_{t+1};
 *   GAE: adv_t = delta_t + gamma lambda m_t adv_{t+1}  else adv_t = ret_t - v_t.
 * dtype-typed device arrays; d_rew is updated in place like the reference. */
int scg_gae(int dtype, void* d_rew, const void* d_v, const void* d_mask, const void* d_terminal_v,
            const void* d_last_v, void* d_ret, void* d_adv, int T, int N, double gamma, double lam,
            int use_gae, void* stream);

const char* scg_last_error(void);
int scg_abi_version(void);
/* sizeof(scg_config) / sizeof(scg_step_out) as compiled into the library, so that FFI bindings (ctypes
 * Structures in safe_control_gym_amd/_lib.py) can verify their struct layout at load time. */
size_t scg_sizeof_config(void);
size_t scg_sizeof_step_out(void);

/* Config specialisation.  The hot path is latency-bound at the headline size (one wave per SIMD), so the
 * fastest kernels are the ones compiled for ONE task config: scg_spec_source() returns the text of a small
 * header (every hot parameter as an exact hexadecimal literal + the FNV-1a hash of that text); compiling
 * the library sources with `hipcc -DSCG_SPEC -include <that header>` yields a library with the same ABI
 * whose kernels have the config baked in.  scg_create() of such a library refuses any other config.
 * `buf` may be NULL to query the length.  Instance fields (num_envs, env_id_offset, seed) are not part of
 * the hash; dtype is. */
int scg_spec_source(const scg_config* cfg, char* buf, size_t capacity, size_t* length, uint64_t* hash);
uint64_t scg_spec_hash(void);    /* 0 = generic library, else the hash this library was specialised for */
uint64_t scg_source_hash(void);  /* digest of the kernel sources the library was compiled from (build systems compare it
                                    with the tree so that a stale library is rebuilt instead of loaded) */

#ifdef __cplusplus
}
#endif
#endif /* SCG_HIP_H */

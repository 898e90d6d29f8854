// scg_params.h — device-side, dtype-typed mirror of scg_config + workspace layout.
//
// The host converts the double-precision scg_config (include/scg_hip.h) into DevParams<T> once at
// scg_create; kernels read it through a `const DevParams<T>* __restrict__` (uniform address -> scalar
// loads into SGPRs, no per-lane traffic).
#pragma once
#include <stdint.h>

#include "../../include/scg_hip.h"

namespace scg {

template <typename T>
struct DevDist {
    int32_t kind, dim, step_offset, max_step;
    int32_t offset_slot;        // row of the per-env random-offset table, -1 when the offset is fixed
    int32_t pad;
    T duration, half_duration, decay_rate, two_pi_freq;
    T a[SCG_MAX_STATE], b[SCG_MAX_STATE], mask[SCG_MAX_STATE];
};

template <typename T>
struct DevRand {
    int32_t kind, n_choice;
    T p0, p1;
    T choices[SCG_MAX_CHOICE];
};

template <typename T>
struct DevRow {
    int32_t kind, var, index, strict;
    int32_t is_box, pad;        // evaluated through the BoxRow table (hot path) instead of the generic loop
    T sign, b, round_scale, inv_round_scale;
    T coef[SCG_MAX_STATE];
};

template <typename T>
struct DevParams {
    int32_t system, num_envs, env_id_offset, integrator;
    uint32_t key0, key1;
    int32_t substeps, ctrl_steps;
    T pyb_dt, ctrl_dt;
    int32_t task, cost, obs_goal_horizon, goal_rows;
    int32_t rew_exponential, done_on_oob, done_on_violation, use_penalty;
    int32_t obs_wrap_angle, normalized_action, info_goal_reached, goal_in_lds;
    int32_t nx, nu, nobs, ns, np;       // state_dim, action_dim, obs_dim, raw state arrays, per-env params
    int32_t per_env_params;             // randomized_inertial_prop: read params from the workspace
    int32_t randomized_init, n_offset_slots;
    int32_t auto_reset, pad1;
    T goal_tolerance, constraint_penalty;
    T rew_state_weight[SCG_MAX_STATE], rew_act_weight[SCG_MAX_ACTION];
    T q_diag[SCG_MAX_STATE], r_diag[SCG_MAX_ACTION];
    T mse_weight[SCG_MAX_STATE], u_goal[SCG_MAX_ACTION];
    T state_low[SCG_MAX_STATE], state_high[SCG_MAX_STATE];
    T x_threshold, theta_threshold;
    T act_scale, hover_thrust;
    T act_low[SCG_MAX_ACTION], act_high[SCG_MAX_ACTION];
    T kf, km, pwm2rpm_scale, pwm2rpm_const, pwm_min, pwm_max;
    T gravity, arm, vmax, pole_box_width;
    T base_param[SCG_MAX_PARAM];
    DevRand<T> param_rand[SCG_MAX_PARAM];
    T init_state[SCG_MAX_STATE];
    DevRand<T> init_rand[SCG_MAX_STATE];
    int32_t n_dist[3];
    int32_t adversary_channel;
    DevDist<T> dist[3][SCG_MAX_DISTURB];
    int32_t n_con_rows, n_state_con_rows;
    int32_t n_box_rows, n_generic_rows;
    DevRow<T> con[SCG_MAX_CON_ROWS];
    T quad_P[SCG_MAX_QUAD_CON][SCG_MAX_STATE * SCG_MAX_STATE];
    // device pointers
    const T* x_goal;            // [goal_rows][nx]
    T* state;                   // [ns][N]
    T* param;                   // [np][N]
    int32_t* step;              // [N] ctrl_step_counter
    uint32_t* episode;          // [N]
    int32_t* dist_offset;       // [n_offset_slots][N]
    uint8_t* oob_attr;          // [N] persistent `self.out_of_bounds` attribute (stale-on-goal quirk)
};

// Box constraint row (bounded / default / abs_bound), grouped by constrained variable.
//   packed = row | state_pos << 8 | flags << 16 | slot << 24;  flags: bit0 strict, bit1 abs, bit2 negative
//   sign; slot: state k -> k, input j -> SCG_MAX_STATE + j (rows are sorted by slot, state rows first)
template <typename T>
struct BoxRow {
    int32_t packed;
    T b;
};

template <typename T>
struct HotRand {
    int32_t kind;
    T p0, p1;
};

// One disturbance of a channel list, hot copy (DevDist without the fields only the host needs).  Lives at the END of
// CfgParams: kernels without disturbances (DIST = false) never stage it into LDS, specialised builds read it as
// compile-time constants (kind / mask / magnitude branches fold, loops over the list and the dimensions unroll).
template <typename T>
struct HotDist {
    int32_t kind, step_offset, max_step, offset_slot;       // offset_slot: row of the per-env offset table, -1 = fixed offset
    T half_duration, decay_rate, two_pi_freq;
    T a[SCG_MAX_STATE], b[SCG_MAX_STATE], mask[SCG_MAX_STATE];
};

// ---------------------------------------------------------------------------------------------------
// Hot-path parameters.
//
// CfgParams<T>: every NUMBER the control step needs (no pointers), including the box-constraint table.
//   * generic library: one copy in device memory; each workgroup stages it into LDS with one 16-byte load
//     per thread (all in flight together with the per-env state loads) and reads fields from LDS;
//   * config-specialised builds (SCG_SPEC, see scg_spec.h): a `static constexpr CfgParams<T>` — every
//     field is a compile-time constant, branches on config flags vanish, loops fully unroll and there is
//     no parameter traffic at all.
//   Why: with 65 536 envs every SIMD holds ONE wave, so each dependent memory round trip (scalar or LDS
//   parameter fetch, ~100-600 cycles) is exposed latency: rocprofv3 PMC on the generic kernel shows 62 %
//   of wave cycles in s_waitcnt (profiles/r01_pmc_generic.md).
// InstParams<T>: pointers and per-instance integers, passed by value as a kernel argument.
// ---------------------------------------------------------------------------------------------------
#define SCG_CFG_INT_FIELDS(X)                                                                          \
    X(substeps) X(ctrl_steps) X(task) X(cost) X(obs_goal_horizon) X(goal_rows) X(nobs) X(nx)           \
    X(rew_exponential) X(done_on_oob) X(done_on_violation) X(use_penalty) X(obs_wrap_angle)            \
    X(normalized_action) X(info_goal_reached) X(goal_in_lds) X(per_env_params) X(randomized_init)      \
    X(auto_reset) X(adversary_channel) X(n_con_rows) X(n_state_con_rows) X(n_generic_rows)             \
    X(n_box_rows) X(n_box_state_rows) X(integrator) X(init_compact) X(param_compact)
#define SCG_CFG_INT_ARRAYS(X) X(n_dist, 3)
#define SCG_CFG_T_FIELDS(X)                                                                            \
    X(box_round) X(box_inv_round) X(pyb_dt) X(ctrl_dt) X(goal_tolerance) X(constraint_penalty) X(x_threshold)     \
    X(theta_threshold) X(act_scale) X(hover_thrust) X(kf) X(km) X(pwm2rpm_scale) X(pwm2rpm_const)      \
    X(pwm_min) X(pwm_max) X(gravity) X(arm) X(vmax) X(pole_box_width)
#define SCG_CFG_T_ARRAYS(X)                                                                            \
    X(rew_state_weight, SCG_MAX_STATE) X(rew_act_weight, SCG_MAX_ACTION) X(q_diag, SCG_MAX_STATE)       \
    X(r_diag, SCG_MAX_ACTION) X(mse_weight, SCG_MAX_STATE) X(u_goal, SCG_MAX_ACTION)                    \
    X(state_low, SCG_MAX_STATE) X(state_high, SCG_MAX_STATE) X(act_low, SCG_MAX_ACTION)                 \
    X(act_high, SCG_MAX_ACTION) X(base_param, SCG_MAX_PARAM) X(init_state, SCG_MAX_STATE)

template <typename T>
struct CfgParams {
#define SCG_X(f) int32_t f;
    SCG_CFG_INT_FIELDS(SCG_X)
#undef SCG_X
#define SCG_X(f, n) int32_t f[n];
    SCG_CFG_INT_ARRAYS(SCG_X)
#undef SCG_X
#define SCG_X(f) T f;
    SCG_CFG_T_FIELDS(SCG_X)
#undef SCG_X
#define SCG_X(f, n) T f[n];
    SCG_CFG_T_ARRAYS(SCG_X)
#undef SCG_X
    HotRand<T> param_rand[SCG_MAX_PARAM];
    HotRand<T> init_rand[SCG_MAX_STATE];
    BoxRow<T> box[SCG_MAX_CON_ROWS];     // sorted by variable slot, state rows first
    HotDist<T> dist[3][SCG_MAX_DISTURB]; // must stay the last member (see HotDist)
};

constexpr uint32_t SCG_NO_OFF = 0xffffffffu;

template <typename T>
struct InstParams {
    const DevParams<T>* cold;   // disturbance tables, generic constraint rows, choice lists
    const T* x_goal;            // [goal_rows][nx] in device memory
    // Per-env simulator arrays all live in the caller's workspace: ONE buffer resource (`ws`) and a 32-bit byte
    // offset per array (SCG_NO_OFF = absent), see Slot in scg_env_core.h.
    char* ws;
    uint32_t state_off, param_off, step_off, episode_off, oob_off, dist_off;
    int32_t num_envs, env_id_offset;
    int32_t env_first, env_end; // env range [env_first, env_end) this launch advances (sub-shard launches; whole batch: 0, N)
    uint32_t key0, key1;
    int32_t goal_lds16;         // generic build: number of 16-byte chunks of x_goal staged into LDS (0 = read from global)
    int32_t obs_ext_rows;       // obs_dim / state_dim - 1 (needed before the parameter block is staged)
};

// What the env code sees.
template <typename T>
struct PV {
    const CfgParams<T>& c;
    const InstParams<T>& i;
};

// Number of raw state arrays / per-env params / dims per system.
inline int sys_ns(int system) { return system == SCG_CARTPOLE ? 4 : system == SCG_QUAD_1D ? 2 : system == SCG_QUAD_2D ? 6 : 13; }
inline int sys_nx(int system) { return system == SCG_CARTPOLE ? 4 : system == SCG_QUAD_1D ? 2 : system == SCG_QUAD_2D ? 6 : 12; }
inline int sys_nu(int system) { return system == SCG_CARTPOLE ? 1 : system == SCG_QUAD_1D ? 1 : system == SCG_QUAD_2D ? 2 : 4; }
inline int sys_np(int system) { return system == SCG_CARTPOLE ? 3 : 4; }

}  // namespace scg

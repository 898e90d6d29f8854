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
    // box rows (SPARSE / ABS) regrouped by constrained variable so the kernel indexes registers statically:
    // variable slot v (0..NX-1 state, NX..NX+NU-1 input) owns entries [bv_first[v], bv_first[v+1])
    int32_t n_box_rows, n_generic_rows;
    int32_t bv_first[SCG_MAX_STATE + SCG_MAX_ACTION + 1];
    int32_t bv_row[SCG_MAX_CON_ROWS];        // output row in the stacked constraint vector
    int32_t bv_state_pos[SCG_MAX_CON_ROWS];  // output row among the state-only rows (reset-time evaluation)
    int32_t bv_flags[SCG_MAX_CON_ROWS];      // bit0 strict, bit1 abs
    T bv_sign[SCG_MAX_CON_ROWS], bv_b[SCG_MAX_CON_ROWS], bv_round[SCG_MAX_CON_ROWS], bv_inv_round[SCG_MAX_CON_ROWS];
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

// Number of raw state arrays / per-env params / dims per system.
inline int sys_ns(int system) { return system == SCG_CARTPOLE ? 4 : system == SCG_QUAD_1D ? 2 : system == SCG_QUAD_2D ? 6 : 13; }
inline int sys_nx(int system) { return system == SCG_CARTPOLE ? 4 : system == SCG_QUAD_1D ? 2 : system == SCG_QUAD_2D ? 6 : 12; }
inline int sys_nu(int system) { return system == SCG_CARTPOLE ? 1 : system == SCG_QUAD_1D ? 1 : system == SCG_QUAD_2D ? 2 : 4; }
inline int sys_np(int system) { return system == SCG_CARTPOLE ? 3 : 4; }

}  // namespace scg

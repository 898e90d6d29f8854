// scg_kernels.hip — the C ABI of libscg_hip.so (include/scg_hip.h) + kernel dispatch.
//
// Built two ways from the same sources:
//   hipcc ... scg_kernels.hip                              -> libscg_hip.so (any config, parameters staged in LDS)
//   hipcc ... -DSCG_SPEC -include <generated>.h scg_kernels.hip
//                                                          -> libscg_spec_<hash>.so: one task config baked in as
//                                                             compile-time constants (see emit_spec_source below)
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/scg_hip.h"
#include "scg_once.h"
#include "scg_params.h"
#include "scg_rng.h"
#ifdef SCG_SPEC
#include "scg_spec.h"
#endif
#include "scg_env_kernels.h"
#include "scg_gae_kernels.h"

using namespace scg;

// ------------------------------------------------------------------ error plumbing
static thread_local std::string g_last_error;
static int fail(int code, const std::string& msg) { g_last_error = msg; return code; }
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(SCG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));           \
    } while (0)

extern "C" const char* scg_last_error(void) { return g_last_error.c_str(); }
extern "C" int scg_abi_version(void) { return SCG_ABI_VERSION; }
#ifdef SCG_EXP_TIMELINE
extern "C" int scg_exp_timeline(unsigned long long* out, size_t n) {   // timing-only builds (tools/timeline.py)
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(scg::scg_timeline), n * sizeof(unsigned long long));
}
#endif
extern "C" size_t scg_sizeof_config(void) { return sizeof(scg_config); }
extern "C" size_t scg_sizeof_step_out(void) { return sizeof(scg_step_out); }

// ------------------------------------------------------------------ host side
struct scg_env {
    scg_config cfg;
    int device;
    int dtype;
    int nx, nu, nobs, ns, np;
    int n_offset_slots;
    size_t lds_bytes;
    bool goal_in_lds;
    int goal_lds16;
    void* d_params;          // DevParams<T> (cold block) on device
    void* d_cfg;             // CfgParams<T> on device (generic build: staged into LDS by every workgroup)
    void* d_goal;            // X_GOAL table on device
    // workspace partition (device pointers)
    void* d_workspace;
    void* d_state;
    void* d_param;
    int32_t* d_step;
    uint32_t* d_episode;
    int32_t* d_dist_offset;
    uint8_t* d_oob;
    bool has_reset;
    bool has_dist;
    // step-launch geometry of the specialised builds by shard size (scg_set_step_launch / scg_set_step_wsback): >= wide_min: 256-thread
    // workgroups; wsback_min .. wsback_max: workspace arrays written back
    int wide_min, wsback_min, wsback_max;
};

// Step-launch geometry by shard size, measured on MI355X (profiles/r05_step_kernel_ab.md, profiles/r06_split_wsback_ab.txt); the environment variables of the same names
// override the built-in thresholds at scg_create, scg_set_step_launch per handle.
#ifndef SCG_WSBACK_MIN_ENVS
#define SCG_WSBACK_MIN_ENVS 131072      // Quadrotor shards of 131 072 .. 524 288 envs: workspace arrays written back (step_wsback_kernel)
#endif
#ifndef SCG_WSBACK_MAX_ENVS
#define SCG_WSBACK_MAX_ENVS 524288
#endif
#ifndef SCG_WIDE_MIN_ENVS
#define SCG_WIDE_MIN_ENVS 8388608         // the largest shards: 256-thread workgroups (step_wide_kernel)
#endif
static int launch_default(const char* name, int built_in) {
    const char* s = std::getenv(name);
    return s && *s ? std::atoi(s) : built_in;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static size_t elem_size(int dtype) { return dtype == SCG_F64 ? 8 : 4; }

static int count_offset_slots(const scg_config* c) {
    int n = 0;
    for (int ch = 0; ch < 3; ++ch)
        for (int k = 0; k < c->n_dist[ch]; ++k) {
            const scg_disturbance& d = c->dist[ch][k];
            if ((d.kind == SCG_DIST_IMPULSE || d.kind == SCG_DIST_STEP) && d.step_offset < 0) ++n;
        }
    return n;
}

static int validate(const scg_config* c) {
    if (!c) return fail(SCG_ERR_INVALID, "cfg is NULL");
    if (c->abi_version != SCG_ABI_VERSION) return fail(SCG_ERR_INVALID, "scg_config.abi_version mismatch");
    if (c->system < SCG_CARTPOLE || c->system > SCG_QUAD_3D) return fail(SCG_ERR_INVALID, "unknown system");
    if (c->dtype != SCG_F32 && c->dtype != SCG_F64) return fail(SCG_ERR_INVALID, "unknown dtype");
    if (c->integrator != SCG_INT_PYB_EULER && c->integrator != SCG_INT_RK4) return fail(SCG_ERR_INVALID, "unknown integrator");
    if (c->integrator == SCG_INT_RK4 && (c->n_dist[SCG_CH_DYNAMICS] > 0 || c->adversary_channel == SCG_CH_DYNAMICS))
        return fail(SCG_ERR_INVALID, "SCG_INT_RK4 integrates the disturbance-free prior model: no dynamics disturbance / adversary");
    if (c->num_envs <= 0) return fail(SCG_ERR_INVALID, "num_envs must be positive");
    // kernels address each array as uniform base + 32-bit per-env byte offset (scg_env_core.h: Slot)
    if ((uint64_t)c->num_envs * 8u * (uint64_t)(SCG_MAX_STATE * (1 + (c->obs_goal_horizon > 0 ? c->obs_goal_horizon : 0))) >= (1ull << 32))
        return fail(SCG_ERR_INVALID, "num_envs too large: per-env byte offsets of the observation array must fit 32 bits");
    if (c->substeps <= 0 || c->ctrl_steps <= 0) return fail(SCG_ERR_INVALID, "substeps / ctrl_steps must be positive");
    if (c->obs_goal_horizon < 0 || c->obs_goal_horizon > SCG_MAX_GOAL_HORIZON) return fail(SCG_ERR_INVALID, "obs_goal_horizon out of range");
    if (c->goal_rows <= 0) return fail(SCG_ERR_INVALID, "goal_rows must be positive");
    if (c->task == SCG_TASK_STABILIZATION && c->goal_rows != 1) return fail(SCG_ERR_INVALID, "stabilisation needs goal_rows == 1");
    if (c->n_con_rows < 0 || c->n_con_rows > SCG_MAX_CON_ROWS) return fail(SCG_ERR_INVALID, "too many constraint rows");
    for (int ch = 0; ch < 3; ++ch)
        if (c->n_dist[ch] < 0 || c->n_dist[ch] > SCG_MAX_DISTURB) return fail(SCG_ERR_INVALID, "too many disturbances in a channel");
    if (c->adversary_channel > SCG_CH_DYNAMICS) return fail(SCG_ERR_INVALID, "adversary_channel must be action or dynamics");
    const int nx = sys_nx(c->system), nu = sys_nu(c->system);
    for (int r = 0; r < c->n_con_rows; ++r) {
        const scg_con_row& row = c->con[r];
        const int dim = row.var == 0 ? nx : nu;
        if (row.var != 0 && row.var != 1) return fail(SCG_ERR_INVALID, "constraint var must be state(0) or input(1)");
        if ((row.kind == SCG_ROW_SPARSE || row.kind == SCG_ROW_ABS) && (row.index < 0 || row.index >= dim))
            return fail(SCG_ERR_INVALID, "constraint row index out of range");
        if (row.kind == SCG_ROW_QUADRATIC && (row.index < 0 || row.index >= SCG_MAX_QUAD_CON))
            return fail(SCG_ERR_INVALID, "quadratic constraint index out of range");
    }
    return SCG_OK;
}

extern "C" int scg_dims(const scg_config* cfg, int32_t* state_dim, int32_t* action_dim, int32_t* obs_dim,
                        int32_t* n_state_arrays, int32_t* n_params) {
    if (int rc = validate(cfg)) return rc;
    const int nx = sys_nx(cfg->system);
    int mul = 1;
    // quadrotor.py:700-708 / cartpole.py:464-470
    if (cfg->cost == SCG_COST_RL_REWARD && cfg->obs_goal_horizon > 0)
        mul = cfg->task == SCG_TASK_TRAJ_TRACKING ? 1 + cfg->obs_goal_horizon : 2;
    if (state_dim) *state_dim = nx;
    if (action_dim) *action_dim = sys_nu(cfg->system);
    if (obs_dim) *obs_dim = nx * mul;
    if (n_state_arrays) *n_state_arrays = sys_ns(cfg->system);
    if (n_params) *n_params = sys_np(cfg->system);
    return SCG_OK;
}

struct Layout { size_t state, param, step, episode, offsets, oob, total; };
static Layout layout_of(const scg_config* c) {
    const size_t N = (size_t)c->num_envs, es = elem_size(c->dtype);
    Layout L{};
    size_t off = 0;
    L.state = off; off = align_up(off + sys_ns(c->system) * N * es, 256);
    L.param = off; off = align_up(off + sys_np(c->system) * N * es, 256);
    L.step = off; off = align_up(off + N * 4, 256);
    L.episode = off; off = align_up(off + N * 4, 256);
    L.offsets = off; off = align_up(off + (size_t)count_offset_slots(c) * N * 4, 256);
    L.oob = off; off = align_up(off + N, 256);
    L.total = off;
    return L;
}

extern "C" int scg_workspace_bytes(const scg_config* cfg, size_t* bytes) {
    if (int rc = validate(cfg)) return rc;
    if (!bytes) return fail(SCG_ERR_INVALID, "bytes is NULL");
    *bytes = layout_of(cfg).total;
    return SCG_OK;
}


// A constraint row is served by the hot box table when it is a +-1 sparse row or an abs row and shares the
// common rounding scale; every other row goes through the generic loop.
static bool row_is_box(const scg_con_row& row, double common_round) {
    const bool boxk = row.kind == SCG_ROW_SPARSE || row.kind == SCG_ROW_ABS;
    return boxk && row.round_scale == common_round && (row.kind == SCG_ROW_ABS || row.sign == 1.0 || row.sign == -1.0);
}
template <typename T>
static void fill_params(const scg_env* e, const scg_config& c, DevParams<T>& p) {
    std::memset(&p, 0, sizeof(p));
    p.system = c.system; p.num_envs = c.num_envs; p.env_id_offset = c.env_id_offset; p.integrator = c.integrator;
    p.key0 = (uint32_t)(c.seed & 0xffffffffu); p.key1 = (uint32_t)(c.seed >> 32);
    p.substeps = c.substeps; p.ctrl_steps = c.ctrl_steps; p.pyb_dt = (T)c.pyb_dt; p.ctrl_dt = (T)c.ctrl_dt;
    p.task = c.task; p.cost = c.cost; p.obs_goal_horizon = c.obs_goal_horizon; p.goal_rows = c.goal_rows;
    p.rew_exponential = c.rew_exponential; p.done_on_oob = c.done_on_out_of_bound;
    p.done_on_violation = c.done_on_violation; p.use_penalty = c.use_constraint_penalty;
    p.obs_wrap_angle = c.obs_wrap_angle; p.normalized_action = c.normalized_action;
    p.info_goal_reached = c.info_goal_reached;
    p.goal_in_lds = e->goal_in_lds;
    p.nx = e->nx; p.nu = e->nu; p.nobs = e->nobs; p.ns = e->ns; p.np = e->np;
    p.per_env_params = c.randomized_inertial_prop;
    p.randomized_init = c.randomized_init;
    p.auto_reset = c.auto_reset;
    p.n_offset_slots = e->n_offset_slots;
    p.goal_tolerance = (T)c.goal_tolerance; p.constraint_penalty = (T)c.constraint_penalty;
    for (int k = 0; k < SCG_MAX_STATE; ++k) {
        p.rew_state_weight[k] = (T)c.rew_state_weight[k]; p.q_diag[k] = (T)c.q_diag[k];
        p.mse_weight[k] = (T)c.mse_weight[k]; p.state_low[k] = (T)c.state_low[k]; p.state_high[k] = (T)c.state_high[k];
        p.init_state[k] = (T)c.init_state[k];
        p.init_rand[k].kind = c.init_rand[k].kind; p.init_rand[k].n_choice = c.init_rand[k].n_choice;
        p.init_rand[k].p0 = (T)c.init_rand[k].p0; p.init_rand[k].p1 = (T)c.init_rand[k].p1;
        for (int q = 0; q < SCG_MAX_CHOICE; ++q) p.init_rand[k].choices[q] = (T)c.init_rand[k].choices[q];
    }
    for (int k = 0; k < SCG_MAX_ACTION; ++k) {
        p.rew_act_weight[k] = (T)c.rew_act_weight[k]; p.r_diag[k] = (T)c.r_diag[k]; p.u_goal[k] = (T)c.u_goal[k];
        p.act_low[k] = (T)c.act_low[k]; p.act_high[k] = (T)c.act_high[k];
    }
    p.x_threshold = (T)c.x_threshold; p.theta_threshold = (T)c.theta_threshold;
    p.act_scale = (T)c.act_scale; p.hover_thrust = (T)c.hover_thrust;
    p.kf = (T)c.kf; p.km = (T)c.km; p.pwm2rpm_scale = (T)c.pwm2rpm_scale; p.pwm2rpm_const = (T)c.pwm2rpm_const;
    p.pwm_min = (T)c.pwm_min; p.pwm_max = (T)c.pwm_max;
    p.gravity = (T)c.gravity; p.arm = (T)c.arm; p.vmax = (T)c.max_coordinate_velocity; p.pole_box_width = (T)c.pole_box_width;
    for (int k = 0; k < SCG_MAX_PARAM; ++k) {
        p.base_param[k] = (T)c.base_param[k];
        p.param_rand[k].kind = c.param_rand[k].kind; p.param_rand[k].n_choice = c.param_rand[k].n_choice;
        p.param_rand[k].p0 = (T)c.param_rand[k].p0; p.param_rand[k].p1 = (T)c.param_rand[k].p1;
        for (int q = 0; q < SCG_MAX_CHOICE; ++q) p.param_rand[k].choices[q] = (T)c.param_rand[k].choices[q];
    }
    int slot = 0;
    for (int ch = 0; ch < 3; ++ch) {
        p.n_dist[ch] = c.n_dist[ch];
        for (int k = 0; k < c.n_dist[ch]; ++k) {
            const scg_disturbance& s = c.dist[ch][k];
            DevDist<T>& d = p.dist[ch][k];
            d.kind = s.kind; d.dim = s.dim; d.step_offset = s.step_offset; d.max_step = s.max_step;
            d.offset_slot = -1;
            if ((s.kind == SCG_DIST_IMPULSE || s.kind == SCG_DIST_STEP) && s.step_offset < 0) d.offset_slot = slot++;
            d.duration = (T)s.duration; d.half_duration = (T)(s.duration / 2.0); d.decay_rate = (T)s.decay_rate;
            d.two_pi_freq = (T)(2.0 * M_PI * s.frequency);
            for (int j = 0; j < SCG_MAX_STATE; ++j) { d.a[j] = (T)s.a[j]; d.b[j] = (T)s.b[j]; d.mask[j] = (T)s.mask[j]; }
        }
    }
    p.adversary_channel = c.adversary_channel;
    p.n_con_rows = c.n_con_rows; p.n_state_con_rows = c.n_state_con_rows;
    for (int r = 0; r < c.n_con_rows; ++r) {
        const scg_con_row& s = c.con[r];
        DevRow<T>& d = p.con[r];
        d.kind = s.kind; d.var = s.var; d.index = s.index; d.strict = s.strict;
        d.is_box = 0;
        d.sign = (T)s.sign; d.b = (T)s.b;
        // rounding to `decimals` places is only meaningful in double precision (np.round(., 8) on float64)
        const bool round = s.round_scale > 0 && sizeof(T) == 8;
        d.round_scale = round ? (T)s.round_scale : (T)0;
        d.inv_round_scale = round ? (T)(1.0 / s.round_scale) : (T)0;
        for (int j = 0; j < SCG_MAX_STATE; ++j) d.coef[j] = (T)s.coef[j];
    }
    for (int q = 0; q < SCG_MAX_QUAD_CON; ++q)
        for (int j = 0; j < SCG_MAX_STATE * SCG_MAX_STATE; ++j) p.quad_P[q][j] = (T)c.quad_P[q][j];
    p.x_goal = (const T*)e->d_goal;
    p.state = (T*)e->d_state; p.param = (T*)e->d_param; p.step = e->d_step; p.episode = e->d_episode;
    p.dist_offset = e->d_dist_offset; p.oob_attr = e->d_oob;
}


// ---- hot configuration block -------------------------------------------------------------------
// scg_config -> CfgParams<double>: the single place where the hot parameters are derived.
static void build_cfg(const scg_config& c, CfgParams<double>& h) {
    std::memset(&h, 0, sizeof(h));
    int32_t nx, nu, nobs, ns, np;
    scg_dims(&c, &nx, &nu, &nobs, &ns, &np);
    h.integrator = c.integrator;
    h.substeps = c.substeps; h.ctrl_steps = c.ctrl_steps; h.task = c.task; h.cost = c.cost;
    h.obs_goal_horizon = c.obs_goal_horizon; h.goal_rows = c.goal_rows; h.nobs = nobs; h.nx = nx;
    h.rew_exponential = c.rew_exponential; h.done_on_oob = c.done_on_out_of_bound;
    h.done_on_violation = c.done_on_violation; h.use_penalty = c.use_constraint_penalty;
    h.obs_wrap_angle = c.obs_wrap_angle; h.normalized_action = c.normalized_action;
    h.info_goal_reached = c.info_goal_reached;
    h.goal_in_lds = (size_t)c.goal_rows * nx * elem_size(c.dtype) <= LDS_GOAL_LIMIT;
    h.per_env_params = c.randomized_inertial_prop; h.randomized_init = c.randomized_init;
    // compact Philox word layout (scg_rng.h) for a group without two-word (normal) draws
    h.init_compact = 1; h.param_compact = 1;
    for (int k = 0; k < SCG_MAX_STATE; ++k) if (c.init_rand[k].kind == SCG_RAND_NORMAL) h.init_compact = 0;
    for (int k = 0; k < SCG_MAX_PARAM; ++k) if (c.param_rand[k].kind == SCG_RAND_NORMAL) h.param_compact = 0;
    h.auto_reset = c.auto_reset; h.adversary_channel = c.adversary_channel;
    h.n_con_rows = c.n_con_rows; h.n_state_con_rows = c.n_state_con_rows;
    for (int k = 0; k < 3; ++k) h.n_dist[k] = c.n_dist[k];
    // disturbance lists (hot copy; offset slots numbered in channel/list order like count_offset_slots)
    int oslot = 0;
    for (int ch = 0; ch < 3; ++ch)
        for (int k = 0; k < c.n_dist[ch]; ++k) {
            const scg_disturbance& sd = c.dist[ch][k];
            HotDist<double>& d = h.dist[ch][k];
            d.kind = sd.kind; d.step_offset = sd.step_offset; d.max_step = sd.max_step;
            d.offset_slot = -1;
            if ((sd.kind == SCG_DIST_IMPULSE || sd.kind == SCG_DIST_STEP) && sd.step_offset < 0) d.offset_slot = oslot++;
            d.half_duration = sd.duration / 2.0; d.decay_rate = sd.decay_rate; d.two_pi_freq = 2.0 * M_PI * sd.frequency;
            for (int j = 0; j < SCG_MAX_STATE; ++j) { d.a[j] = sd.a[j]; d.b[j] = sd.b[j]; d.mask[j] = sd.mask[j]; }
        }
    h.pyb_dt = c.pyb_dt; h.ctrl_dt = c.ctrl_dt; h.goal_tolerance = c.goal_tolerance; h.constraint_penalty = c.constraint_penalty;
    h.x_threshold = c.x_threshold; h.theta_threshold = c.theta_threshold; h.act_scale = c.act_scale;
    h.hover_thrust = c.hover_thrust; h.kf = c.kf; h.km = c.km; h.pwm2rpm_scale = c.pwm2rpm_scale;
    h.pwm2rpm_const = c.pwm2rpm_const; h.pwm_min = c.pwm_min; h.pwm_max = c.pwm_max; h.gravity = c.gravity;
    h.arm = c.arm; h.vmax = c.max_coordinate_velocity; h.pole_box_width = c.pole_box_width;
    for (int k = 0; k < SCG_MAX_STATE; ++k) {
        h.rew_state_weight[k] = c.rew_state_weight[k]; h.q_diag[k] = c.q_diag[k]; h.mse_weight[k] = c.mse_weight[k];
        h.state_low[k] = c.state_low[k]; h.state_high[k] = c.state_high[k]; h.init_state[k] = c.init_state[k];
        h.init_rand[k].kind = c.init_rand[k].kind; h.init_rand[k].p0 = c.init_rand[k].p0; h.init_rand[k].p1 = c.init_rand[k].p1;
    }
    for (int k = 0; k < SCG_MAX_ACTION; ++k) {
        h.rew_act_weight[k] = c.rew_act_weight[k]; h.r_diag[k] = c.r_diag[k]; h.u_goal[k] = c.u_goal[k];
        h.act_low[k] = c.act_low[k]; h.act_high[k] = c.act_high[k];
    }
    for (int k = 0; k < SCG_MAX_PARAM; ++k) {
        h.base_param[k] = c.base_param[k];
        h.param_rand[k].kind = c.param_rand[k].kind; h.param_rand[k].p0 = c.param_rand[k].p0; h.param_rand[k].p1 = c.param_rand[k].p1;
    }
    // box rows (SPARSE with sign +-1, ABS) sharing one rounding scale, sorted by variable slot (state first)
    double common = -1.0;
    for (int r = 0; r < c.n_con_rows; ++r)
        if (c.con[r].kind == SCG_ROW_SPARSE || c.con[r].kind == SCG_ROW_ABS) { common = c.con[r].round_scale; break; }
    std::vector<int> state_pos(c.n_con_rows > 0 ? c.n_con_rows : 1, 0);
    int sp = 0, n = 0;
    for (int r = 0; r < c.n_con_rows; ++r) { state_pos[r] = sp; if (c.con[r].var == 0) ++sp; }
    for (int slot = 0; slot < SCG_MAX_STATE + SCG_MAX_ACTION; ++slot) {
        if (slot == SCG_MAX_STATE) h.n_box_state_rows = n;
        const int var = slot < SCG_MAX_STATE ? 0 : 1, index = slot < SCG_MAX_STATE ? slot : slot - SCG_MAX_STATE;
        for (int r = 0; r < c.n_con_rows; ++r) {
            if (!row_is_box(c.con[r], common) || c.con[r].var != var || c.con[r].index != index) continue;
            const scg_con_row& row = c.con[r];
            const int flags = (row.strict ? 1 : 0) | (row.kind == SCG_ROW_ABS ? 2 : 0) | (row.sign < 0 ? 4 : 0);
            h.box[n].packed = r | (state_pos[r] << 8) | (flags << 16) | (slot << 24);
            h.box[n].b = row.b;
            ++n;
        }
    }
    h.n_box_rows = n;
    h.n_generic_rows = c.n_con_rows - n;
    h.box_round = common > 0 ? common : 0.0;
    h.box_inv_round = common > 0 ? 1.0 / common : 0.0;
}

template <typename T>
static void convert_cfg(const CfgParams<double>& d, CfgParams<T>& o) {
    std::memset(&o, 0, sizeof(o));
#define SCG_X(f) o.f = d.f;
    SCG_CFG_INT_FIELDS(SCG_X)
#undef SCG_X
#define SCG_X(f, n) for (int k = 0; k < n; ++k) o.f[k] = d.f[k];
    SCG_CFG_INT_ARRAYS(SCG_X)
#undef SCG_X
#define SCG_X(f) o.f = (T)d.f;
    SCG_CFG_T_FIELDS(SCG_X)
#undef SCG_X
#define SCG_X(f, n) for (int k = 0; k < n; ++k) o.f[k] = (T)d.f[k];
    SCG_CFG_T_ARRAYS(SCG_X)
#undef SCG_X
    for (int k = 0; k < SCG_MAX_PARAM; ++k) { o.param_rand[k].kind = d.param_rand[k].kind; o.param_rand[k].p0 = (T)d.param_rand[k].p0; o.param_rand[k].p1 = (T)d.param_rand[k].p1; }
    for (int k = 0; k < SCG_MAX_STATE; ++k) { o.init_rand[k].kind = d.init_rand[k].kind; o.init_rand[k].p0 = (T)d.init_rand[k].p0; o.init_rand[k].p1 = (T)d.init_rand[k].p1; }
    for (int k = 0; k < SCG_MAX_CON_ROWS; ++k) { o.box[k].packed = d.box[k].packed; o.box[k].b = (T)d.box[k].b; }
    for (int ch = 0; ch < 3; ++ch)
        for (int k = 0; k < SCG_MAX_DISTURB; ++k) {
            const HotDist<double>& a = d.dist[ch][k];
            HotDist<T>& b = o.dist[ch][k];
            b.kind = a.kind; b.step_offset = a.step_offset; b.max_step = a.max_step; b.offset_slot = a.offset_slot;
            b.half_duration = (T)a.half_duration; b.decay_rate = (T)a.decay_rate; b.two_pi_freq = (T)a.two_pi_freq;
            for (int j = 0; j < SCG_MAX_STATE; ++j) { b.a[j] = (T)a.a[j]; b.b[j] = (T)a.b[j]; b.mask[j] = (T)a.mask[j]; }
        }
    // np.round(., decimals) is only meaningful in double precision
    if (sizeof(T) != 8) { o.box_round = (T)0; o.box_inv_round = (T)0; }
}

// ---- config specialisation: C++ source of a constexpr CfgParams<T> for this config ------------------
static uint64_t fnv1a(const std::string& s) {
    uint64_t h = 1469598103934665603ULL;
    for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ULL; }
    return h;
}

static std::string spec_body(const scg_config& c) {
    CfgParams<double> h;
    build_cfg(c, h);
    std::string s;
    char buf[256];
    auto addi = [&](const char* name, long v) { std::snprintf(buf, sizeof buf, "    c.%s = %ld;\n", name, v); s += buf; };
    auto addt = [&](const char* name, double v) { std::snprintf(buf, sizeof buf, "    c.%s = (T)%a;\n", name, v); s += buf; };
#define SCG_X(f) addi(#f, (long)h.f);
    SCG_CFG_INT_FIELDS(SCG_X)
#undef SCG_X
#define SCG_X(f, n) for (int k = 0; k < n; ++k) { std::snprintf(buf, sizeof buf, "    c.%s[%d] = %ld;\n", #f, k, (long)h.f[k]); s += buf; }
    SCG_CFG_INT_ARRAYS(SCG_X)
#undef SCG_X
#define SCG_X(f) addt(#f, h.f);
    SCG_CFG_T_FIELDS(SCG_X)
#undef SCG_X
#define SCG_X(f, n) for (int k = 0; k < n; ++k) { std::snprintf(buf, sizeof buf, "    c.%s[%d] = (T)%a;\n", #f, k, h.f[k]); s += buf; }
    SCG_CFG_T_ARRAYS(SCG_X)
#undef SCG_X
    for (int k = 0; k < SCG_MAX_PARAM; ++k) {
        std::snprintf(buf, sizeof buf, "    c.param_rand[%d].kind = %d; c.param_rand[%d].p0 = (T)%a; c.param_rand[%d].p1 = (T)%a;\n",
                      k, h.param_rand[k].kind, k, h.param_rand[k].p0, k, h.param_rand[k].p1); s += buf;
    }
    for (int k = 0; k < SCG_MAX_STATE; ++k) {
        std::snprintf(buf, sizeof buf, "    c.init_rand[%d].kind = %d; c.init_rand[%d].p0 = (T)%a; c.init_rand[%d].p1 = (T)%a;\n",
                      k, h.init_rand[k].kind, k, h.init_rand[k].p0, k, h.init_rand[k].p1); s += buf;
    }
    for (int k = 0; k < h.n_box_rows; ++k) {
        std::snprintf(buf, sizeof buf, "    c.box[%d].packed = %d; c.box[%d].b = (T)%a;\n", k, h.box[k].packed, k, h.box[k].b); s += buf;
    }
    for (int ch = 0; ch < 3; ++ch)
        for (int k = 0; k < h.n_dist[ch]; ++k) {
            const HotDist<double>& d = h.dist[ch][k];
            std::snprintf(buf, sizeof buf, "    c.dist[%d][%d].kind = %d; c.dist[%d][%d].step_offset = %d; c.dist[%d][%d].max_step = %d; c.dist[%d][%d].offset_slot = %d;\n",
                          ch, k, d.kind, ch, k, d.step_offset, ch, k, d.max_step, ch, k, d.offset_slot); s += buf;
            std::snprintf(buf, sizeof buf, "    c.dist[%d][%d].half_duration = (T)%a; c.dist[%d][%d].decay_rate = (T)%a; c.dist[%d][%d].two_pi_freq = (T)%a;\n",
                          ch, k, d.half_duration, ch, k, d.decay_rate, ch, k, d.two_pi_freq); s += buf;
            for (int j = 0; j < SCG_MAX_STATE; ++j) {
                std::snprintf(buf, sizeof buf, "    c.dist[%d][%d].a[%d] = (T)%a; c.dist[%d][%d].b[%d] = (T)%a; c.dist[%d][%d].mask[%d] = (T)%a;\n",
                              ch, k, j, d.a[j], ch, k, j, d.b[j], ch, k, j, d.mask[j]); s += buf;
            }
        }
    s += "    if (sizeof(T) != 8) { c.box_round = (T)0; c.box_inv_round = (T)0; }\n";
    std::snprintf(buf, sizeof buf, "// system %d dtype %d dist %d\n", c.system, c.dtype,
                  (int)(c.n_dist[0] > 0 || c.n_dist[1] > 0 || c.n_dist[2] > 0 || c.adversary_channel >= 0));
    s += buf;
    return s;
}

static std::string emit_spec_source(const scg_config& c, uint64_t* hash_out) {
    const std::string body = spec_body(c);
    const uint64_t hash = fnv1a(body);
    if (hash_out) *hash_out = hash;
    char buf[256];
    std::string s = "// Generated by scg_spec_source() (libscg_hip.so) — one task config as compile-time constants.\n";
    std::snprintf(buf, sizeof buf, "#define SCG_SPEC_HASH 0x%016llxULL\n#define SCG_SPEC_SYS %d\n#define SCG_SPEC_DTYPE %d\n#define SCG_SPEC_DIST %d\n",
                  (unsigned long long)hash, c.system, c.dtype,
                  (int)(c.n_dist[0] > 0 || c.n_dist[1] > 0 || c.n_dist[2] > 0 || c.adversary_channel >= 0));
    s += buf;
    s += "#define SCG_SPEC_FILL(c) \\\n";
    // body as a macro so that scg_spec.h can expand it inside a constexpr function template
    std::string m;
    for (char ch : body) { if (ch == '\n') m += " \\\n"; else m += ch; }
    s += m;
    s += "\n";
    return s;
}

extern "C" int scg_spec_source(const scg_config* cfg, char* buf, size_t capacity, size_t* length, uint64_t* hash) {
    if (int rc = validate(cfg)) return rc;
    uint64_t h = 0;
    const std::string src = emit_spec_source(*cfg, &h);
    if (length) *length = src.size() + 1;
    if (hash) *hash = h;
    if (buf) {
        if (capacity < src.size() + 1) return fail(SCG_ERR_INVALID, "buffer too small for the specialisation source");
        std::memcpy(buf, src.c_str(), src.size() + 1);
    }
    return SCG_OK;
}

// Digest of the kernel sources this library was compiled from (-DSCG_SRC_HASH, see _lib.py::source_hash); 0 = unknown.
#ifndef SCG_SRC_HASH
#define SCG_SRC_HASH 0ULL
#endif
extern "C" uint64_t scg_source_hash(void) { return SCG_SRC_HASH; }
// the same digest as text inside the file, so that a build script can read it without loading the library
#define SCG_STR2(x) #x
#define SCG_STR(x) SCG_STR2(x)
extern "C" const char* scg_source_hash_tag(void) { return "SCG_SRC_HASH:" SCG_STR(SCG_SRC_HASH); }

// 0 for the generic library, the baked-in config hash for a specialised one.
extern "C" uint64_t scg_spec_hash(void) {
#ifdef SCG_SPEC
    return SCG_SPEC_HASH;
#else
    return 0;
#endif
}

extern "C" int scg_destroy(scg_env* env);

template <typename T>
static int upload(scg_env* e, const double* h_x_goal) {
    const size_t ng = (size_t)e->cfg.goal_rows * e->nx;
    std::vector<T> tab(ng);
    for (size_t k = 0; k < ng; ++k) tab[k] = (T)h_x_goal[k];
    HIP_TRY(hipMalloc(&e->d_goal, align_up(ng * sizeof(T), 16) + 16));
    HIP_TRY(hipMemset(e->d_goal, 0, align_up(ng * sizeof(T), 16) + 16));
    HIP_TRY(hipMemcpy(e->d_goal, tab.data(), ng * sizeof(T), hipMemcpyHostToDevice));
    // hot block
    CfgParams<double> hd;
    build_cfg(e->cfg, hd);
    CfgParams<T>* hc = new (std::nothrow) CfgParams<T>;
    DevParams<T>* hp = new (std::nothrow) DevParams<T>;
    if (!hc || !hp) { delete hc; delete hp; return fail(SCG_ERR_NOMEM, "host allocation failed"); }
    convert_cfg<T>(hd, *hc);
    fill_params<T>(e, e->cfg, *hp);
    // rows served by the box table are skipped by the generic loop; other sparse rows become one-hot dense rows
    for (int k = 0; k < hd.n_box_rows; ++k) hp->con[hd.box[k].packed & 0xff].is_box = 1;
    for (int r = 0; r < e->cfg.n_con_rows; ++r) {
        DevRow<T>& d = hp->con[r];
        if (!d.is_box && d.kind == SCG_ROW_SPARSE) {
            for (int j = 0; j < SCG_MAX_STATE; ++j) d.coef[j] = (T)0;
            d.coef[d.index] = d.sign;
            d.kind = SCG_ROW_DENSE;
        }
    }
    hp->n_box_rows = hd.n_box_rows; hp->n_generic_rows = hd.n_generic_rows;
    hipError_t err = hipMalloc(&e->d_params, sizeof(DevParams<T>));
    if (err == hipSuccess) err = hipMemcpy(e->d_params, hp, sizeof(DevParams<T>), hipMemcpyHostToDevice);
    const size_t cfg_bytes = lds16(sizeof(CfgParams<T>));
    if (err == hipSuccess) err = hipMalloc(&e->d_cfg, cfg_bytes);
    if (err == hipSuccess) err = hipMemset(e->d_cfg, 0, cfg_bytes);
    if (err == hipSuccess) err = hipMemcpy(e->d_cfg, hc, sizeof(CfgParams<T>), hipMemcpyHostToDevice);
    delete hc;
    delete hp;
    if (err != hipSuccess) return fail(SCG_ERR_HIP, std::string("uploading parameters: ") + hipGetErrorString(err));
    return SCG_OK;
}

template <typename T>
static InstParams<T> inst_of(const scg_env* e) {
    InstParams<T> I;
    I.cold = (const DevParams<T>*)e->d_params; I.x_goal = (const T*)e->d_goal;
    I.ws = (char*)e->d_workspace;
    auto off = [&](const void* q) { return q ? (uint32_t)((const char*)q - (const char*)e->d_workspace) : SCG_NO_OFF; };
    I.state_off = off(e->d_state); I.param_off = off(e->d_param); I.step_off = off(e->d_step);
    I.episode_off = off(e->d_episode); I.oob_off = off(e->d_oob); I.dist_off = off(e->d_dist_offset);
    I.num_envs = e->cfg.num_envs; I.env_id_offset = e->cfg.env_id_offset;
    I.env_first = 0; I.env_end = e->cfg.num_envs;
    I.key0 = (uint32_t)(e->cfg.seed & 0xffffffffu); I.key1 = (uint32_t)(e->cfg.seed >> 32);
    I.goal_lds16 = e->goal_lds16; I.obs_ext_rows = e->nobs / e->nx - 1;
    return I;
}

extern "C" int scg_create(const scg_config* cfg, const double* h_x_goal, int device, void* d_workspace,
                          size_t workspace_bytes, scg_env** out) {
    if (int rc = validate(cfg)) return rc;
    if (!h_x_goal || !d_workspace || !out) return fail(SCG_ERR_INVALID, "NULL argument to scg_create");
#ifdef SCG_SPEC
    {
        uint64_t h = 0;
        (void)emit_spec_source(*cfg, &h);
        if (h != SCG_SPEC_HASH)
            return fail(SCG_ERR_INVALID, "this library is specialised for another task config (hash mismatch); "
                                         "use libscg_hip.so or rebuild the specialisation");
    }
#endif
    const Layout L = layout_of(cfg);
    if (workspace_bytes < L.total) return fail(SCG_ERR_INVALID, "workspace too small (see scg_workspace_bytes)");
    if (L.total >= 0xffff0000ull) return fail(SCG_ERR_INVALID, "num_envs too large: the workspace must stay below 4 GiB (32-bit array offsets)");
    if ((uintptr_t)d_workspace % 256 != 0) return fail(SCG_ERR_INVALID, "workspace must be 256-byte aligned");
    HIP_TRY(hipSetDevice(device));
    scg_env* e = new (std::nothrow) scg_env();
    if (!e) return fail(SCG_ERR_NOMEM, "host allocation failed");
    e->cfg = *cfg; e->device = device; e->dtype = cfg->dtype;
    int32_t nx, nu, nobs, ns, np;
    scg_dims(cfg, &nx, &nu, &nobs, &ns, &np);
    e->nx = nx; e->nu = nu; e->nobs = nobs; e->ns = ns; e->np = np;
    e->n_offset_slots = count_offset_slots(cfg);
    const size_t tab_bytes = (size_t)cfg->goal_rows * nx * elem_size(cfg->dtype);
    e->goal_in_lds = tab_bytes <= LDS_GOAL_LIMIT;
    e->goal_lds16 = e->goal_in_lds ? (int)(align_up(tab_bytes, 16) / 16) : 0;
    const size_t cfg_sz = cfg->dtype == SCG_F64 ? sizeof(CfgParams<double>) : sizeof(CfgParams<float>);
#ifdef SCG_SPEC
    e->lds_bytes = 0; e->goal_lds16 = 0;
#else
    e->lds_bytes = lds16(cfg_sz) + (size_t)e->goal_lds16 * 16;      // (kernels without disturbances stage less of it)
#endif
    unsigned char* w = (unsigned char*)d_workspace;
    e->d_workspace = d_workspace;
    e->d_state = w + L.state; e->d_param = w + L.param; e->d_step = (int32_t*)(w + L.step);
    e->d_episode = (uint32_t*)(w + L.episode); e->d_dist_offset = (int32_t*)(w + L.offsets); e->d_oob = w + L.oob;
    e->d_params = nullptr; e->d_goal = nullptr; e->d_cfg = nullptr; e->has_reset = false;
    e->wsback_min = launch_default("SCG_WSBACK_MIN_ENVS", SCG_WSBACK_MIN_ENVS);
    e->wsback_max = launch_default("SCG_WSBACK_MAX_ENVS", SCG_WSBACK_MAX_ENVS);
    e->wide_min = launch_default("SCG_WIDE_MIN_ENVS", SCG_WIDE_MIN_ENVS);
    e->has_dist = cfg->n_dist[0] > 0 || cfg->n_dist[1] > 0 || cfg->n_dist[2] > 0 || cfg->adversary_channel >= 0;
    hipError_t err = hipMemset(d_workspace, 0, L.total);
    if (err == hipSuccess) err = hipMemset(e->d_episode, 0xff, (size_t)cfg->num_envs * 4);   // first reset -> episode 0
    if (err != hipSuccess) { delete e; return fail(SCG_ERR_HIP, std::string("hipMemset: ") + hipGetErrorString(err)); }
    int rc = cfg->dtype == SCG_F64 ? upload<double>(e, h_x_goal) : upload<float>(e, h_x_goal);
    if (rc) { scg_destroy(e); return rc; }
    *out = e;
    return SCG_OK;
}

extern "C" int scg_destroy(scg_env* env) {
    if (!env) return SCG_OK;
    (void)hipSetDevice(env->device);
    if (env->d_goal) (void)hipFree(env->d_goal);
    if (env->d_params) (void)hipFree(env->d_params);
    if (env->d_cfg) (void)hipFree(env->d_cfg);
    delete env;
    return SCG_OK;
}


// The specialised build instantiates the kernels for its own dtype only.
#ifdef SCG_SPEC
#if SCG_SPEC_DTYPE == 1
#define SCG_BY_DTYPE(env, fn, ...) fn<double>(__VA_ARGS__)
#else
#define SCG_BY_DTYPE(env, fn, ...) fn<float>(__VA_ARGS__)
#endif
#else
#define SCG_BY_DTYPE(env, fn, ...) ((env)->dtype == SCG_F64 ? fn<double>(__VA_ARGS__) : fn<float>(__VA_ARGS__))
#endif
// scg_step_out -> kernel-side table.  When every bound array (with its extent) fits a 4 GiB window the kernel addresses
// them through one buffer resource (base + 32-bit offsets); otherwise through one resource per array.
template <typename T>
static void out_tabs(const scg_env* e, const scg_step_out* o, OutTabOne* one, OutTabPtr* each, bool* one_base) {
    one->base = nullptr;
    for (int k = 0; k < OUT_COUNT; ++k) { one->off[k] = SCG_NO_OFF; each->ptr[k] = nullptr; }
    *one_base = true;
    if (!o) return;
    const size_t N = (size_t)e->cfg.num_envs, sT = sizeof(T);
    void* const p[OUT_COUNT] = {o->d_obs, o->d_reward, o->d_done, o->d_flags, o->d_c_values, o->d_mse, o->d_terminal_obs,
                                o->d_state, o->d_noisy_action, o->d_ep_stats, o->d_fin_stats};
    const size_t ext[OUT_COUNT] = {N * e->nobs * sT, N * sT, N, N, (size_t)(e->cfg.n_con_rows > 0 ? e->cfg.n_con_rows : 1) * N * sT, N * sT,
                                   N * e->nobs * sT, (size_t)e->nx * N * sT, (size_t)e->nu * N * sT, 4 * N * sT, 4 * N * sT};
    uintptr_t lo = UINTPTR_MAX, hi = 0;
    for (int k = 0; k < OUT_COUNT; ++k) {
        each->ptr[k] = (char*)p[k];
        if (!p[k]) continue;
        lo = std::min(lo, (uintptr_t)p[k]);
        hi = std::max(hi, (uintptr_t)p[k] + ext[k]);
    }
    if (hi == 0) return;
    if (hi - lo >= 0xffff0000ull) { *one_base = false; return; }
    one->base = (char*)lo;
    for (int k = 0; k < OUT_COUNT; ++k)
        if (p[k]) one->off[k] = (uint32_t)((uintptr_t)p[k] - lo);
}

#define DISPATCH_SYS_D(env, T, CALL)                                           \
    switch ((env)->cfg.system) {                                               \
        case SCG_CARTPOLE: { constexpr int S = SCG_CARTPOLE; CALL; } break;    \
        case SCG_QUAD_1D: { constexpr int S = SCG_QUAD_1D; CALL; } break;      \
        case SCG_QUAD_2D: { constexpr int S = SCG_QUAD_2D; CALL; } break;      \
        default: { constexpr int S = SCG_QUAD_3D; CALL; } break;               \
    }
// DIST kernel variant only when a disturbance or an adversary is configured
#ifdef SCG_SPEC
// specialised build: exactly one (system, DIST) combination is instantiated
#define DISPATCH_SYS(env, T, CALL)                                             \
    { constexpr int S = SCG_SPEC_SYS; constexpr bool DD = SCG_SPEC_DIST != 0; CALL; }
#else
#define DISPATCH_SYS(env, T, CALL)                                             \
    if ((env)->has_dist) { constexpr bool DD = true; DISPATCH_SYS_D(env, T, CALL) } \
    else { constexpr bool DD = false; DISPATCH_SYS_D(env, T, CALL) }
#endif

template <typename T>
static int launch_reset(scg_env* env, const uint8_t* mask, const scg_step_out* out, hipStream_t st) {
    const int grid = (env->cfg.num_envs + BLOCK - 1) / BLOCK;
    bool one_base;
    OutTabOne O1;
    OutTabPtr O;
    out_tabs<T>(env, out, &O1, &O, &one_base);
    const CfgParams<T>* C = (const CfgParams<T>*)env->d_cfg;
    const InstParams<T> I = inst_of<T>(env);
    DISPATCH_SYS(env, T, (reset_kernel<S, T, DD><<<dim3(grid), dim3(BLOCK), env->lds_bytes, st>>>(C, I, mask, O)));
    HIP_TRY(hipGetLastError());
    return SCG_OK;
}

template <typename T>
static int launch_step(scg_env* env, int first, int count, const void* action, const void* adv, const scg_step_out* out, hipStream_t st) {
    const int grid = (count + BLOCK - 1) / BLOCK;
    bool one_base;
    OutTabOne O1;
    OutTabPtr O;
    out_tabs<T>(env, out, &O1, &O, &one_base);
    const CfgParams<T>* C = (const CfgParams<T>*)env->d_cfg;
    InstParams<T> I = inst_of<T>(env);
    I.env_first = first; I.env_end = first + count;
#ifdef SCG_SPEC
    if constexpr (SCG_SPEC_SYS != SCG_CARTPOLE) {        // (CartPole's 50-substep chain gains nothing from it: 13.3 -> 13.9 us at 262 144 envs)
        if (count >= env->wsback_min && count <= env->wsback_max && count < env->wide_min) {
            if (one_base) {
                DISPATCH_SYS(env, T, (step_wsback_kernel<S, T, DD, true><<<dim3(grid), dim3(BLOCK), env->lds_bytes, st>>>(C, I, (const T*)action, (const T*)adv, O1)));
            } else {
                DISPATCH_SYS(env, T, (step_wsback_kernel<S, T, DD, false><<<dim3(grid), dim3(BLOCK), env->lds_bytes, st>>>(C, I, (const T*)action, (const T*)adv, O)));
            }
            HIP_TRY(hipGetLastError());
            return SCG_OK;
        }
    }
    if (count >= env->wide_min) {
        const int grid_w = (count + WIDE_BLOCK - 1) / WIDE_BLOCK;
        if (one_base) {
            DISPATCH_SYS(env, T, (step_wide_kernel<S, T, DD, true><<<dim3(grid_w), dim3(WIDE_BLOCK), env->lds_bytes, st>>>(C, I, (const T*)action, (const T*)adv, O1)));
        } else {
            DISPATCH_SYS(env, T, (step_wide_kernel<S, T, DD, false><<<dim3(grid_w), dim3(WIDE_BLOCK), env->lds_bytes, st>>>(C, I, (const T*)action, (const T*)adv, O)));
        }
        HIP_TRY(hipGetLastError());
        return SCG_OK;
    }
#endif
    if (one_base) {
        DISPATCH_SYS(env, T, (step_kernel<S, T, DD, true><<<dim3(grid), dim3(BLOCK), env->lds_bytes, st>>>(C, I, (const T*)action, (const T*)adv, O1)));
    } else {
        DISPATCH_SYS(env, T, (step_kernel<S, T, DD, false><<<dim3(grid), dim3(BLOCK), env->lds_bytes, st>>>(C, I, (const T*)action, (const T*)adv, O)));
    }
    HIP_TRY(hipGetLastError());
    return SCG_OK;
}

template <typename T>
static int launch_rollout(scg_env* env, int k, const scg_rollout_out* o, hipStream_t st) {
    const int grid = (env->cfg.num_envs + BLOCK - 1) / BLOCK;
    const CfgParams<T>* C = (const CfgParams<T>*)env->d_cfg;
    const InstParams<T> I = inst_of<T>(env);
    T* rs = o ? (T*)o->d_reward_sum : nullptr;
    int32_t* dc = o ? o->d_done_count : nullptr;
    int32_t* vc = o ? o->d_violation_count : nullptr;
    T* lo = o ? (T*)o->d_last_obs : nullptr;
    DISPATCH_SYS(env, T, (rollout_random_kernel<S, T, DD><<<dim3(grid), dim3(BLOCK), env->lds_bytes, st>>>(C, I, k, rs, dc, vc, lo)));
    HIP_TRY(hipGetLastError());
    return SCG_OK;
}

extern "C" int scg_reset(scg_env* env, const uint8_t* d_mask, const scg_step_out* out, void* stream) {
    if (!env) return fail(SCG_ERR_INVALID, "env is NULL");
    HIP_TRY(hipSetDevice(env->device));
    int rc = SCG_BY_DTYPE(env, launch_reset, env, d_mask, out, (hipStream_t)stream);
    if (rc == SCG_OK && !d_mask) env->has_reset = true;
    return rc;
}

extern "C" int scg_step_range(scg_env* env, int first_env, int n_envs, const void* d_action, const void* d_adv_action,
                              const scg_step_out* out, void* stream) {
    if (!env) return fail(SCG_ERR_INVALID, "env is NULL");
    if (first_env < 0 || n_envs <= 0 || (int64_t)first_env + n_envs > env->cfg.num_envs)
        return fail(SCG_ERR_INVALID, "env range out of bounds");
    if (first_env % 64 != 0) return fail(SCG_ERR_INVALID, "first_env must be a multiple of 64 (one wave = 64 envs)");
    if (!d_action) return fail(SCG_ERR_INVALID, "d_action is NULL");
    if (!out || !out->d_obs || !out->d_reward || !out->d_done || !out->d_flags)
        return fail(SCG_ERR_INVALID, "scg_step needs d_obs, d_reward, d_done and d_flags");
    if (((uintptr_t)out->d_obs | (uintptr_t)out->d_terminal_obs | (uintptr_t)out->d_ep_stats | (uintptr_t)out->d_fin_stats) & 15)
        return fail(SCG_ERR_INVALID, "per-env row outputs (obs, terminal_obs, ep/fin stats) must be 16-byte aligned");
    // benchmark_env.py:230-235: "You must call env.reset() at least once before using env.step()."
    if (!env->has_reset) return fail(SCG_ERR_STATE, "scg_reset (all envs) must be called before scg_step");
    HIP_TRY(hipSetDevice(env->device));
    return SCG_BY_DTYPE(env, launch_step, env, first_env, n_envs, d_action, d_adv_action, out, (hipStream_t)stream);
}

extern "C" int scg_step(scg_env* env, const void* d_action, const void* d_adv_action, const scg_step_out* out, void* stream) {
    if (!env) return fail(SCG_ERR_INVALID, "env is NULL");
    return scg_step_range(env, 0, env->cfg.num_envs, d_action, d_adv_action, out, stream);
}

extern "C" int scg_rollout_random(scg_env* env, int k_steps, const scg_rollout_out* out, void* stream) {
    if (!env) return fail(SCG_ERR_INVALID, "env is NULL");
    if (k_steps <= 0) return fail(SCG_ERR_INVALID, "k_steps must be positive");
    if (!env->has_reset) return fail(SCG_ERR_STATE, "scg_reset (all envs) must be called before scg_rollout_random");
    HIP_TRY(hipSetDevice(env->device));
    return SCG_BY_DTYPE(env, launch_rollout, env, k_steps, out, (hipStream_t)stream);
}

template <typename T>
static int launch_sequence(scg_env* env, int k, const scg_sequence* q, hipStream_t st) {
    const int grid = (env->cfg.num_envs + BLOCK - 1) / BLOCK;
    const CfgParams<T>* C = (const CfgParams<T>*)env->d_cfg;
    const InstParams<T> I = inst_of<T>(env);
    SeqArgs<T> A;
    A.actions = (const T*)q->d_actions; A.adv = (const T*)q->d_adv_actions; A.k_steps = k;
    A.obs = (T*)q->d_obs; A.reward = (T*)q->d_reward; A.done = q->d_done; A.flags = q->d_flags;
    A.terminal_obs = (T*)q->d_terminal_obs; A.mse = (T*)q->d_mse; A.c_values = (T*)q->d_c_values;
    A.ep_stats = (T*)q->d_ep_stats; A.fin_stats = (T*)q->d_fin_stats;
    A.state = (T*)q->d_state; A.noisy_action = (T*)q->d_noisy_action;
    DISPATCH_SYS(env, T, (step_sequence_kernel<S, T, DD><<<dim3(grid), dim3(BLOCK), env->lds_bytes, st>>>(C, I, A)));
    HIP_TRY(hipGetLastError());
    return SCG_OK;
}

extern "C" int scg_step_sequence(scg_env* env, int k_steps, const scg_sequence* seq, void* stream) {
    if (!env || !seq) return fail(SCG_ERR_INVALID, "NULL argument to scg_step_sequence");
    if (k_steps <= 0) return fail(SCG_ERR_INVALID, "k_steps must be positive");
    if (!seq->d_actions || !seq->d_obs || !seq->d_reward || !seq->d_done || !seq->d_flags)
        return fail(SCG_ERR_INVALID, "scg_step_sequence needs d_actions, d_obs, d_reward, d_done and d_flags");
    if (((uintptr_t)seq->d_obs | (uintptr_t)seq->d_terminal_obs | (uintptr_t)seq->d_ep_stats | (uintptr_t)seq->d_fin_stats) & 15)
        return fail(SCG_ERR_INVALID, "per-env row outputs (obs, terminal_obs, ep/fin stats) must be 16-byte aligned");
    {
        const size_t esz = env->dtype == SCG_F64 ? 8 : 4;
        int32_t nx, nu, nobs, ns, np;
        if (scg_dims(&env->cfg, &nx, &nu, &nobs, &ns, &np) != SCG_OK) return SCG_ERR_INVALID;
        if (((size_t)env->cfg.num_envs * nobs * esz) % 16 != 0)
            return fail(SCG_ERR_INVALID, "num_envs x obs_dim x sizeof(T) must be a multiple of 16 (row alignment of the stacked arrays)");
    }
    if (!env->has_reset) return fail(SCG_ERR_STATE, "scg_reset (all envs) must be called before scg_step_sequence");
    HIP_TRY(hipSetDevice(env->device));
    return SCG_BY_DTYPE(env, launch_sequence, env, k_steps, seq, (hipStream_t)stream);
}

extern "C" int scg_rollout_policy(scg_env* env, const scg_policy* pol, int k_steps, const scg_policy_rollout* out, void* stream) {
    if (!env || !pol || !out) return fail(SCG_ERR_INVALID, "NULL argument to scg_rollout_policy");
#if defined(SCG_SPEC) && defined(SCG_POLICY_H) && SCG_SPEC_DTYPE == 0
    if (k_steps <= 0) return fail(SCG_ERR_INVALID, "k_steps must be positive");
    if (!env->has_reset) return fail(SCG_ERR_STATE, "scg_reset (all envs) must be called before scg_rollout_policy");
    if (pol->hidden != SCG_POLICY_H || pol->activation != SCG_POLICY_ACT)
        return fail(SCG_ERR_INVALID, "this library was compiled for another policy shape (hidden / activation)");
    if (!pol->d_params || !out->d_obs || !out->d_act || !out->d_logp || !out->d_reward || !out->d_done || !out->d_flags)
        return fail(SCG_ERR_INVALID, "scg_rollout_policy needs d_params, d_obs, d_act, d_logp, d_reward, d_done and d_flags");
    if (((uintptr_t)out->d_obs | (uintptr_t)out->d_terminal_obs | (uintptr_t)out->d_ep_stats | (uintptr_t)out->d_episode_acc) & 15)
        return fail(SCG_ERR_INVALID, "row outputs must be 16-byte aligned");
    HIP_TRY(hipSetDevice(env->device));
    constexpr int S = SCG_SPEC_SYS;
    constexpr bool DD = SCG_SPEC_DIST != 0;
    if constexpr ((scg_make_spec_cfg<float>().nobs * sizeof(float)) % 16 == 0) {      // rows leave as 16-byte pieces: obs[t] must stay aligned
        if (((size_t)env->cfg.num_envs * scg_make_spec_cfg<float>().nobs * sizeof(float)) % 16 != 0)
            return fail(SCG_ERR_INVALID, "num_envs x obs_dim x 4 must be a multiple of 16 (row alignment of the [t]-stacked obs)");
    }
    PolicyArgs A;
    A.params = pol->d_params; A.W1 = pol->W1; A.b1 = pol->b1; A.W2 = pol->W2; A.b2 = pol->b2; A.W3 = pol->W3; A.b3 = pol->b3;
    A.logstd_off = pol->logstd_off; A.deterministic = pol->deterministic; A.k_steps = k_steps;
    A.obs = (float*)out->d_obs; A.act = (float*)out->d_act; A.logp = (float*)out->d_logp; A.reward = (float*)out->d_reward;
    A.done = out->d_done; A.flags = out->d_flags; A.terminal_obs = (float*)out->d_terminal_obs;
    A.ep_stats = (float*)out->d_ep_stats; A.episode_acc = (float*)out->d_episode_acc; A.max_episodes = out->max_episodes;
    const InstParams<float> I = inst_of<float>(env);
    constexpr int nobs = scg_make_spec_cfg<float>().nobs;
    // Launch geometry (scg_env_kernels.h): envs per wave 32 while that gives at most two waves per SIMD (2 x 4 x 256 CUs), else 64; waves
    // per workgroup 4 up to one wave per SIMD, 8 above (two waves per SIMD behind one weight image).  SCG_ROLLOUT_EPW = 32 | 64 and
    // SCG_ROLLOUT_WPW = 4 | 8 override (tests run the geometries on small batches; results do not depend on them: Philox streams
    // are per env, the MFMA sequence per env is the same).
    int epw = env->cfg.num_envs <= 65536 ? 32 : 64;
    int wpw = env->cfg.num_envs <= 32768 ? 4 : 8;
    if (const char* o = getenv("SCG_ROLLOUT_EPW")) { if (atoi(o) == 32 || atoi(o) == 64) epw = atoi(o); }
    if (const char* o = getenv("SCG_ROLLOUT_WPW")) { if (atoi(o) == 4 || atoi(o) == 8) wpw = atoi(o); }
    const size_t bytes = MlpLds<nobs, SCG_POLICY_H, Dims<S>::NU, 16>::END * sizeof(float) + (size_t)wpw * 64 * nobs * sizeof(float);
    const size_t bytes8 = MlpLds<nobs, SCG_POLICY_H, Dims<S>::NU, 16>::END * sizeof(float) + (size_t)8 * 64 * nobs * sizeof(float);
    static scg::PerDeviceOnce attr;         // (per device, scg_once.h: the caller has made the handle's device current)
    int attr_dev;
    if (attr.pending(&attr_dev)) {
        HIP_TRY(hipFuncSetAttribute((const void*)rollout_policy_kernel<S, DD, 64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes8));
        HIP_TRY(hipFuncSetAttribute((const void*)rollout_policy_kernel<S, DD, 32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes8));
        HIP_TRY(hipFuncSetAttribute((const void*)rollout_policy_kernel<S, DD, 64, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes8));
        HIP_TRY(hipFuncSetAttribute((const void*)rollout_policy_kernel<S, DD, 32, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes8));
        attr.commit(attr_dev);
    }
    const int per_wg = epw * wpw;
    const dim3 grid((env->cfg.num_envs + per_wg - 1) / per_wg), block(64 * wpw);
    hipStream_t st = (hipStream_t)stream;
    if (epw == 64 && wpw == 4) rollout_policy_kernel<S, DD, 64, 4><<<grid, block, bytes, st>>>(I, A);
    else if (epw == 64) rollout_policy_kernel<S, DD, 64, 8><<<grid, block, bytes, st>>>(I, A);
    else if (wpw == 4) rollout_policy_kernel<S, DD, 32, 4><<<grid, block, bytes, st>>>(I, A);
    else rollout_policy_kernel<S, DD, 32, 8><<<grid, block, bytes, st>>>(I, A);
    HIP_TRY(hipGetLastError());
    return SCG_OK;
#else
    (void)k_steps; (void)stream;
    return fail(SCG_ERR_INVALID, "scg_rollout_policy needs a library specialised for the task config and the policy shape "
                                 "(float32): build it with _lib.build_spec(cfg, policy=(hidden, activation))");
#endif
}

// ---- host accessors ---------------------------------------------------------------------------
template <typename T>
static int copy_soa(scg_env* env, void* d_base, int n_arrays, double* h_out, const double* h_in, int first, int n, hipStream_t st) {
    const size_t N = (size_t)env->cfg.num_envs;
    if (first < 0 || n < 0 || (size_t)first + n > N) return fail(SCG_ERR_INVALID, "env range out of bounds");
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<T> tmp((size_t)n);
    for (int k = 0; k < n_arrays; ++k) {
        T* d = (T*)d_base + (size_t)k * N + first;
        if (h_in) {
            for (int i = 0; i < n; ++i) tmp[i] = (T)h_in[(size_t)i * n_arrays + k];
            HIP_TRY(hipMemcpy(d, tmp.data(), (size_t)n * sizeof(T), hipMemcpyHostToDevice));
        } else {
            HIP_TRY(hipMemcpy(tmp.data(), d, (size_t)n * sizeof(T), hipMemcpyDeviceToHost));
            for (int i = 0; i < n; ++i) h_out[(size_t)i * n_arrays + k] = (double)tmp[i];
        }
    }
    return SCG_OK;
}

extern "C" int scg_set_state(scg_env* env, const double* h_state, int first_env, int n, void* stream) {
    if (!env || !h_state) return fail(SCG_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(env->device));
    return env->dtype == SCG_F64 ? copy_soa<double>(env, env->d_state, env->ns, nullptr, h_state, first_env, n, (hipStream_t)stream)
                                 : copy_soa<float>(env, env->d_state, env->ns, nullptr, h_state, first_env, n, (hipStream_t)stream);
}
extern "C" int scg_get_state(scg_env* env, double* h_state, int first_env, int n, void* stream) {
    if (!env || !h_state) return fail(SCG_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(env->device));
    return env->dtype == SCG_F64 ? copy_soa<double>(env, env->d_state, env->ns, h_state, nullptr, first_env, n, (hipStream_t)stream)
                                 : copy_soa<float>(env, env->d_state, env->ns, h_state, nullptr, first_env, n, (hipStream_t)stream);
}
extern "C" int scg_set_params(scg_env* env, const double* h_params, int first_env, int n, void* stream) {
    if (!env || !h_params) return fail(SCG_ERR_INVALID, "NULL argument");
    if (!env->cfg.randomized_inertial_prop) return fail(SCG_ERR_STATE, "per-env parameters exist only with randomized_inertial_prop");
    HIP_TRY(hipSetDevice(env->device));
    return env->dtype == SCG_F64 ? copy_soa<double>(env, env->d_param, env->np, nullptr, h_params, first_env, n, (hipStream_t)stream)
                                 : copy_soa<float>(env, env->d_param, env->np, nullptr, h_params, first_env, n, (hipStream_t)stream);
}
extern "C" int scg_get_params(scg_env* env, double* h_params, int first_env, int n, void* stream) {
    if (!env || !h_params) return fail(SCG_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(env->device));
    if (!env->cfg.randomized_inertial_prop) {
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < env->np; ++k) h_params[(size_t)i * env->np + k] = env->cfg.base_param[k];
        return SCG_OK;
    }
    return env->dtype == SCG_F64 ? copy_soa<double>(env, env->d_param, env->np, h_params, nullptr, first_env, n, (hipStream_t)stream)
                                 : copy_soa<float>(env, env->d_param, env->np, h_params, nullptr, first_env, n, (hipStream_t)stream);
}
extern "C" int scg_rng_layout_version(void) { return SCG_RNG_LAYOUT_VERSION; }
extern "C" int scg_set_step_wsback(scg_env* env, int wsback_min_envs, int wsback_max_envs) {
    if (!env) return fail(SCG_ERR_INVALID, "env is NULL");
    if (wsback_min_envs >= 0) env->wsback_min = wsback_min_envs;
    if (wsback_max_envs >= 0) env->wsback_max = wsback_max_envs;
    return SCG_OK;
}
extern "C" int scg_set_step_launch(scg_env* env, int split_max_envs, int wide_min_envs) {
    if (!env) return fail(SCG_ERR_INVALID, "env is NULL");
    (void)split_max_envs;       // (the split launch of rounds 5-6 is gone — see scg_env_kernels.h; the argument is accepted and ignored)
    if (wide_min_envs >= 0) env->wide_min = wide_min_envs;
    return SCG_OK;
}

extern "C" int scg_set_seed(scg_env* env, uint64_t seed) {
    if (!env) return fail(SCG_ERR_INVALID, "env is NULL");
    env->cfg.seed = seed;
    return SCG_OK;
}

extern "C" int scg_set_counters(scg_env* env, const int32_t* h_step, const uint32_t* h_episode, int first_env, int n, void* stream) {
    if (!env) return fail(SCG_ERR_INVALID, "env is NULL");
    if (first_env < 0 || n < 0 || first_env + n > env->cfg.num_envs) return fail(SCG_ERR_INVALID, "env range out of bounds");
    HIP_TRY(hipSetDevice(env->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (h_step) HIP_TRY(hipMemcpy(env->d_step + first_env, h_step, (size_t)n * 4, hipMemcpyHostToDevice));
    if (h_episode) HIP_TRY(hipMemcpy(env->d_episode + first_env, h_episode, (size_t)n * 4, hipMemcpyHostToDevice));
    return SCG_OK;
}
extern "C" int scg_get_counters(scg_env* env, int32_t* h_step, uint32_t* h_episode, int first_env, int n, void* stream) {
    if (!env) return fail(SCG_ERR_INVALID, "env is NULL");
    if (first_env < 0 || n < 0 || first_env + n > env->cfg.num_envs) return fail(SCG_ERR_INVALID, "env range out of bounds");
    HIP_TRY(hipSetDevice(env->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (h_step) HIP_TRY(hipMemcpy(h_step, env->d_step + first_env, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (h_episode) HIP_TRY(hipMemcpy(h_episode, env->d_episode + first_env, (size_t)n * 4, hipMemcpyDeviceToHost));
    return SCG_OK;
}

// ---- GAE ----------------------------------------------------------------------------------------
template <typename T>
static int launch_gae(void* rew, const void* v, const void* mask, const void* term, const void* last, void* ret, void* adv,
                      int Tn, int N, double gamma, double lam, int use_gae, hipStream_t st) {
    // >= 16 waves' worth of envs: the per-env walk already fills the chip with coalesced traffic;
    // below that, parallelise over time with the wave-level segmented scan.
    constexpr int CH = 8;
    const int S = (Tn + CH - 1) / CH;
    if (N >= 1024 && S >= 2 && S <= 16) {
        // many envs, a collector's horizon: CH time steps per thread, one memory round trip, S x the waves (scg_gae_kernels.h (a'))
        gae_seg_kernel<T, CH><<<dim3((N + 63) / 64), dim3(64, S), (size_t)S * 4 * 64 * sizeof(T), st>>>(
            (T*)rew, (const T*)v, (const T*)mask, (const T*)term, (const T*)last, (T*)ret, (T*)adv, Tn, N, (T)gamma, (T)lam, use_gae);
    } else if (N >= 1024 || Tn < 64) {
        const int grid = (N + GAE_BLOCK - 1) / GAE_BLOCK;
        gae_env_kernel<T><<<dim3(grid), dim3(GAE_BLOCK), 0, st>>>((T*)rew, (const T*)v, (const T*)mask, (const T*)term,
                                                             (const T*)last, (T*)ret, (T*)adv, Tn, N, (T)gamma, (T)lam, use_gae);
    } else {
        gae_wave_kernel<T><<<dim3(N), dim3(64), 0, st>>>((T*)rew, (const T*)v, (const T*)mask, (const T*)term,
                                                        (const T*)last, (T*)ret, (T*)adv, Tn, N, (T)gamma, (T)lam, use_gae);
    }
    HIP_TRY(hipGetLastError());
    return SCG_OK;
}

template <typename T>
static int launch_prior(scg_env* env, const void* x, const void* u, int n, double eps, void* f, void* A, void* B, void* xn,
                        hipStream_t st) {
    const int grid = (n + 255) / 256;
    const CfgParams<T>* C = (const CfgParams<T>*)env->d_cfg;
    const InstParams<T> I = inst_of<T>(env);
    DISPATCH_SYS(env, T, (prior_model_kernel<S, T, DD><<<dim3(grid), dim3(256), 0, st>>>(C, I, n, (const T*)x, (const T*)u, (T)eps,
                                                                                        (T*)f, (T*)A, (T*)B, (T*)xn)));
    HIP_TRY(hipGetLastError());
    return SCG_OK;
}

extern "C" int scg_prior_model(scg_env* env, const void* d_x, const void* d_u, int n, double eps, void* d_f, void* d_A,
                               void* d_B, void* d_xnext, void* stream) {
    if (!env) return fail(SCG_ERR_INVALID, "env is NULL");
    if (!d_x || !d_u || n <= 0) return fail(SCG_ERR_INVALID, "scg_prior_model needs d_x, d_u and n > 0");
    if (eps <= 0.0) return fail(SCG_ERR_INVALID, "eps must be positive");
    HIP_TRY(hipSetDevice(env->device));
    return SCG_BY_DTYPE(env, launch_prior, env, d_x, d_u, n, eps, d_f, d_A, d_B, d_xnext, (hipStream_t)stream);
}

extern "C" int scg_gae(int dtype, void* d_rew, const void* d_v, const void* d_mask, const void* d_terminal_v,
                       const void* d_last_v, void* d_ret, void* d_adv, int T, int N, double gamma, double lam,
                       int use_gae, void* stream) {
    if (!d_rew || !d_v || !d_mask || !d_last_v || !d_ret || !d_adv) return fail(SCG_ERR_INVALID, "NULL buffer passed to scg_gae");
    if (T <= 0 || N <= 0) return fail(SCG_ERR_INVALID, "T and N must be positive");
    if (dtype == SCG_F64)
        return launch_gae<double>(d_rew, d_v, d_mask, d_terminal_v, d_last_v, d_ret, d_adv, T, N, gamma, lam, use_gae, (hipStream_t)stream);
    if (dtype == SCG_F32)
        return launch_gae<float>(d_rew, d_v, d_mask, d_terminal_v, d_last_v, d_ret, d_adv, T, N, gamma, lam, use_gae, (hipStream_t)stream);
    return fail(SCG_ERR_INVALID, "unknown dtype");
}

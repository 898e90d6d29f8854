// scg_kernels.hip — gfx950 kernels + the C ABI of libscg_hip.so (include/scg_hip.h).
//
// Kernel geometry (CDNA4: 64-lane waves, 256 CUs in 8 XCDs, 160 KB LDS/CU):
//   * one thread = one environment; 256-thread workgroups (4 waves, one per SIMD of a CU);
//   * raw simulator state is SoA ([component][env]) so each wave's loads/stores are 256 contiguous bytes;
//   * the X_GOAL reference table and nothing else is shared between lanes: it is staged once per
//     workgroup into LDS (rows are indexed by each env's own step counter after de-synchronised
//     resets, so a scalar/broadcast path is not enough);
//   * the whole control step (action pre-processing, disturbance draws, PYB_FREQ/CTRL_FREQ integrator
//     substeps, observation/reward/done/info/constraints, episode statistics, auto-reset) is ONE launch;
//   * no MFMA: there is no dense contraction on this path, the bound is HBM bandwidth.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/scg_hip.h"
#include "scg_env_core.h"
#include "scg_params.h"
#include "scg_rng.h"

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
extern "C" size_t scg_sizeof_config(void) { return sizeof(scg_config); }
extern "C" size_t scg_sizeof_step_out(void) { return sizeof(scg_step_out); }

constexpr int BLOCK = 256;
constexpr size_t LDS_GOAL_LIMIT = 64 * 1024;

// ------------------------------------------------------------------ kernels
template <typename T>
__device__ __forceinline__ const T* stage_goal(const DevParams<T>* __restrict__ P, unsigned char* smem) {
    if (!P->goal_in_lds) return P->x_goal;
    T* tab = reinterpret_cast<T*>(smem);
    const int n = P->goal_rows * P->nx;
    for (int k = threadIdx.x; k < n; k += blockDim.x) tab[k] = P->x_goal[k];
    __syncthreads();
    return tab;
}

template <int SYS, typename T, bool DIST>
__global__ __launch_bounds__(BLOCK) void reset_kernel(const DevParams<T>* __restrict__ P,
                                                      const uint8_t* __restrict__ mask, StepOut<T> O) {
    extern __shared__ __align__(16) unsigned char smem[];
    using Ops = EnvOps<SYS, T, DIST>;
    using D = Dims<SYS>;
    const T* goal = stage_goal(P, smem);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P->num_envs) return;
    if (mask && !mask[i]) return;
    const RngKey key{P->key0, P->key1};
    typename Ops::E e;
    Ops::load(P, i, e);
    Ops::reset(P, i, e, key);
    T st[D::NX];
    Ops::state_vector(e, st);
    if (O.obs) Ops::write_obs(P, goal, st, e, key, 1, 0u, 0, i, O.obs + (size_t)i * P->nobs);
    if (O.c_values && P->n_state_con_rows > 0) Ops::constraints(P, st, st, O.c_values + i, (size_t)P->num_envs, true);
    if (O.state) {
#pragma unroll
        for (int k = 0; k < D::NX; ++k) O.state[(size_t)k * P->num_envs + i] = st[k];
    }
    if (O.ep_return) O.ep_return[i] = (T)0;
    if (O.ep_length) O.ep_length[i] = 0;
    if (O.ep_violation) O.ep_violation[i] = (T)0;
    if (O.ep_mse) O.ep_mse[i] = (T)0;
    if (P->oob_attr) P->oob_attr[i] = 0;
    Ops::store(P, i, e, true);
}

template <int SYS, typename T, bool DIST>
__global__ __launch_bounds__(BLOCK) void step_kernel(const DevParams<T>* __restrict__ P, const T* __restrict__ action,
                                                     const T* __restrict__ adv, StepOut<T> O) {
    extern __shared__ __align__(16) unsigned char smem[];
    using Ops = EnvOps<SYS, T, DIST>;
    using D = Dims<SYS>;
    const T* goal = stage_goal(P, smem);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = P->num_envs;
    if (i >= N) return;
    const RngKey key{P->key0, P->key1};
    typename Ops::E e;
    Ops::load(P, i, e);
    T act[D::NU];
#pragma unroll
    for (int j = 0; j < D::NU; ++j) act[j] = action[(size_t)i * D::NU + j];
    T advv[D::DYN > D::NU ? D::DYN : D::NU];
    const T* advp = nullptr;
    if (adv && P->adversary_channel >= 0) {
        const int ad = P->adversary_channel == SCG_CH_ACTION ? D::NU : D::DYN;
        for (int j = 0; j < ad; ++j) advv[j] = adv[(size_t)i * ad + j];
        advp = advv;
    }
    T st[D::NX], noisy[D::NU];
    const int32_t c0 = e.step;
    typename Ops::StepResult r = Ops::step(P, goal, e, act, advp, key, i, st, noisy,
                                           O.c_values ? O.c_values + i : nullptr, (size_t)N);
    if (O.reward) O.reward[i] = r.reward;
    if (O.done) O.done[i] = r.done ? 1 : 0;
    if (O.flags) O.flags[i] = r.flags;
    if (O.mse) O.mse[i] = r.mse;
    if (O.noisy_action) {
#pragma unroll
        for (int j = 0; j < D::NU; ++j) O.noisy_action[(size_t)j * N + i] = noisy[j];
    }
    // columnar VecRecordEpisodeStatistics (record_episode_statistics.py:139-166)
    if (O.ep_return) {
        const T acc = O.ep_return[i] + r.reward;
        if (r.done && O.fin_return) O.fin_return[i] = acc;
        O.ep_return[i] = r.done ? (T)0 : acc;
    }
    if (O.ep_length) {
        const int32_t acc = O.ep_length[i] + 1;
        if (r.done && O.fin_length) O.fin_length[i] = acc;
        O.ep_length[i] = r.done ? 0 : acc;
    }
    if (O.ep_violation) {
        const T acc = O.ep_violation[i] + ((r.flags & FLAG_VIOLATION) ? (T)1 : (T)0);
        if (r.done && O.fin_violation) O.fin_violation[i] = acc;
        O.ep_violation[i] = r.done ? (T)0 : acc;
    }
    if (O.ep_mse) {
        const T acc = O.ep_mse[i] + r.mse;
        if (r.done && O.fin_mse) O.fin_mse[i] = acc;
        O.ep_mse[i] = r.done ? (T)0 : acc;
    }
    // observation of the step: terminal_observation where done, else the returned obs
    const bool do_reset = r.done && P->auto_reset;
    if (r.done && !P->auto_reset && O.terminal_obs)
        Ops::write_obs(P, goal, st, e, key, c0 + 2, (uint32_t)(c0 + 1), c0, i, O.terminal_obs + (size_t)i * P->nobs);
    if (do_reset) {
        if (O.terminal_obs) Ops::write_obs(P, goal, st, e, key, c0 + 2, (uint32_t)(c0 + 1), c0, i, O.terminal_obs + (size_t)i * P->nobs);
        Ops::reset(P, i, e, key);               // auto-reset (dummy_vec_env.py:33-38)
        Ops::state_vector(e, st);
        if (O.obs) Ops::write_obs(P, goal, st, e, key, 1, 0u, 0, i, O.obs + (size_t)i * P->nobs);
    } else {
        if (O.obs) Ops::write_obs(P, goal, st, e, key, c0 + 2, (uint32_t)(c0 + 1), c0, i, O.obs + (size_t)i * P->nobs);
    }
    if (O.state) {
#pragma unroll
        for (int k = 0; k < D::NX; ++k) O.state[(size_t)k * N + i] = st[k];
    }
    Ops::store(P, i, e, do_reset);
}

template <int SYS, typename T, bool DIST>
__global__ __launch_bounds__(BLOCK) void rollout_random_kernel(const DevParams<T>* __restrict__ P, int k_steps,
                                                               T* __restrict__ reward_sum, int32_t* __restrict__ done_count,
                                                               int32_t* __restrict__ violation_count, T* __restrict__ last_obs) {
    extern __shared__ __align__(16) unsigned char smem[];
    using Ops = EnvOps<SYS, T, DIST>;
    using D = Dims<SYS>;
    const T* goal = stage_goal(P, smem);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P->num_envs) return;
    const RngKey key{P->key0, P->key1};
    typename Ops::E e;
    Ops::load(P, i, e);
    T rsum = (T)0;
    int32_t dones = 0, viols = 0;
    bool dirty = false;
    T st[D::NX];
    for (int k = 0; k < k_steps; ++k) {
        // actions ~ U(-1, 1): Philox channel 4, item 0, word j
        U4 w = rng_words(key, e.gid, e.episode, (uint32_t)e.step, rng_tag(RNG_CH_RANDOM_ACTION, 0, 0));
        T act[D::NU], noisy[D::NU];
#pragma unroll
        for (int j = 0; j < D::NU; ++j) act[j] = (T)-1 + (T)2 * u01<T>(u4_get(w, j));
        typename Ops::StepResult r = Ops::step(P, goal, e, act, nullptr, key, i, st, noisy, nullptr, 0);
        rsum += r.reward;
        viols += (r.flags & FLAG_VIOLATION) ? 1 : 0;
        if (r.done) {
            ++dones;
            if (P->auto_reset) {
                dirty = true;
                Ops::reset(P, i, e, key);
                Ops::state_vector(e, st);
            }
        }
    }
    if (reward_sum) reward_sum[i] = rsum;
    if (done_count) done_count[i] = dones;
    if (violation_count) violation_count[i] = viols;
    if (last_obs) {
        const bool fresh = e.step == 0;
        const int32_t c0 = e.step - 1;
        Ops::write_obs(P, goal, st, e, key, fresh ? 1 : c0 + 2, fresh ? 0u : (uint32_t)(c0 + 1), fresh ? 0 : c0, i,
                       last_obs + (size_t)i * P->nobs);
    }
    Ops::store(P, i, e, dirty);
}

// ---- GAE / returns (controllers/ppo/ppo_utils.py:374-400), buffers [T][N] -------------------------
// (a) one thread per env walking T backwards: every load/store is coalesced across the wave.
template <typename T>
__global__ __launch_bounds__(BLOCK) void gae_env_kernel(T* __restrict__ rew, const T* __restrict__ v, const T* __restrict__ mask,
                                                        const T* __restrict__ term_v, const T* __restrict__ last_v,
                                                        T* __restrict__ ret, T* __restrict__ adv, int Tn, int N, T gamma,
                                                        T lam, int use_gae) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    T run_ret = last_v[n], run_adv = (T)0, v_next = last_v[n];
    for (int t = Tn - 1; t >= 0; --t) {
        const size_t idx = (size_t)t * N + n;
        T r = rew[idx];
        if (term_v) { r += gamma * term_v[idx]; rew[idx] = r; }
        const T m = mask[idx], vt = v[idx];
        run_ret = r + gamma * m * run_ret;
        if (use_gae) {
            const T delta = r + gamma * m * v_next - vt;
            run_adv = run_adv * lam * gamma * m + delta;
        } else {
            run_adv = run_ret - vt;
        }
        ret[idx] = run_ret;
        adv[idx] = run_adv;
        v_next = vt;
    }
}

// (b) small N (the reference's own shape, T=1000 x N=4): one 64-lane wave per env, segmented affine scan
// over time.  Each element is the map x -> a x + b; (a,b) o (a',b') = (a a', b + a b'); mask = 0 gives a = 0,
// i.e. the segment boundary.  Lane L of a chunk holds time t_hi - L, so an inclusive scan along the lanes
// composes the maps in the order the sequential recursion applies them.
template <typename T>
__device__ __forceinline__ void affine_scan64(T& a, T& b) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T pa = __shfl_up(a, off, 64), pb = __shfl_up(b, off, 64);
        if (lane >= off) { b = a * pb + b; a = a * pa; }
    }
}

template <typename T>
__global__ __launch_bounds__(64) void gae_wave_kernel(T* __restrict__ rew, const T* __restrict__ v, const T* __restrict__ mask,
                                                      const T* __restrict__ term_v, const T* __restrict__ last_v,
                                                      T* __restrict__ ret, T* __restrict__ adv, int Tn, int N, T gamma, T lam,
                                                      int use_gae) {
    const int n = blockIdx.x;
    const int lane = threadIdx.x;
    T carry_ret = last_v[n], carry_adv = (T)0, carry_v = last_v[n];
    for (int t_hi = Tn - 1; t_hi >= 0; t_hi -= 64) {
        const int t = t_hi - lane;
        const bool live = t >= 0;
        const size_t idx = live ? (size_t)t * N + n : 0;
        T r = (T)0, m = (T)1, vt = (T)0;
        if (live) {
            r = rew[idx];
            if (term_v) { r += gamma * term_v[idx]; rew[idx] = r; }
            m = mask[idx];
            vt = v[idx];
        }
        T v_next = __shfl_up(vt, 1, 64);
        if (lane == 0) v_next = carry_v;
        // returns: ret_t = r_t + (gamma m_t) ret_{t+1}
        T a1 = live ? gamma * m : (T)1, b1 = live ? r : (T)0;
        affine_scan64(a1, b1);
        const T my_ret = a1 * carry_ret + b1;
        T my_adv;
        if (use_gae) {
            const T delta = r + gamma * m * v_next - vt;
            T a2 = live ? lam * gamma * m : (T)1, b2 = live ? delta : (T)0;
            affine_scan64(a2, b2);
            my_adv = a2 * carry_adv + b2;
        } else {
            my_adv = my_ret - vt;
        }
        if (live) { ret[idx] = my_ret; adv[idx] = my_adv; }
        // carry = value at the earliest time of this chunk = last live lane
        const int last_lane = t_hi >= 63 ? 63 : t_hi;
        carry_ret = __shfl(my_ret, last_lane, 64);
        carry_adv = __shfl(my_adv, last_lane, 64);
        carry_v = __shfl(vt, last_lane, 64);
    }
}

// ------------------------------------------------------------------ host side
struct scg_env {
    scg_config cfg;
    int device;
    int dtype;
    int nx, nu, nobs, ns, np;
    int n_offset_slots;
    size_t lds_bytes;
    void* d_params;          // DevParams<T> on device
    void* d_goal;            // X_GOAL table on device
    // workspace partition (device pointers)
    void* d_state;
    void* d_param;
    int32_t* d_step;
    uint32_t* d_episode;
    int32_t* d_dist_offset;
    uint8_t* d_oob;
    bool has_reset;
    bool has_dist;
};

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
    if (c->integrator != SCG_INT_PYB_EULER) return fail(SCG_ERR_INVALID, "only SCG_INT_PYB_EULER is implemented");
    if (c->num_envs <= 0) return fail(SCG_ERR_INVALID, "num_envs must be positive");
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
    p.goal_in_lds = e->lds_bytes > 0;
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
        d.sign = (T)s.sign; d.b = (T)s.b;
        // rounding to `decimals` places is only meaningful in double precision (np.round(., 8) on float64)
        const bool round = s.round_scale > 0 && sizeof(T) == 8;
        d.round_scale = round ? (T)s.round_scale : (T)0;
        d.inv_round_scale = round ? (T)(1.0 / s.round_scale) : (T)0;
        for (int j = 0; j < SCG_MAX_STATE; ++j) d.coef[j] = (T)s.coef[j];
    }
    for (int q = 0; q < SCG_MAX_QUAD_CON; ++q)
        for (int j = 0; j < SCG_MAX_STATE * SCG_MAX_STATE; ++j) p.quad_P[q][j] = (T)c.quad_P[q][j];
    // box rows regrouped by variable slot (state k -> k, input j -> SCG_MAX_STATE + j)
    {
        int n = 0, generic = 0;
        std::vector<int> state_pos(c.n_con_rows, 0);
        int sp = 0;
        for (int r = 0; r < c.n_con_rows; ++r) { state_pos[r] = sp; if (c.con[r].var == 0) ++sp; }
        for (int slot = 0; slot < SCG_MAX_STATE + SCG_MAX_ACTION; ++slot) {
            p.bv_first[slot] = n;
            const int var = slot < SCG_MAX_STATE ? 0 : 1, index = slot < SCG_MAX_STATE ? slot : slot - SCG_MAX_STATE;
            for (int r = 0; r < c.n_con_rows; ++r) {
                const DevRow<T>& d = p.con[r];
                if ((d.kind == SCG_ROW_SPARSE || d.kind == SCG_ROW_ABS) && d.var == var && d.index == index) {
                    p.bv_row[n] = r; p.bv_state_pos[n] = state_pos[r];
                    p.bv_flags[n] = (d.strict ? 1 : 0) | (d.kind == SCG_ROW_ABS ? 2 : 0);
                    p.bv_sign[n] = d.sign; p.bv_b[n] = d.b; p.bv_round[n] = d.round_scale; p.bv_inv_round[n] = d.inv_round_scale;
                    ++n;
                }
            }
        }
        p.bv_first[SCG_MAX_STATE + SCG_MAX_ACTION] = n;
        for (int r = 0; r < c.n_con_rows; ++r)
            if (p.con[r].kind != SCG_ROW_SPARSE && p.con[r].kind != SCG_ROW_ABS) ++generic;
        p.n_box_rows = n; p.n_generic_rows = generic;
    }
    p.x_goal = (const T*)e->d_goal;
    p.state = (T*)e->d_state; p.param = (T*)e->d_param; p.step = e->d_step; p.episode = e->d_episode;
    p.dist_offset = e->d_dist_offset; p.oob_attr = e->d_oob;
}

template <typename T>
static int upload(scg_env* e, const double* h_x_goal) {
    const size_t ng = (size_t)e->cfg.goal_rows * e->nx;
    std::vector<T> tab(ng);
    for (size_t k = 0; k < ng; ++k) tab[k] = (T)h_x_goal[k];
    HIP_TRY(hipMalloc(&e->d_goal, ng * sizeof(T)));
    HIP_TRY(hipMemcpy(e->d_goal, tab.data(), ng * sizeof(T), hipMemcpyHostToDevice));
    DevParams<T>* hp = new (std::nothrow) DevParams<T>;
    if (!hp) return fail(SCG_ERR_NOMEM, "host allocation failed");
    fill_params<T>(e, e->cfg, *hp);
    hipError_t err = hipMalloc(&e->d_params, sizeof(DevParams<T>));
    if (err == hipSuccess) err = hipMemcpy(e->d_params, hp, sizeof(DevParams<T>), hipMemcpyHostToDevice);
    delete hp;
    if (err != hipSuccess) return fail(SCG_ERR_HIP, std::string("uploading parameters: ") + hipGetErrorString(err));
    return SCG_OK;
}

extern "C" int scg_create(const scg_config* cfg, const double* h_x_goal, int device, void* d_workspace,
                          size_t workspace_bytes, scg_env** out) {
    if (int rc = validate(cfg)) return rc;
    if (!h_x_goal || !d_workspace || !out) return fail(SCG_ERR_INVALID, "NULL argument to scg_create");
    const Layout L = layout_of(cfg);
    if (workspace_bytes < L.total) return fail(SCG_ERR_INVALID, "workspace too small (see scg_workspace_bytes)");
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
    e->lds_bytes = tab_bytes <= LDS_GOAL_LIMIT ? align_up(tab_bytes, 16) : 0;
    unsigned char* w = (unsigned char*)d_workspace;
    e->d_state = w + L.state; e->d_param = w + L.param; e->d_step = (int32_t*)(w + L.step);
    e->d_episode = (uint32_t*)(w + L.episode); e->d_dist_offset = (int32_t*)(w + L.offsets); e->d_oob = w + L.oob;
    e->d_params = nullptr; e->d_goal = nullptr; e->has_reset = false;
    e->has_dist = cfg->n_dist[0] > 0 || cfg->n_dist[1] > 0 || cfg->n_dist[2] > 0 || cfg->adversary_channel >= 0;
    hipError_t err = hipMemset(d_workspace, 0, L.total);
    if (err == hipSuccess) err = hipMemset(e->d_episode, 0xff, (size_t)cfg->num_envs * 4);   // first reset -> episode 0
    if (err != hipSuccess) { delete e; return fail(SCG_ERR_HIP, std::string("hipMemset: ") + hipGetErrorString(err)); }
    int rc = cfg->dtype == SCG_F64 ? upload<double>(e, h_x_goal) : upload<float>(e, h_x_goal);
    if (rc) { if (e->d_goal) (void)hipFree(e->d_goal); if (e->d_params) (void)hipFree(e->d_params); delete e; return rc; }
    *out = e;
    return SCG_OK;
}

extern "C" int scg_destroy(scg_env* env) {
    if (!env) return SCG_OK;
    (void)hipSetDevice(env->device);
    if (env->d_goal) (void)hipFree(env->d_goal);
    if (env->d_params) (void)hipFree(env->d_params);
    delete env;
    return SCG_OK;
}

template <typename T>
static StepOut<T> typed_out(const scg_step_out* o) {
    StepOut<T> t{};
    if (!o) return t;
    t.obs = (T*)o->d_obs; t.reward = (T*)o->d_reward; t.done = o->d_done; t.flags = o->d_flags;
    t.c_values = (T*)o->d_c_values; t.mse = (T*)o->d_mse; t.terminal_obs = (T*)o->d_terminal_obs;
    t.state = (T*)o->d_state; t.noisy_action = (T*)o->d_noisy_action;
    t.ep_return = (T*)o->d_ep_return; t.ep_length = o->d_ep_length; t.ep_violation = (T*)o->d_ep_violation;
    t.ep_mse = (T*)o->d_ep_mse; t.fin_return = (T*)o->d_fin_return; t.fin_length = o->d_fin_length;
    t.fin_violation = (T*)o->d_fin_violation; t.fin_mse = (T*)o->d_fin_mse;
    return t;
}

#define DISPATCH_SYS_D(env, T, CALL)                                           \
    switch ((env)->cfg.system) {                                               \
        case SCG_CARTPOLE: { constexpr int S = SCG_CARTPOLE; CALL; } break;    \
        case SCG_QUAD_1D: { constexpr int S = SCG_QUAD_1D; CALL; } break;      \
        case SCG_QUAD_2D: { constexpr int S = SCG_QUAD_2D; CALL; } break;      \
        default: { constexpr int S = SCG_QUAD_3D; CALL; } break;               \
    }
// DIST kernel variant only when a disturbance or an adversary is configured
#define DISPATCH_SYS(env, T, CALL)                                             \
    if ((env)->has_dist) { constexpr bool DD = true; DISPATCH_SYS_D(env, T, CALL) } \
    else { constexpr bool DD = false; DISPATCH_SYS_D(env, T, CALL) }

template <typename T>
static int launch_reset(scg_env* env, const uint8_t* mask, const scg_step_out* out, hipStream_t st) {
    const int grid = (env->cfg.num_envs + BLOCK - 1) / BLOCK;
    StepOut<T> O = typed_out<T>(out);
    const DevParams<T>* P = (const DevParams<T>*)env->d_params;
    DISPATCH_SYS(env, T, (reset_kernel<S, T, DD><<<dim3(grid), dim3(BLOCK), env->lds_bytes, st>>>(P, mask, O)));
    HIP_TRY(hipGetLastError());
    return SCG_OK;
}

template <typename T>
static int launch_step(scg_env* env, const void* action, const void* adv, const scg_step_out* out, hipStream_t st) {
    const int grid = (env->cfg.num_envs + BLOCK - 1) / BLOCK;
    StepOut<T> O = typed_out<T>(out);
    const DevParams<T>* P = (const DevParams<T>*)env->d_params;
    DISPATCH_SYS(env, T, (step_kernel<S, T, DD><<<dim3(grid), dim3(BLOCK), env->lds_bytes, st>>>(P, (const T*)action, (const T*)adv, O)));
    HIP_TRY(hipGetLastError());
    return SCG_OK;
}

template <typename T>
static int launch_rollout(scg_env* env, int k, const scg_rollout_out* o, hipStream_t st) {
    const int grid = (env->cfg.num_envs + BLOCK - 1) / BLOCK;
    const DevParams<T>* P = (const DevParams<T>*)env->d_params;
    T* rs = o ? (T*)o->d_reward_sum : nullptr;
    int32_t* dc = o ? o->d_done_count : nullptr;
    int32_t* vc = o ? o->d_violation_count : nullptr;
    T* lo = o ? (T*)o->d_last_obs : nullptr;
    DISPATCH_SYS(env, T, (rollout_random_kernel<S, T, DD><<<dim3(grid), dim3(BLOCK), env->lds_bytes, st>>>(P, k, rs, dc, vc, lo)));
    HIP_TRY(hipGetLastError());
    return SCG_OK;
}

extern "C" int scg_reset(scg_env* env, const uint8_t* d_mask, const scg_step_out* out, void* stream) {
    if (!env) return fail(SCG_ERR_INVALID, "env is NULL");
    HIP_TRY(hipSetDevice(env->device));
    int rc = env->dtype == SCG_F64 ? launch_reset<double>(env, d_mask, out, (hipStream_t)stream)
                                   : launch_reset<float>(env, d_mask, out, (hipStream_t)stream);
    if (rc == SCG_OK && !d_mask) env->has_reset = true;
    return rc;
}

extern "C" int scg_step(scg_env* env, const void* d_action, const void* d_adv_action, const scg_step_out* out, void* stream) {
    if (!env) return fail(SCG_ERR_INVALID, "env is NULL");
    if (!d_action) return fail(SCG_ERR_INVALID, "d_action is NULL");
    // benchmark_env.py:230-235: "You must call env.reset() at least once before using env.step()."
    if (!env->has_reset) return fail(SCG_ERR_STATE, "scg_reset (all envs) must be called before scg_step");
    HIP_TRY(hipSetDevice(env->device));
    return env->dtype == SCG_F64 ? launch_step<double>(env, d_action, d_adv_action, out, (hipStream_t)stream)
                                 : launch_step<float>(env, d_action, d_adv_action, out, (hipStream_t)stream);
}

extern "C" int scg_rollout_random(scg_env* env, int k_steps, const scg_rollout_out* out, void* stream) {
    if (!env) return fail(SCG_ERR_INVALID, "env is NULL");
    if (k_steps <= 0) return fail(SCG_ERR_INVALID, "k_steps must be positive");
    if (!env->has_reset) return fail(SCG_ERR_STATE, "scg_reset (all envs) must be called before scg_rollout_random");
    HIP_TRY(hipSetDevice(env->device));
    return env->dtype == SCG_F64 ? launch_rollout<double>(env, k_steps, out, (hipStream_t)stream)
                                 : launch_rollout<float>(env, k_steps, out, (hipStream_t)stream);
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
    if (N >= 1024 || Tn < 64) {
        const int grid = (N + BLOCK - 1) / BLOCK;
        gae_env_kernel<T><<<dim3(grid), dim3(BLOCK), 0, st>>>((T*)rew, (const T*)v, (const T*)mask, (const T*)term,
                                                             (const T*)last, (T*)ret, (T*)adv, Tn, N, (T)gamma, (T)lam, use_gae);
    } else {
        gae_wave_kernel<T><<<dim3(N), dim3(64), 0, st>>>((T*)rew, (const T*)v, (const T*)mask, (const T*)term,
                                                        (const T*)last, (T*)ret, (T*)adv, Tn, N, (T)gamma, (T)lam, use_gae);
    }
    HIP_TRY(hipGetLastError());
    return SCG_OK;
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

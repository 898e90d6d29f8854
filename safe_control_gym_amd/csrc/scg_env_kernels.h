// scg_env_kernels.h — reset / step / fused-rollout kernels.
//
// Kernel geometry (CDNA4: 64-lane waves, 256 CUs in 8 XCDs, 160 KB LDS/CU):
//   * one thread = one environment; 256-thread workgroups (4 waves, one per SIMD of a CU);
//   * raw simulator state is SoA ([component][env]) so each wave's loads/stores are 256 contiguous bytes;
//   * the whole control step (action pre-processing, disturbance draws, PYB_FREQ/CTRL_FREQ integrator
//     substeps, observation/reward/done/info/constraints, episode statistics, auto-reset) is ONE launch;
//   * no MFMA: there is no dense contraction on this path, the bound is HBM bandwidth.
//
// Two builds of the same code:
//   generic  (libscg_hip.so): CfgParams<T> lives in device memory; every workgroup stages it and the
//            X_GOAL table into LDS (one 16-byte load per thread for the parameters, up to three for the
//            table, all in flight together with the per-env state loads, ONE wait), then reads from LDS.
//   SCG_SPEC (libscg_spec_<hash>.so, generated per task config by scg_spec.h): CfgParams<T> is a
//            `static constexpr` object — parameters are immediates, config branches are resolved at
//            compile time, the substep loop has a constant trip count; X_GOAL rows are read straight from
//            global memory (L2-resident) and there is no LDS, no barrier.
#pragma once
#include <hip/hip_runtime.h>

#include "scg_env_core.h"

namespace scg {

constexpr int BLOCK = 256;
constexpr size_t LDS_GOAL_LIMIT = 64 * 1024;

constexpr size_t lds16(size_t sz) { return (sz + 15) / 16 * 16; }
static_assert(sizeof(CfgParams<double>) <= 3 * 256 * 16, "CfgParams must fit three 16-byte loads per thread");

#ifdef SCG_SPEC
#define SCG_DEV_GOAL_TABLE 0
#else
#define SCG_DEV_GOAL_TABLE 1
#endif

struct StageRegs {
    uint4 c0, c1, c2, g0, g1, g2;
    int n_cfg16, n_goal16;
    const uint4* goal_src;
};

#ifndef SCG_SPEC
template <typename T>
__device__ __forceinline__ StageRegs stage_issue(const CfgParams<T>* __restrict__ Cg, const InstParams<T>& I) {
    StageRegs R;
    R.c0 = R.c1 = R.c2 = R.g0 = R.g1 = R.g2 = make_uint4(0, 0, 0, 0);
    R.n_cfg16 = (int)(lds16(sizeof(CfgParams<T>)) / 16);
    const uint4* src = reinterpret_cast<const uint4*>(Cg);
    const int t = (int)threadIdx.x;
    if (t < R.n_cfg16) R.c0 = src[t];
    if (t + BLOCK < R.n_cfg16) R.c1 = src[t + BLOCK];
    if (t + 2 * BLOCK < R.n_cfg16) R.c2 = src[t + 2 * BLOCK];
    R.goal_src = reinterpret_cast<const uint4*>(I.x_goal);
    R.n_goal16 = I.goal_lds16;
    if (t < R.n_goal16) R.g0 = R.goal_src[t];
    if (t + BLOCK < R.n_goal16) R.g1 = R.goal_src[t + BLOCK];
    if (t + 2 * BLOCK < R.n_goal16) R.g2 = R.goal_src[t + 2 * BLOCK];
    return R;
}

template <typename T>
__device__ __forceinline__ GoalTab<T> stage_commit(unsigned char* smem, const StageRegs& R, const InstParams<T>& I,
                                                   const CfgParams<T>*& cfg_lds) {
    const int t = (int)threadIdx.x;
    uint4* dst = reinterpret_cast<uint4*>(smem);
    if (t < R.n_cfg16) dst[t] = R.c0;
    if (t + BLOCK < R.n_cfg16) dst[t + BLOCK] = R.c1;
    if (t + 2 * BLOCK < R.n_cfg16) dst[t + 2 * BLOCK] = R.c2;
    uint4* gdst = dst + R.n_cfg16;
    if (t < R.n_goal16) gdst[t] = R.g0;
    if (t + BLOCK < R.n_goal16) gdst[t + BLOCK] = R.g1;
    if (t + 2 * BLOCK < R.n_goal16) gdst[t + 2 * BLOCK] = R.g2;
    for (int k = 3 * BLOCK + t; k < R.n_goal16; k += BLOCK) gdst[k] = R.goal_src[k];     // tables > 12 KB
    __syncthreads();
    cfg_lds = reinterpret_cast<const CfgParams<T>*>(smem);
    GoalTab<T> G;
    G.lds = reinterpret_cast<const T*>(gdst);
    G.glob = I.x_goal;
    G.in_lds = R.n_goal16 > 0;
    return G;
}
#endif

// SCG_PROLOGUE: issue every HBM request of this thread, then (generic build) commit the LDS staging.
// Leaves `P` (parameter view), `goal` (X_GOAL table), `live`.
#ifdef SCG_SPEC
#define SCG_CFG_REF(T) (scg_spec_cfg<T>())
#endif

template <int SYS, typename T, bool DIST>
__global__ __launch_bounds__(BLOCK) void reset_kernel(const CfgParams<T>* __restrict__ Cg, const InstParams<T> I,
                                                      const uint8_t* __restrict__ mask, StepOut<T> O) {
    using Ops = EnvOps<SYS, T, DIST>;
    using D = Dims<SYS>;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = I.num_envs;
    const bool live = i < N && (!mask || mask[i]);
    typename Ops::E e;
#ifdef SCG_SPEC
    constexpr CfgParams<T> kcfg = scg_make_spec_cfg<T>();     // compile-time constants (see scg_spec.h)
    const PV<T> P{kcfg, I};
    const GoalTab<T> goal{nullptr, I.x_goal, false};
    if (!live) return;
    Ops::load_state(P, i, e);
#else
    extern __shared__ __align__(16) unsigned char smem[];
    const StageRegs SR = stage_issue(Cg, I);
    {
        const PV<T> Pg{*Cg, I};
        if (live) Ops::load_state(Pg, i, e);
    }
    const CfgParams<T>* cl;
    const GoalTab<T> goal = stage_commit<T>(smem, SR, I, cl);
    if (!live) return;
    const PV<T> P{*cl, I};
#endif
    Ops::load_params(P, i, e);
    const RngKey key{I.key0, I.key1};
    Ops::reset(P, i, e, key);
    T st[D::NX];
    Ops::state_vector(e, st);
    if (O.obs) Ops::write_obs(P, goal, st, e, key, 1, 0u, 0, i, O.obs + (size_t)i * P.c.nobs);
    if (O.c_values && P.c.n_state_con_rows > 0) Ops::constraints(P, st, st, O.c_values + i, (size_t)N, true);
    if (O.state) {
#pragma unroll
        for (int k = 0; k < D::NX; ++k) O.state[(size_t)k * N + i] = st[k];
    }
    if (O.ep_return) O.ep_return[i] = (T)0;
    if (O.ep_length) O.ep_length[i] = 0;
    if (O.ep_violation) O.ep_violation[i] = (T)0;
    if (O.ep_mse) O.ep_mse[i] = (T)0;
    if (I.oob_attr) I.oob_attr[i] = 0;
    Ops::store(P, i, e, true);
}

// Per-thread output addresses, computed at kernel entry so that every kernarg (pointer) is fetched in the
// first scalar-load round instead of lazily at its first use deep inside the kernel.
template <typename T>
struct OutPtrs {
    T* obs; T* reward; uint8_t* done; uint8_t* flags; T* c_values; T* mse; T* terminal_obs; T* state;
    T* noisy_action; T* ep_return; int32_t* ep_length; T* ep_violation; T* ep_mse;
    T* fin_return; int32_t* fin_length; T* fin_violation; T* fin_mse;
};

template <typename T>
__device__ __forceinline__ OutPtrs<T> out_ptrs(const StepOut<T>& O, int i, int nobs) {
    OutPtrs<T> p;
    p.obs = O.obs ? O.obs + (size_t)i * nobs : nullptr;
    p.terminal_obs = O.terminal_obs ? O.terminal_obs + (size_t)i * nobs : nullptr;
    p.reward = O.reward ? O.reward + i : nullptr;
    p.done = O.done ? O.done + i : nullptr;
    p.flags = O.flags ? O.flags + i : nullptr;
    p.c_values = O.c_values ? O.c_values + i : nullptr;
    p.mse = O.mse ? O.mse + i : nullptr;
    p.state = O.state ? O.state + i : nullptr;
    p.noisy_action = O.noisy_action ? O.noisy_action + i : nullptr;
    p.ep_return = O.ep_return ? O.ep_return + i : nullptr;
    p.ep_length = O.ep_length ? O.ep_length + i : nullptr;
    p.ep_violation = O.ep_violation ? O.ep_violation + i : nullptr;
    p.ep_mse = O.ep_mse ? O.ep_mse + i : nullptr;
    p.fin_return = O.fin_return ? O.fin_return + i : nullptr;
    p.fin_length = O.fin_length ? O.fin_length + i : nullptr;
    p.fin_violation = O.fin_violation ? O.fin_violation + i : nullptr;
    p.fin_mse = O.fin_mse ? O.fin_mse + i : nullptr;
    return p;
}

template <typename T>
__device__ __forceinline__ void fence_kernargs(const InstParams<T>& I, const T* action, const T* adv, const StepOut<T>& O) {
    sreg_fence(I.cold); sreg_fence(I.x_goal); sreg_fence(I.state); sreg_fence(I.param); sreg_fence(I.step); sreg_fence(I.episode);
    sreg_fence(I.oob_attr); sreg_fence(I.num_envs); sreg_fence(I.env_id_offset); sreg_fence(I.key0); sreg_fence(I.key1);
    sreg_fence(action); sreg_fence(adv);
    sreg_fence(O.obs); sreg_fence(O.reward); sreg_fence(O.done); sreg_fence(O.flags); sreg_fence(O.c_values);
    sreg_fence(O.mse); sreg_fence(O.terminal_obs); sreg_fence(O.state); sreg_fence(O.noisy_action);
    sreg_fence(O.ep_return); sreg_fence(O.ep_length); sreg_fence(O.ep_violation); sreg_fence(O.ep_mse);
    sreg_fence(O.fin_return); sreg_fence(O.fin_length); sreg_fence(O.fin_violation); sreg_fence(O.fin_mse);
}

template <int SYS, typename T, bool DIST>
__global__ __launch_bounds__(BLOCK) void step_kernel(const CfgParams<T>* __restrict__ Cg, const InstParams<T> I,
                                                     const T* __restrict__ action, const T* __restrict__ adv, StepOut<T> O) {
    using Ops = EnvOps<SYS, T, DIST>;
    using D = Dims<SYS>;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const int N = I.num_envs;
    const bool live = i < N;
    // ---- memory round 1: kernargs — every pointer is fetched in this block, one scalar-memory round
    fence_kernargs(I, action, adv, O);
    typename Ops::E e;
    T act[D::NU];
    T ep_ret = (T)0, ep_viol = (T)0, ep_mse = (T)0;
    int32_t ep_len = 0;
#ifdef SCG_SPEC
    constexpr CfgParams<T> kcfg = scg_make_spec_cfg<T>();     // compile-time constants (see scg_spec.h)
    const PV<T> P{kcfg, I};
    const GoalTab<T> goal{nullptr, I.x_goal, false};
    if (!live) return;
    const PV<T>& Pg = P;
    const int nobs_early = kcfg.nobs;
#else
    extern __shared__ __align__(16) unsigned char smem[];
    const StageRegs SR = stage_issue(Cg, I);
    const PV<T> Pg{*Cg, I};
    const int nobs_early = D::NX * (1 + I.obs_ext_rows);
#endif
    const OutPtrs<T> Q = out_ptrs(O, live ? i : 0, nobs_early);
    // ---- memory round 2: everything this thread needs from HBM, requested before the single wait
    if (live) {
        Ops::load_state(Pg, i, e);
#pragma unroll
        for (int j = 0; j < D::NU; ++j) act[j] = action[(size_t)i * D::NU + j];
        // unconditional loads (an unbound accumulator reads a valid dummy address): a branch per pointer
        // would split the requests over several dependent rounds
        const T* dummy_t = I.state + i;
        ep_ret = *(Q.ep_return ? Q.ep_return : dummy_t);
        ep_len = *(Q.ep_length ? Q.ep_length : I.step + i);
        ep_viol = *(Q.ep_violation ? Q.ep_violation : dummy_t);
        ep_mse = *(Q.ep_mse ? Q.ep_mse : dummy_t);
    }
#ifndef SCG_SPEC
    const CfgParams<T>* cl;
    const GoalTab<T> goal = stage_commit<T>(smem, SR, I, cl);
    if (!live) return;
    const PV<T> P{*cl, I};
#endif
    Ops::load_params(P, i, e);
    const RngKey key{I.key0, I.key1};
    const int32_t c0 = e.step;
    // ---- memory round 3 (overlapped with the integrator): reference rows of X_GOAL for this step
    const bool pre_rows = P.c.task == SCG_TASK_TRAJ_TRACKING;
    const bool pre_ext = pre_rows && P.c.cost == SCG_COST_RL_REWARD && P.c.obs_goal_horizon == 1;
    T ref_pre[D::NX], ext_pre[D::NX], ext_reset[D::NX];
    if (pre_rows) {
        const int last = P.c.goal_rows - 1;
        int r1 = c0 + 1; r1 = r1 > last ? last : r1;
#pragma unroll
        for (int k = 0; k < D::NX; ++k) ref_pre[k] = goal[r1 * D::NX + k];
        if (pre_ext) {
            int r2 = c0 + 2; r2 = r2 > last ? last : r2;
            const int r0 = 1 > last ? last : 1;
#pragma unroll
            for (int k = 0; k < D::NX; ++k) ext_pre[k] = goal[r2 * D::NX + k];
#pragma unroll
            for (int k = 0; k < D::NX; ++k) ext_reset[k] = goal[r0 * D::NX + k];
        }
    }
    T advv[D::DYN > D::NU ? D::DYN : D::NU];
    const T* advp = nullptr;
    if constexpr (DIST) {
        if (adv && P.c.adversary_channel >= 0) {
            const int ad = P.c.adversary_channel == SCG_CH_ACTION ? D::NU : D::DYN;
            for (int j = 0; j < ad; ++j) advv[j] = adv[(size_t)i * ad + j];
            advp = advv;
        }
    }
    T st[D::NX], noisy[D::NU];
    typename Ops::StepResult r = Ops::step(P, goal, e, act, advp, key, i, st, noisy, Q.c_values, (size_t)N,
                                           pre_rows ? ref_pre : nullptr, pre_ext ? ext_pre : nullptr,
                                           pre_ext ? ext_reset : nullptr);
    if (Q.reward) *Q.reward = r.reward;
    if (Q.done) *Q.done = r.done ? 1 : 0;
    if (Q.flags) *Q.flags = r.flags;
    if (Q.mse) *Q.mse = r.mse;
    if (Q.noisy_action) {
#pragma unroll
        for (int j = 0; j < D::NU; ++j) Q.noisy_action[(size_t)j * N] = noisy[j];
    }
    // columnar VecRecordEpisodeStatistics (record_episode_statistics.py:139-166)
    if (Q.ep_return) {
        const T acc = ep_ret + r.reward;
        if (r.done && Q.fin_return) *Q.fin_return = acc;
        *Q.ep_return = r.done ? (T)0 : acc;
    }
    if (Q.ep_length) {
        const int32_t acc = ep_len + 1;
        if (r.done && Q.fin_length) *Q.fin_length = acc;
        *Q.ep_length = r.done ? 0 : acc;
    }
    if (Q.ep_violation) {
        const T acc = ep_viol + ((r.flags & FLAG_VIOLATION) ? (T)1 : (T)0);
        if (r.done && Q.fin_violation) *Q.fin_violation = acc;
        *Q.ep_violation = r.done ? (T)0 : acc;
    }
    if (Q.ep_mse) {
        const T acc = ep_mse + r.mse;
        if (r.done && Q.fin_mse) *Q.fin_mse = acc;
        *Q.ep_mse = r.done ? (T)0 : acc;
    }
    // observation of the step: goes to terminal_observation where the env is about to auto-reset, else it is the
    // returned obs (two write_obs call sites only: the disturbance code is inlined into each)
    const bool do_reset = r.done && P.c.auto_reset;
    {
        T* dst = do_reset ? Q.terminal_obs : Q.obs;
        if (dst) Ops::write_obs(P, goal, st, e, key, c0 + 2, (uint32_t)(c0 + 1), c0, i, dst, pre_ext ? ext_pre : nullptr);
        if (r.done && !P.c.auto_reset && Q.terminal_obs && Q.obs) {
            // single-env semantics (BenchmarkEnv.step): the terminal observation is also the returned one
            for (int k = 0; k < P.c.nobs; ++k) Q.terminal_obs[k] = Q.obs[k];
        }
    }
    if (do_reset) {
        Ops::reset(P, i, e, key);               // auto-reset (dummy_vec_env.py:33-38)
        Ops::state_vector(e, st);
        if (Q.obs) Ops::write_obs(P, goal, st, e, key, 1, 0u, 0, i, Q.obs, pre_ext ? ext_reset : nullptr);
    }
    if (Q.state) {
#pragma unroll
        for (int k = 0; k < D::NX; ++k) Q.state[(size_t)k * N] = st[k];
    }
    Ops::store(P, i, e, do_reset);
}

template <int SYS, typename T, bool DIST>
__global__ __launch_bounds__(BLOCK) void rollout_random_kernel(const CfgParams<T>* __restrict__ Cg, const InstParams<T> I,
                                                               int k_steps, T* __restrict__ reward_sum,
                                                               int32_t* __restrict__ done_count,
                                                               int32_t* __restrict__ violation_count, T* __restrict__ last_obs) {
    using Ops = EnvOps<SYS, T, DIST>;
    using D = Dims<SYS>;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < I.num_envs;
    typename Ops::E e;
#ifdef SCG_SPEC
    constexpr CfgParams<T> kcfg = scg_make_spec_cfg<T>();     // compile-time constants (see scg_spec.h)
    const PV<T> P{kcfg, I};
    const GoalTab<T> goal{nullptr, I.x_goal, false};
    if (!live) return;
    Ops::load_state(P, i, e);
#else
    extern __shared__ __align__(16) unsigned char smem[];
    const StageRegs SR = stage_issue(Cg, I);
    {
        const PV<T> Pg{*Cg, I};
        if (live) Ops::load_state(Pg, i, e);
    }
    const CfgParams<T>* cl;
    const GoalTab<T> goal = stage_commit<T>(smem, SR, I, cl);
    if (!live) return;
    const PV<T> P{*cl, I};
#endif
    Ops::load_params(P, i, e);
    const RngKey key{I.key0, I.key1};
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
            if (P.c.auto_reset) {
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
                       last_obs + (size_t)i * P.c.nobs);
    }
    Ops::store(P, i, e, dirty);
}

}  // namespace scg

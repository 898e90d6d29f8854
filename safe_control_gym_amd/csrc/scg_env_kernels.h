// scg_env_kernels.h — reset / step / fused-rollout kernels.
//
// Kernel geometry (CDNA4: 64-lane waves, 256 CUs in 8 XCDs, 160 KB LDS/CU):
//   * one thread = one environment; workgroups of one wave (specialised build) or four (generic build, LDS staging);
//   * raw simulator state is SoA ([component][env]) so each wave's loads/stores are 256 contiguous bytes;
//   * the whole control step (action pre-processing, disturbance draws, PYB_FREQ/CTRL_FREQ integrator
//     substeps, observation/reward/done/info/constraints, episode statistics, auto-reset) is ONE launch;
//   * no MFMA: there is no dense contraction on this path, the bound is HBM bandwidth.
//
// Two builds of the same code:
//   generic  (libscg_hip.so): CfgParams<T> lives in device memory; every workgroup stages it and the
//            X_GOAL table into LDS (one 16-byte load per thread for the parameters, up to three for the
//            table, all in flight together with the per-env state loads, ONE wait), then reads from LDS.
//   SCG_SPEC (libscg_spec_<hash>.so, generated per task config by scg_spec.h): CfgParams<T> is a constexpr
//            object — parameters are immediates, config branches are resolved at compile time, the substep
//            loop is straight-line code; X_GOAL rows are read straight from global memory (L2-resident); LDS
//            is used only for the wave-private transpose of the observation rows (no workgroup barrier).
#pragma once
#include <hip/hip_runtime.h>

#include "scg_env_core.h"
#if defined(SCG_SPEC) && defined(SCG_POLICY_H)
#include "scg_mlp.h"
#endif

namespace scg {

// Workgroup size.  Specialised build: one wave per workgroup — nothing is shared between waves (the observation
// transpose is wave-private), and single-wave workgroups dispatch and balance better: 5.73 -> 5.62 us per launch at
// 65 536 envs (tools/exp_variants.sh).  Generic build: 256 threads stage the parameter
// block and the X_GOAL table into LDS cooperatively.
#ifndef SCG_BLOCK
#ifdef SCG_SPEC
#define SCG_BLOCK 64
#else
#define SCG_BLOCK 256
#endif
#endif
constexpr int BLOCK = SCG_BLOCK;
constexpr size_t LDS_GOAL_LIMIT = 64 * 1024;

constexpr size_t lds16(size_t sz) { return (sz + 15) / 16 * 16; }
static_assert(sizeof(CfgParams<double>) <= 3 * 256 * 16, "CfgParams must fit three 16-byte loads per thread");
static_assert(offsetof(CfgParams<double>, dist) + sizeof(HotDist<double>) * 3 * SCG_MAX_DISTURB == sizeof(CfgParams<double>), "dist must be the last member of CfgParams");

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
// bytes of CfgParams a kernel stages into LDS: kernels without disturbances stop before the disturbance table
template <typename T, bool DIST>
constexpr size_t cfg_stage_bytes() { return lds16(DIST ? sizeof(CfgParams<T>) : offsetof(CfgParams<T>, dist)); }

template <typename T, bool DIST>
__device__ __forceinline__ StageRegs stage_issue(const CfgParams<T>* __restrict__ Cg, const InstParams<T>& I) {
    StageRegs R;
    R.c0 = R.c1 = R.c2 = R.g0 = R.g1 = R.g2 = make_uint4(0, 0, 0, 0);
    R.n_cfg16 = (int)(cfg_stage_bytes<T, DIST>() / 16);
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

// Per-lane output slots, built at kernel entry (see Slot / OutTab).
template <typename T>
struct OutPtrs {
    Slot<T> obs, reward; Slot<uint8_t> done, flags; Slot<T> c_values, mse, terminal_obs, state, noisy_action, ep_stats, fin_stats;
};

template <typename V>
__device__ __forceinline__ Slot<V> out_slot(const OutTabOne& O, int k, int i, int elems = 1) {
    return slot_in<V>(make_rsrc(O.base), O.off[k], i, elems);
}
template <typename V>
__device__ __forceinline__ Slot<V> out_slot(const OutTabPtr& O, int k, int i, int elems = 1) {
    return slot(reinterpret_cast<V*>(O.ptr[k]), i, elems);
}

template <typename T, typename TAB>
__device__ __forceinline__ OutPtrs<T> out_ptrs(const TAB& O, int i, int nobs) {
    OutPtrs<T> p;
    p.obs = out_slot<T>(O, OUT_OBS, i, nobs);
    p.terminal_obs = out_slot<T>(O, OUT_TERMINAL_OBS, i, nobs);
    p.reward = out_slot<T>(O, OUT_REWARD, i);
    p.done = out_slot<uint8_t>(O, OUT_DONE, i);
    p.flags = out_slot<uint8_t>(O, OUT_FLAGS, i);
    p.c_values = out_slot<T>(O, OUT_C_VALUES, i);
    p.mse = out_slot<T>(O, OUT_MSE, i);
    p.state = out_slot<T>(O, OUT_STATE, i);
    p.noisy_action = out_slot<T>(O, OUT_NOISY_ACTION, i);
    p.ep_stats = out_slot<T>(O, OUT_EP_STATS, i, 4);
    p.fin_stats = out_slot<T>(O, OUT_FIN_STATS, i, 4);
    return p;
}

template <int SYS, typename T, bool DIST>
__global__ __launch_bounds__(BLOCK) void reset_kernel(const CfgParams<T>* __restrict__ Cg, const InstParams<T> I,
                                                      const uint8_t* __restrict__ mask, const OutTabPtr OT) {
    using Ops = EnvOps<SYS, T, DIST>;
    using D = Dims<SYS>;
    const int i = I.env_first + blockIdx.x * blockDim.x + threadIdx.x;
    const int N = I.num_envs;
    const bool live = i < I.env_end && (!mask || mask[i]);
    typename Ops::E e;
#ifdef SCG_SPEC
    constexpr CfgParams<T> kcfg = scg_make_spec_cfg<T>();     // compile-time constants (see scg_spec.h)
    const PV<T> P{kcfg, I};
    const GoalTab<T> goal{nullptr, I.x_goal, false};
    if (!live) return;
    Ops::load_state(P, i, e);
#else
    extern __shared__ __align__(16) unsigned char smem[];
    const StageRegs SR = stage_issue<T, DIST>(Cg, I);
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
    T st[D::NX];
    Ops::reset(P, i, e, key, st);
    const OutPtrs<T> Q = out_ptrs<T>(OT, i, P.c.nobs);
    if (Q.obs) Ops::write_obs(P, goal, st, e, key, 1, 0u, 0, i, Q.obs);
    if (Q.c_values && P.c.n_state_con_rows > 0) Ops::constraints(P, st, st, Q.c_values, (size_t)N, true);
    if (Q.state) {
#pragma unroll
        for (int k = 0; k < D::NX; ++k) Q.state.store(st[k], (size_t)k * N);
    }
    if (Q.ep_stats) {
        const T zero4[4] = {(T)0, (T)0, (T)0, (T)0};
        Q.ep_stats.template store_row<4>(zero4);
    }
    if (I.oob_off != SCG_NO_OFF) slot_in<uint8_t>(make_rsrc(I.ws), I.oob_off, i).store(0);
    Ops::store(P, i, e, true);
}

// Observation rows of one full wave -> memory as fully coalesced 16-byte stores.  Each lane holds the row of its own
// env (NROW elements, RB = row bytes, a multiple of 16); stored lane-by-lane that is a 16-byte piece every RB bytes —
// 64 partial cache lines per instruction, which the write path handles at a fraction of the speed of full lines
// (profiles/r01_latency_budget.md §3).  The wave's rows form one contiguous 64*RB block, so the rows are transposed through
// LDS (wave-private region, no workgroup barrier) and piece p of the block is written by lane p % 64.
template <typename T, int NROW, int AUX>
__device__ __forceinline__ void store_rows_coalesced(const Slot<T, AUX>& dst, const T* row, unsigned char* lds_wave, int lane) {
    constexpr int RB = NROW * (int)sizeof(T);
    static_assert(RB % 16 == 0, "row pitch must be a multiple of 16 bytes");
    constexpr int per = 16 / (int)sizeof(T);
    struct Piece { T e[per]; };
    if constexpr (RB == 16) {           // a 16-byte row per lane IS the contiguous block (CartPole's float obs): no transpose to do
        Piece p;
#pragma unroll
        for (int j = 0; j < per; ++j) p.e[j] = row[j];
        buf_st128<AUX>(__builtin_bit_cast(u32x4, p), dst.r, dst.off, dst.soff, 0u);
        return;
    }
#pragma unroll
    for (int c = 0; c < RB / 16; ++c) {
        Piece p;
#pragma unroll
        for (int j = 0; j < per; ++j) p.e[j] = row[c * per + j];
        *reinterpret_cast<Piece*>(lds_wave + lane * RB + c * 16) = p;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t block_off = dst.off - (uint32_t)(lane * RB);      // byte offset of the wave's first row
#pragma unroll
    for (int c = 0; c < RB / 16; ++c) {
        const Piece p = *reinterpret_cast<const Piece*>(lds_wave + (c * 64 + lane) * 16);
        buf_st128<AUX>(__builtin_bit_cast(u32x4, p), dst.r, block_off + (uint32_t)(lane * 16), dst.soff, (uint32_t)(c * 1024));
    }
}

template <typename T>
__device__ __forceinline__ void fence_out(const OutTabOne& O) {
    sreg_fence(O.base);
#pragma unroll
    for (int k = 0; k < OUT_COUNT; ++k) sreg_fence(O.off[k]);
}
template <typename T>
__device__ __forceinline__ void fence_out(const OutTabPtr& O) {
#pragma unroll
    for (int k = 0; k < OUT_COUNT; ++k) sreg_fence(O.ptr[k]);
}
template <typename T, typename TAB>
__device__ __forceinline__ void fence_kernargs(const InstParams<T>& I, const T* action, const T* adv, const TAB& O) {
    sreg_fence(I.cold); sreg_fence(I.x_goal); sreg_fence(I.ws); sreg_fence(I.state_off); sreg_fence(I.param_off);
    sreg_fence(I.step_off); sreg_fence(I.episode_off); sreg_fence(I.oob_off); sreg_fence(I.dist_off); sreg_fence(I.num_envs);
    sreg_fence(I.env_first); sreg_fence(I.env_end);
    sreg_fence(I.env_id_offset); sreg_fence(I.key0); sreg_fence(I.key1);
    sreg_fence(action); sreg_fence(adv);
    fence_out<T>(O);
}

// (Rounds 5-6 carried a SPLIT launch here — two waves per 64 envs, one producing reward / done / constraint rows / statistics, the other
//  observation / auto-reset / state.  As independent workgroups it won 5-7.5 % on shards <= 32 768 envs but relied on dispatch order:
//  the state wave overwrites the workspace the score wave reads.  Ordered properly — the two waves in one workgroup behind one
//  s_barrier — it LOSES: 3.37 vs 3.22 us at 16 384 envs, 3.80 vs 3.40 us at 32 768 (profiles/r06_split_wsback_ab.txt).  Removed.)
// BLK: threads per workgroup of the kernel that inlines this body (BLOCK, or WIDE_BLOCK for step_wide_kernel).
// WSWB: the workspace arrays are stored write-back (EnvOps' WSAUX = 0) instead of write-through — step_wsback_kernel.
template <int SYS, typename T, bool DIST, bool ONE, int BLK = BLOCK, bool WSWB = false>
__device__ __forceinline__ void step_body(const CfgParams<T>* __restrict__ Cg, const InstParams<T>& I,
                                          const T* __restrict__ action, const T* __restrict__ adv,
                                          const typename OutTabOf<ONE>::type& O, const int wg) {
    using Ops = EnvOps<SYS, T, DIST, SCG_ST_AUX, WSWB ? 0 : SCG_ST_AUX>;
    using D = Dims<SYS>;
    const int tid = (int)threadIdx.x;
    const int i = I.env_first + wg * BLK + tid;
    const int N = I.num_envs;
    const bool live = i < I.env_end;
    // ---- memory round 1: kernargs — every pointer is fetched in this block, one scalar-memory round
    SCG_TL(0);
    fence_kernargs<T>(I, action, adv, O);
    SCG_TL(1);
    typename Ops::E e;
    T act[D::NU];
    T ep[4] = {(T)0, (T)0, (T)0, (T)0};      // running (return, length, violations, mse) of the episode
#ifdef SCG_SPEC
    constexpr CfgParams<T> kcfg = scg_make_spec_cfg<T>();     // compile-time constants (see scg_spec.h)
    const PV<T> P{kcfg, I};
    const GoalTab<T> goal{nullptr, I.x_goal, false};
    if (!live) return;
    const bool full_wave = __builtin_amdgcn_read_exec() == ~0ull;      // tail waves keep the per-lane store paths
    const PV<T>& Pg = P;
    const int nobs_early = kcfg.nobs;
#else
    extern __shared__ __align__(16) unsigned char smem[];
    const StageRegs SR = stage_issue<T, DIST>(Cg, I);
    const PV<T> Pg{*Cg, I};
    const int nobs_early = D::NX * (1 + I.obs_ext_rows);
#endif
    OutPtrs<T> Q = out_ptrs<T>(O, live ? i : 0, nobs_early);
#ifdef SCG_EXP_NO_CVAL
    Q.c_values.soff = SCG_NO_OFF;
#endif
#ifdef SCG_EXP_NO_OBS
    Q.obs.soff = SCG_NO_OFF; Q.terminal_obs.soff = SCG_NO_OFF;
#endif
    // ---- memory round 2: everything this thread needs from HBM, requested before the single wait
    if (live) {
        Ops::load_state(Pg, i, e);
#pragma unroll
        for (int j = 0; j < D::NU; ++j) act[j] = action[(size_t)i * D::NU + j];     // caller's tensor: plain global load
        // unconditional load (an unbound accumulator reads a valid dummy address): a branch would split the
        // requests over two dependent rounds
        (Q.ep_stats ? Q.ep_stats : slot_in<T>(make_rsrc(I.ws), I.state_off, 0, 4)).template load_row<4>(ep);
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
#ifdef SCG_EXP_TIMELINE
#pragma unroll
    for (int k = 0; k < D::NS; ++k) vreg_fence(e.s[k]);
    vreg_fence(act[0]); vreg_fence(ep[3]); vreg_fence(e.episode);
#endif
    SCG_TL(2);
    // ---- memory round 3 (overlapped with the integrator): reference rows of X_GOAL for this step
    const bool pre_rows = P.c.task == SCG_TASK_TRAJ_TRACKING;
    const bool pre_ref = pre_rows;                      // the reference row feeds reward / mse only when tracking
    const bool pre_ext = pre_rows && P.c.cost == SCG_COST_RL_REWARD && P.c.obs_goal_horizon == 1;
    T ref_pre[D::NX], ext_pre[D::NX], ext_reset[D::NX];
    if (pre_rows) {
        const int last = P.c.goal_rows - 1;
        int r1 = c0 + 1; r1 = r1 > last ? last : r1;
#ifdef SCG_EXP_UNIFORM_GOAL
        r1 = 1;
#endif
        if (pre_ref) {
#pragma unroll
            for (int k = 0; k < D::NX; ++k) ref_pre[k] = goal[r1 * D::NX + k];
        }
        if (pre_ext) {
            int r2 = c0 + 2; r2 = r2 > last ? last : r2;
#ifdef SCG_EXP_UNIFORM_GOAL
            r2 = 2;
#endif
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
                                           pre_ref ? ref_pre : nullptr, pre_ext ? ext_pre : nullptr,
                                           pre_ext ? ext_reset : nullptr);
    Q.reward.store(r.reward);           // obs / reward / done / flags are always bound (checked by scg_step)
    Q.done.store(r.done ? 1 : 0);
    Q.flags.store(r.flags);
    if (Q.mse) Q.mse.store(r.mse);
    if (Q.noisy_action) {
#pragma unroll
        for (int j = 0; j < D::NU; ++j) Q.noisy_action.store(noisy[j], (size_t)j * N);
    }
    // columnar VecRecordEpisodeStatistics (record_episode_statistics.py:139-166)
    if (Q.ep_stats) {
        ep[0] += r.reward;
        ep[1] += (T)1;
        ep[2] += (r.flags & FLAG_VIOLATION) ? (T)1 : (T)0;
        ep[3] += r.mse;
        if (r.done && Q.fin_stats) Q.fin_stats.template store_row<4>(ep);
        const T nxt[4] = {r.done ? (T)0 : ep[0], r.done ? (T)0 : ep[1], r.done ? (T)0 : ep[2], r.done ? (T)0 : ep[3]};
        Q.ep_stats.template store_row<4>(nxt);
    }
    // observation of the step: goes to terminal_observation where the env is about to auto-reset, else it is the
    // returned obs (two write_obs call sites only: the disturbance code is inlined into each)
#ifdef SCG_EXP_NO_RESET
    const bool do_reset = false;
#else
    const bool do_reset = r.done && P.c.auto_reset;
#endif
    if (Ops::obs_is_row(P)) {
        // One register row per env: the step's observation; a lane that auto-resets copies it to
        // terminal_observation and replaces it by the observation of the fresh episode; then ONE store of the rows.
        T row[2 * D::NX];
        int nrow = Ops::obs_row(P, goal, st, e, key, c0 + 2, (uint32_t)(c0 + 1), c0, i, pre_ext ? ext_pre : nullptr, row);
        if (r.done && Q.terminal_obs) Ops::store_obs_row(P, row, nrow, Q.terminal_obs);   // (also the non-auto-reset case)
        SCG_TL(6);
        if (do_reset) {
            Ops::reset(P, i, e, key, st);           // auto-reset (dummy_vec_env.py:33-38)
            nrow = Ops::obs_row(P, goal, st, e, key, 1, 0u, 0, i, pre_ext ? ext_reset : nullptr, row);
        }
#ifdef SCG_SPEC
        constexpr int kNobs = kcfg.nobs;
        constexpr bool can_transpose = (kNobs * (int)sizeof(T)) % 16 == 0 && (kNobs == D::NX || kNobs == 2 * D::NX);
        if constexpr (can_transpose) {
            __shared__ __align__(16) unsigned char s_obs[BLK * kNobs * sizeof(T)];
            const int lane = tid & 63;
            if (full_wave) store_rows_coalesced<T, kNobs>(Q.obs, row, s_obs + (tid >> 6) * (64 * kNobs * (int)sizeof(T)), lane);
            else Ops::store_obs_row(P, row, nrow, Q.obs);
        } else {
            Ops::store_obs_row(P, row, nrow, Q.obs);
        }
#else
        Ops::store_obs_row(P, row, nrow, Q.obs);
#endif
    } else {
        Ops::write_obs(P, goal, st, e, key, c0 + 2, (uint32_t)(c0 + 1), c0, i, Q.obs, nullptr);
        if (r.done && Q.terminal_obs)
            Ops::write_obs(P, goal, st, e, key, c0 + 2, (uint32_t)(c0 + 1), c0, i, Q.terminal_obs, nullptr);
        SCG_TL(6);
        if (do_reset) {
            Ops::reset(P, i, e, key, st);
            Ops::write_obs(P, goal, st, e, key, 1, 0u, 0, i, Q.obs, nullptr);
        }
    }
    if (Q.state) {
#pragma unroll
        for (int k = 0; k < D::NX; ++k) Q.state.store(st[k], (size_t)k * N);
    }
    Ops::store(P, i, e, do_reset);
    SCG_TL(7);
}

template <int SYS, typename T, bool DIST, bool ONE>
__global__ __launch_bounds__(BLOCK) void step_kernel(const CfgParams<T>* __restrict__ Cg, const InstParams<T> I,
                                                     const T* __restrict__ action, const T* __restrict__ adv,
                                                     const typename OutTabOf<ONE>::type O) {
    step_body<SYS, T, DIST, ONE>(Cg, I, action, adv, O, (int)blockIdx.x);
}

#ifdef SCG_SPEC
// Wide workgroups for the largest shards (>= SCG_WIDE_MIN_ENVS envs): the four waves of a 256-thread workgroup write 1 KB contiguous
// per output array and 4 x fewer workgroups are dispatched — 1018.5 -> 953.7 us per launch at 16 777 216 envs (-6.4 %; 512 / 1024
// threads the same), nothing reproducible at <= 4 M envs (+-3 % run to run), where the one-wave workgroups stay
// (profiles/r05_step_kernel_ab.md).
constexpr int WIDE_BLOCK = 256;
template <int SYS, typename T, bool DIST, bool ONE>
__global__ __launch_bounds__(WIDE_BLOCK) void step_wide_kernel(const CfgParams<T>* __restrict__ Cg, const InstParams<T> I,
                                                               const T* __restrict__ action, const T* __restrict__ adv,
                                                               const typename OutTabOf<ONE>::type O) {
    step_body<SYS, T, DIST, ONE, WIDE_BLOCK>(Cg, I, action, adv, O, (int)blockIdx.x);
}

// Mid-size shards (SCG_WSBACK_MIN_ENVS <= envs <= SCG_WSBACK_MAX_ENVS, Quadrotor systems): the same one-wave workgroups with the
// workspace arrays written BACK (see EnvOps' WSAUX) — everything the caller sees is still written through.
template <int SYS, typename T, bool DIST, bool ONE>
__global__ __launch_bounds__(BLOCK) void step_wsback_kernel(const CfgParams<T>* __restrict__ Cg, const InstParams<T> I,
                                                            const T* __restrict__ action, const T* __restrict__ adv,
                                                            const typename OutTabOf<ONE>::type O) {
    step_body<SYS, T, DIST, ONE, BLOCK, true>(Cg, I, action, adv, O, (int)blockIdx.x);
}

#endif

// Output slots of the K-steps-per-launch kernels: write-back stores (SCG_SEQ_ST_AUX, see Slot in scg_env_core.h).
template <typename V>
__device__ __forceinline__ Slot<V, SCG_SEQ_ST_AUX> seq_slot(V* base, int lane_index, int elems_per_lane = 1) {
    return slot(base, lane_index, elems_per_lane).template with<SCG_SEQ_ST_AUX>();
}

template <int SYS, typename T, bool DIST>
__global__ __launch_bounds__(BLOCK) void rollout_random_kernel(const CfgParams<T>* __restrict__ Cg, const InstParams<T> I,
                                                               int k_steps, T* __restrict__ reward_sum,
                                                               int32_t* __restrict__ done_count,
                                                               int32_t* __restrict__ violation_count, T* __restrict__ last_obs) {
    using Ops = EnvOps<SYS, T, DIST, SCG_SEQ_ST_AUX>;
    using D = Dims<SYS>;
    const int i = I.env_first + blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < I.env_end;
    typename Ops::E e;
#ifdef SCG_SPEC
    constexpr CfgParams<T> kcfg = scg_make_spec_cfg<T>();     // compile-time constants (see scg_spec.h)
    const PV<T> P{kcfg, I};
    const GoalTab<T> goal{nullptr, I.x_goal, false};
    if (!live) return;
    Ops::load_state(P, i, e);
#else
    extern __shared__ __align__(16) unsigned char smem[];
    const StageRegs SR = stage_issue<T, DIST>(Cg, I);
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
        typename Ops::StepResult r = Ops::step(P, goal, e, act, nullptr, key, i, st, noisy, seq_slot((T*)nullptr, 0), 0);
        rsum += r.reward;
        viols += (r.flags & FLAG_VIOLATION) ? 1 : 0;
        if (r.done) {
            ++dones;
            if (P.c.auto_reset) {
                dirty = true;
                Ops::reset(P, i, e, key, st);
            }
        }
    }
    if (reward_sum) seq_slot(reward_sum, i).store(rsum);
    if (done_count) seq_slot(done_count, i).store(dones);
    if (violation_count) seq_slot(violation_count, i).store(viols);
    if (last_obs) {
        const bool fresh = e.step == 0;
        const int32_t c0 = e.step - 1;
        Ops::write_obs(P, goal, st, e, key, fresh ? 1 : c0 + 2, fresh ? 0u : (uint32_t)(c0 + 1), fresh ? 0 : c0, i,
                       seq_slot(last_obs, i, P.c.nobs));
    }
    Ops::store(P, i, e, dirty);
}

// K control steps per launch with CALLER-SUPPLIED action sequences (scg_step_sequence): the loop
//     for t in range(K): obs[t], rew[t], done[t], info = vec_env.step(actions[t])
// of any open-loop consumer (sampling-based MPC / MPPI scoring candidate sequences, replay of logged actions, system
// identification, and the bench's synthetic-action workload) as ONE launch.  Per-step semantics are exactly scg_step's
// (same EnvOps::step, auto-reset, terminal observation, episode statistics); the state stays in registers between steps, every
// per-step output goes to a [K]-stacked array, and the action of step t+1 is requested before step t is integrated.  What
// it removes per control step: the 1.5 us dispatch floor, the state / counter round trip and the end-of-kernel store drain.
template <typename T>
struct SeqArgs {
    const T* actions;       // [K][N][nu]
    const T* adv;           // [K][N][adv_dim] or null
    int32_t k_steps;
    T* obs; T* reward; uint8_t* done; uint8_t* flags;       // [K][N](x nobs)
    T* terminal_obs;        // [K][N][nobs] or null
    T* mse;                 // [K][N] or null
    T* c_values;            // [K][rows][N] or null
    T* ep_stats;            // [N][4] or null (running totals, read-modify-written once per launch)
    T* fin_stats;           // [K][N][4] or null
    T* state;               // [K][nx][N] or null (env.state after the step and auto-reset)
    T* noisy_action;        // [K][nu][N] or null
};

template <int SYS, typename T, bool DIST>
__global__ __launch_bounds__(BLOCK) void step_sequence_kernel(const CfgParams<T>* __restrict__ Cg, const InstParams<T> I,
                                                              const SeqArgs<T> A) {
    using Ops = EnvOps<SYS, T, DIST, SCG_SEQ_ST_AUX>;
    using D = Dims<SYS>;
    const int i = I.env_first + blockIdx.x * blockDim.x + threadIdx.x;
    const int N = I.num_envs;
    const bool live = i < I.env_end;
    typename Ops::E e;
#ifdef SCG_SPEC
    constexpr CfgParams<T> kcfg = scg_make_spec_cfg<T>();     // compile-time constants (see scg_spec.h)
    const PV<T> P{kcfg, I};
    const GoalTab<T> goal{nullptr, I.x_goal, false};
    if (!live) return;
    const bool full_wave = __builtin_amdgcn_read_exec() == ~0ull;
    Ops::load_state(P, i, e);
#else
    extern __shared__ __align__(16) unsigned char smem[];
    const StageRegs SR = stage_issue<T, DIST>(Cg, I);
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
    const int nobs = P.c.nobs;
    const int rows = P.c.n_con_rows;
    T ep[4] = {(T)0, (T)0, (T)0, (T)0};
    if (A.ep_stats) seq_slot(A.ep_stats, i, 4).template load_row<4>(ep);
    int ad = 0;
    if constexpr (DIST) {
        if (A.adv && P.c.adversary_channel >= 0) ad = P.c.adversary_channel == SCG_CH_ACTION ? D::NU : D::DYN;
    }
    T act[D::NU], act_next[D::NU];
#pragma unroll
    for (int j = 0; j < D::NU; ++j) act_next[j] = A.actions[(size_t)i * D::NU + j];
    bool dirty = false;
    T st[D::NX];
    for (int t = 0; t < A.k_steps; ++t) {
#pragma unroll
        for (int j = 0; j < D::NU; ++j) act[j] = act_next[j];
        if (t + 1 < A.k_steps) {                        // next step's action: requested now, consumed one step later
#pragma unroll
            for (int j = 0; j < D::NU; ++j) act_next[j] = A.actions[((size_t)(t + 1) * N + i) * D::NU + j];
        }
        T advv[D::DYN > D::NU ? D::DYN : D::NU];
        const T* advp = nullptr;
        if constexpr (DIST) {
            if (ad > 0) {
                for (int j = 0; j < ad; ++j) advv[j] = A.adv[((size_t)t * N + i) * ad + j];
                advp = advv;
            }
        }
        const size_t tn = (size_t)t * N;
        const int32_t c0 = e.step;
        T noisy[D::NU];
        const Slot<T, SCG_SEQ_ST_AUX> cv = A.c_values ? seq_slot(A.c_values + (size_t)t * rows * N, i) : seq_slot((T*)nullptr, 0);
        typename Ops::StepResult r = Ops::step(P, goal, e, act, advp, key, i, st, noisy, cv, (size_t)N);
        seq_slot(A.reward + tn, i).store(r.reward);
        seq_slot(A.done + tn, i).store((uint8_t)(r.done ? 1 : 0));
        seq_slot(A.flags + tn, i).store((uint8_t)r.flags);
        if (A.mse) seq_slot(A.mse + tn, i).store(r.mse);
        if (A.noisy_action) {
            const Slot<T, SCG_SEQ_ST_AUX> na = seq_slot(A.noisy_action + tn * D::NU, i);
#pragma unroll
            for (int j = 0; j < D::NU; ++j) na.store(noisy[j], (size_t)j * N);
        }
        ep[0] += r.reward;
        ep[1] += (T)1;
        ep[2] += (r.flags & FLAG_VIOLATION) ? (T)1 : (T)0;
        ep[3] += r.mse;
        if (r.done) {
            if (A.fin_stats) seq_slot(A.fin_stats + tn * 4, i, 4).template store_row<4>(ep);
            ep[0] = ep[1] = ep[2] = ep[3] = (T)0;
        }
        const bool do_reset = r.done && P.c.auto_reset;
        const Slot<T, SCG_SEQ_ST_AUX> o_dst = seq_slot(A.obs + tn * nobs, i, nobs);
        if (Ops::obs_is_row(P)) {
            T row[2 * D::NX];
            int nrow = Ops::obs_row(P, goal, st, e, key, c0 + 2, (uint32_t)(c0 + 1), c0, i, nullptr, row);
            if (r.done && A.terminal_obs) Ops::store_obs_row(P, row, nrow, seq_slot(A.terminal_obs + tn * nobs, i, nobs));
            if (do_reset) {
                dirty = true;
                Ops::reset(P, i, e, key, st);
                nrow = Ops::obs_row(P, goal, st, e, key, 1, 0u, 0, i, nullptr, row);
            }
#ifdef SCG_SPEC
            constexpr int kNobs = kcfg.nobs;
            constexpr bool can_transpose = (kNobs * (int)sizeof(T)) % 16 == 0 && (kNobs == D::NX || kNobs == 2 * D::NX);
            if constexpr (can_transpose) {
                __shared__ __align__(16) unsigned char s_obs[BLOCK * kNobs * sizeof(T)];
                const int lane = (int)(threadIdx.x & 63);
                if (full_wave) store_rows_coalesced<T, kNobs>(o_dst, row, s_obs + (threadIdx.x >> 6) * (64 * kNobs * (int)sizeof(T)), lane);
                else Ops::store_obs_row(P, row, nrow, o_dst);
            } else {
                Ops::store_obs_row(P, row, nrow, o_dst);
            }
#else
            Ops::store_obs_row(P, row, nrow, o_dst);
#endif
        } else {
            if (r.done && A.terminal_obs)
                Ops::write_obs(P, goal, st, e, key, c0 + 2, (uint32_t)(c0 + 1), c0, i, seq_slot(A.terminal_obs + tn * nobs, i, nobs), nullptr);
            if (do_reset) {
                dirty = true;
                Ops::reset(P, i, e, key, st);
                Ops::write_obs(P, goal, st, e, key, 1, 0u, 0, i, o_dst, nullptr);
            } else {
                Ops::write_obs(P, goal, st, e, key, c0 + 2, (uint32_t)(c0 + 1), c0, i, o_dst, nullptr);
            }
        }
        if (A.state) {
            const Slot<T, SCG_SEQ_ST_AUX> so = seq_slot(A.state + tn * D::NX, i);
#pragma unroll
            for (int k = 0; k < D::NX; ++k) so.store(st[k], (size_t)k * N);
        }
    }
    if (A.ep_stats) seq_slot(A.ep_stats, i, 4).template store_row<4>(ep);
    Ops::store(P, i, e, dirty);
}

#if defined(SCG_SPEC) && defined(SCG_POLICY_H)
// ---------------------------------------------------------------------------------------------------------------
// K control steps per launch WITH THE POLICY IN THE LOOP (PPO.train_step's collector, controllers/ppo/ppo.py:266-284, and
// PPO.run's evaluation loop, :210-257): per step, for the 64 envs of a wave, the actor MLP obs -> H -> H -> act_dim
// (ppo_utils.py:149-199) runs on the matrix cores in exact float32 (scg_mlp.h), the action is sampled
// (mean + exp(logstd) N(0,1), Philox channel 5) or taken as the mean, the env steps (same EnvOps::step as scg_step, state
// in registers), and the rollout-buffer rows of step t are written: obs[t], act[t], logp[t], rew[t], done[t], flags[t],
// terminal_obs[t] where done.  The critic is NOT evaluated here: values of all (t, env) rows are one batched
// scg_mlp_forward over the finished obs buffer (full-chip MFMA instead of one wave per SIMD).
// Built only into libscg_spec_<hash>_pol<H>_<act>.so (-DSCG_POLICY_H= -DSCG_POLICY_ACT=): obs_dim and act_dim are the
// specialised config's, the hidden width and activation the policy's.
// One launch replaces K x (policy forward, sampling, log-prob, scg_step, episode statistics) launches; with 65 536 envs the
// per-step cost is the actor's ~290 MFMA issues per 32 envs (15 us) + the 2.5 us of in-register simulation.
struct PolicyArgs {
    const float* params; int32_t W1, b1, W2, b2, W3, b3, logstd_off; int32_t deterministic;
    int32_t k_steps;
    float* obs; float* act; float* logp; float* reward; uint8_t* done; uint8_t* flags; float* terminal_obs;
    float* ep_stats; float* episode_acc; int32_t max_episodes;
};
constexpr uint32_t RNG_CH_POLICY = 5;

// EPW = envs per wave.  64: lane = env, the wave runs the actor on its two 32-env column tiles one after the other.
// 32 (shards of <= 32 768 envs, where 64 envs per wave would leave SIMDs idle: 16 384 envs = 256 waves on 1024 SIMDs):
// lane (c, h) of both halves carries env c — the two halves simulate the same env redundantly (those lanes would idle
// otherwise) and between them hold the two k-rows of the MFMA's B operand, so the actor is ONE tile per step with no lane
// exchange; half 0 stores.  Twice the waves, each with half the matrix work per control step.
// WPW = waves per workgroup behind ONE weight image.  8 = two waves per SIMD (the kernel needs < 256 registers: h1 / h2 do not live
// across the env step): a control step is ~290 dependent MFMAs per column tile followed layer by layer by the activations of 16 NT
// values per lane on the vector unit (tanh: v_exp + v_rcp at quarter rate) and then the in-register simulation — with one wave per
// SIMD the matrix pipe idles through both; a second wave's products run under them (round 6: 65 536 envs = 2048 waves of 32 envs).
template <int SYS, bool DIST, int EPW, int WPW>
__global__ __launch_bounds__(64 * WPW) void rollout_policy_kernel(const InstParams<float> I, const PolicyArgs A) {
    using T = float;
    using Ops = EnvOps<SYS, T, DIST, SCG_SEQ_ST_AUX>;
    using D = Dims<SYS>;
    constexpr CfgParams<T> kcfg = scg_make_spec_cfg<T>();
    constexpr int NIN = kcfg.nobs, NU = D::NU, HID = SCG_POLICY_H, ACT = SCG_POLICY_ACT;
    static_assert(NIN == D::NX || NIN == 2 * D::NX, "the fused rollout serves single-row observations (goal horizon <= 1)");
    using L = MlpLds<NIN, HID, NU, 16>;
    constexpr int L1Q = L::L1Q;
    extern __shared__ __align__(16) float lds[];
    unsigned char* const s_obs = reinterpret_cast<unsigned char*>(lds + L::END);       // [WPW waves][64 rows][NIN] transpose scratch
    const PV<T> P{kcfg, I};
    const GoalTab<T> goal{nullptr, I.x_goal, false};
    {
        const MlpWeights w{A.params + A.W1, A.params + A.b1, A.params + A.W2, A.params + A.b2, A.params + A.W3, A.params + A.b3};
        mlp_fill_lds<NIN, HID, NU, 16, 64 * WPW>(lds, w, threadIdx.x);
    }
    __syncthreads();
    const int N = I.num_envs;
    static_assert(EPW == 64 || EPW == 32, "envs per wave");
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int i0 = EPW == 64 ? blockIdx.x * (64 * WPW) + threadIdx.x : (blockIdx.x * WPW + (threadIdx.x >> 6)) * 32 + (lane & 31);
    const bool live = i0 < N && (EPW == 64 || h == 0);
    const int i = i0 < N ? i0 : N - 1;                // surplus lanes shadow the last env (they take part in the MFMAs, never store)
    const bool full_wave = EPW == 64 && (blockIdx.x * (64 * WPW) + (threadIdx.x & ~63) + 64) <= N;
    unsigned char* const s_wave = s_obs + (threadIdx.x >> 6) * (64 * NIN * (int)sizeof(T));
    typename Ops::E e;
    Ops::load_state(P, i, e);
    Ops::load_params(P, i, e);
    const RngKey key{I.key0, I.key1};
    float ep[4] = {0.0f, 0.0f, 0.0f, 0.0f}, acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (A.ep_stats) seq_slot(A.ep_stats, i, 4).template load_row<4>(ep);
    if (A.episode_acc) seq_slot(A.episode_acc, i, 8).template load_row<8>(acc);
    float logstd[NU], sigma[NU], logp_const = 0.0f;
#pragma unroll
    for (int a = 0; a < NU; ++a) {
        logstd[a] = A.params[A.logstd_off + a];
        sigma[a] = __expf(logstd[a]);
        logp_const -= logstd[a] + 0.91893853320467274f;
    }
    // observation of the current state (what the previous step / reset returned)
    T st[D::NX], row[2 * D::NX];
    Ops::state_vector(e, st);
    {
        const bool fresh = e.step == 0;
        const int32_t c0 = e.step - 1;
        Ops::obs_row(P, goal, st, e, key, fresh ? 1 : c0 + 2, fresh ? 0u : (uint32_t)(c0 + 1), fresh ? 0 : c0, i, nullptr, row);
    }
    bool dirty = false;
    for (int t = 0; t <= A.k_steps; ++t) {
        // ---- rollout row obs[t]
        {
            const Slot<T, SCG_SEQ_ST_AUX> dst = seq_slot(A.obs + (size_t)t * N * NIN, i, NIN);
            if constexpr ((NIN * (int)sizeof(T)) % 16 == 0) {              // 16-byte rows leave through the LDS transpose (as in step_kernel)
                if (full_wave) store_rows_coalesced<T, NIN>(dst, row, s_wave, lane);
                else if (live) dst.template store_row<NIN>(row);
            } else {                                                       // e.g. 6-float rows (Quadrotor2D stabilisation), 2-float (1-D)
                if (live) dst.template store_row<NIN>(row);
            }
        }
        if (t == A.k_steps) break;
        // ---- actor forward for the wave's two 32-env column tiles (lane (c, h) owns env 32 h + c of the wave)
        float xo[L1Q], xr[L1Q];
#pragma unroll
        for (int q = 0; q < L1Q; ++q) {
            const float a0 = d_row(q, 0) < NIN ? row[d_row(q, 0) < NIN ? d_row(q, 0) : 0] : 0.0f;
            const float a1 = d_row(q, 1) < NIN ? row[d_row(q, 1) < NIN ? d_row(q, 1) : 0] : 0.0f;
            xo[q] = h ? a1 : a0;                                        // my own env's rows row(q, h)
            if constexpr (EPW == 64) xr[q] = __shfl_xor(h ? a0 : a1, 32, 64);      // the partner env's rows row(q, h)
        }
        float mean[NU];
        if constexpr (EPW == 32) {                                      // both halves hold env c: xo IS the B operand
            f32x16 h1[L::NT], h2[L::NT];
            mlp_forward_tile<NIN, HID, NU, ACT, 16>(lds, xo, h1, h2, mean, lane);
        } else {
            float x[L1Q], out[NU];
            f32x16 h1[L::NT], h2[L::NT];
#pragma unroll
            for (int q = 0; q < L1Q; ++q) x[q] = h == 0 ? xo[q] : xr[q];                // column tile 0: envs 0..31 of the wave
            mlp_forward_tile<NIN, HID, NU, ACT, 16>(lds, x, h1, h2, out, lane);
#pragma unroll
            for (int a = 0; a < NU; ++a) mean[a] = out[a];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < L1Q; ++q) x[q] = h == 1 ? xo[q] : xr[q];                // column tile 1: envs 32..63
            mlp_forward_tile<NIN, HID, NU, ACT, 16>(lds, x, h1, h2, out, lane);
#pragma unroll
            for (int a = 0; a < NU; ++a) mean[a] = h ? out[a] : mean[a];
        }
        // ---- action and its log-probability (ppo_utils.py:224-231; distributions.py:12-22)
        T act[NU];
        float logp = logp_const;
        if (A.deterministic) {
#pragma unroll
            for (int a = 0; a < NU; ++a) act[a] = mean[a];
        } else {
            const U4 w = rng_words(key, e.gid, e.episode, (uint32_t)e.step, rng_tag(RNG_CH_POLICY, 0, 0));
            float eps[4];
            {
                const float r0 = m_sqrt(-2.0f * m_log(u01<float>(w.x))), u0 = u01<float>(w.y);
                eps[0] = r0 * cos_2pi(u0);
                eps[1] = r0 * cos_2pi(u0 < 0.25f ? u0 + 0.75f : u0 - 0.25f);        // sin(2 pi u) = cos(2 pi (u - 1/4))
                const float r1 = m_sqrt(-2.0f * m_log(u01<float>(w.z))), u1 = u01<float>(w.w);
                eps[2] = r1 * cos_2pi(u1);
                eps[3] = r1 * cos_2pi(u1 < 0.25f ? u1 + 0.75f : u1 - 0.25f);
            }
#pragma unroll
            for (int a = 0; a < NU; ++a) {
                act[a] = __builtin_fmaf(sigma[a], eps[a], mean[a]);
                logp -= 0.5f * eps[a] * eps[a];
            }
        }
        // ---- the control step (identical code to scg_step's kernel)
        const int32_t c0 = e.step;
        T noisy[NU];
        typename Ops::StepResult r = Ops::step(P, goal, e, act, nullptr, key, i, st, noisy, seq_slot((T*)nullptr, 0), 0);
        const size_t tn = (size_t)t * N + i;
        if (live) {
#pragma unroll
            for (int a = 0; a < NU; ++a) A.act[tn * NU + a] = act[a];
            A.logp[tn] = logp;
            A.reward[tn] = r.reward;
            A.done[tn] = r.done ? 1 : 0;
            A.flags[tn] = r.flags;
        }
        ep[0] += r.reward; ep[1] += 1.0f; ep[2] += (r.flags & FLAG_VIOLATION) ? 1.0f : 0.0f; ep[3] += r.mse;
        Ops::obs_row(P, goal, st, e, key, c0 + 2, (uint32_t)(c0 + 1), c0, i, nullptr, row);
        if (r.done) {
            if (A.terminal_obs && live) seq_slot(A.terminal_obs + (size_t)t * N * NIN, i, NIN).template store_row<NIN>(row);
            if (A.max_episodes <= 0 || acc[0] < (float)A.max_episodes) {
                acc[0] += 1.0f; acc[1] += ep[0]; acc[2] += ep[1]; acc[3] += ep[2]; acc[4] += ep[3];
            }
            ep[0] = ep[1] = ep[2] = ep[3] = 0.0f;
            if (P.c.auto_reset) {
                dirty = true;
                Ops::reset(P, i, e, key, st);
                Ops::obs_row(P, goal, st, e, key, 1, 0u, 0, i, nullptr, row);
            }
        }
    }
    if (live) {
        if (A.ep_stats) seq_slot(A.ep_stats, i, 4).template store_row<4>(ep);
        if (A.episode_acc) seq_slot(A.episode_acc, i, 8).template store_row<8>(acc);
        Ops::store(P, i, e, dirty);
    }
}
#endif  // SCG_SPEC && SCG_POLICY_H

// Batched prior-model services (symbolic_systems.py:68-121: fc_func, df_func, fd_func) for model-based controllers that
// linearise / roll out the prior at many points (LQR/iLQR gains along a trajectory, GP-MPC data collection): one thread
// per sample (x, u) evaluates f(x, u), the Jacobians df/dx, df/du by central differences of the SAME device function
// the RK4 integrator mode uses (step eps), and one RK4 step of the control period.  Inertial parameters: the config's.
// Layouts: x [n][nx], u [n][nu], f [n][nx], A [n][nx][nx], B [n][nx][nu], xnext [n][nx] (row-major, any may be null).
template <int SYS, typename T, bool DIST>
__global__ __launch_bounds__(256) void prior_model_kernel(const CfgParams<T>* __restrict__ Cg, const InstParams<T> I, int n,
                                                           const T* __restrict__ xs, const T* __restrict__ us, T eps,
                                                           T* __restrict__ f_out, T* __restrict__ A_out, T* __restrict__ B_out,
                                                           T* __restrict__ xnext_out) {
    using Ops = EnvOps<SYS, T, DIST>;
    using D = Dims<SYS>;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#ifdef SCG_SPEC
    constexpr CfgParams<T> kcfg = scg_make_spec_cfg<T>();
    const PV<T> P{kcfg, I};
#else
    const PV<T> P{*Cg, I};              // service kernel: parameters straight from global memory
#endif
    typename Ops::E e;
#pragma unroll
    for (int k = 0; k < D::NP; ++k) e.par[k] = P.c.base_param[k];
    T x[D::NX], u[D::NU], f0[D::NX];
#pragma unroll
    for (int k = 0; k < D::NX; ++k) x[k] = xs[(size_t)i * D::NX + k];
#pragma unroll
    for (int k = 0; k < D::NU; ++k) u[k] = us[(size_t)i * D::NU + k];
    Ops::sym_f(P, e, x, u, f0);
    if (f_out) {
#pragma unroll
        for (int k = 0; k < D::NX; ++k) f_out[(size_t)i * D::NX + k] = f0[k];
    }
    const T inv2 = (T)0.5 / eps;
    if (A_out) {
        for (int c = 0; c < D::NX; ++c) {
            T xp[D::NX], xm[D::NX], fp[D::NX], fm[D::NX];
#pragma unroll
            for (int k = 0; k < D::NX; ++k) { xp[k] = x[k] + (k == c ? eps : (T)0); xm[k] = x[k] - (k == c ? eps : (T)0); }
            Ops::sym_f(P, e, xp, u, fp);
            Ops::sym_f(P, e, xm, u, fm);
#pragma unroll
            for (int r = 0; r < D::NX; ++r) A_out[((size_t)i * D::NX + r) * D::NX + c] = (fp[r] - fm[r]) * inv2;
        }
    }
    if (B_out) {
        for (int c = 0; c < D::NU; ++c) {
            T up[D::NU], um[D::NU], fp[D::NX], fm[D::NX];
#pragma unroll
            for (int k = 0; k < D::NU; ++k) { up[k] = u[k] + (k == c ? eps : (T)0); um[k] = u[k] - (k == c ? eps : (T)0); }
            Ops::sym_f(P, e, x, up, fp);
            Ops::sym_f(P, e, x, um, fm);
#pragma unroll
            for (int r = 0; r < D::NX; ++r) B_out[((size_t)i * D::NX + r) * D::NU + c] = (fp[r] - fm[r]) * inv2;
        }
    }
    if (xnext_out) {
        const T h = P.c.ctrl_dt;
        T k1[D::NX], k2[D::NX], k3[D::NX], k4[D::NX], y[D::NX];
#pragma unroll
        for (int k = 0; k < D::NX; ++k) { k1[k] = f0[k]; y[k] = x[k] + (T)0.5 * h * k1[k]; }
        Ops::sym_f(P, e, y, u, k2);
#pragma unroll
        for (int k = 0; k < D::NX; ++k) y[k] = x[k] + (T)0.5 * h * k2[k];
        Ops::sym_f(P, e, y, u, k3);
#pragma unroll
        for (int k = 0; k < D::NX; ++k) y[k] = x[k] + h * k3[k];
        Ops::sym_f(P, e, y, u, k4);
#pragma unroll
        for (int k = 0; k < D::NX; ++k)
            xnext_out[(size_t)i * D::NX + k] = x[k] + h * (T)(1.0 / 6.0) * (k1[k] + (T)2 * k2[k] + (T)2 * k3[k] + k4[k]);
    }
}

}  // namespace scg

// scg_sac.hip — ONE gradient step of the reference's SACAgent.update (controllers/sac/sac_utils.py:110-170) as a fixed
// sequence of MI355X kernels, compiled per network shape like scg_learn.hip:
//     hipcc -DSCG_S_NOBS=<obs_dim> -DSCG_S_H=<hidden> -DSCG_S_NU=<act_dim> -DSCG_S_ACT=<0 tanh|1 relu|2 leaky>
//     -> libscg_sac_<nobs>_<h>_<nu>_<act>.so       (C ABI: include/scg_sac.h)
//
// Why a sequence and not one kernel: the step touches five networks (actor, q1, q2, q1', q2') and its phases depend on each
// other through tiny per-sample vectors (action, log-prob, q, dq/da: a few floats per row), which cross global memory between
// launches.  Every MFMA kernel works on 32-sample column tiles, one tile per WORKGROUP, with the hidden layers split by feature
// over the workgroup's waves (see "wide tiles" below); all matrix products on v_mfma_f32_32x32x2_f32 (exact float32).
//   actor_fwd_kernel     batch rows ~ U[0, ring size) (SACBuffer.sample :399-413), then
//                        a, log pi (tanh-Gaussian, reparameterised)                       MLPActor.forward  (:185-222)
//                        + what the policy gradient needs of this pass: the waves' h1 / h2 tiles, tanh u, sigma, the clamp's pass flags
//   q_kernel<1>          q_y(obs, a) and dq_y/da for y = 1, 2 (forward + data gradient)   compute_policy_loss (:110-127)
//   actor_grad_kernel    d mean(alpha log pi - min q)/d(actor) from the stored pass: starts at the loss derivatives
//   reduce_kernel        sum of the partials + Adam: actor (+ log_alpha), soft update of the actor's target copy
//   actor_fwd_kernel     a', log pi' at next_obs with the UPDATED actor                   compute_q_loss    (:129-141)
//                        [scg_sac_update_n: + the NEXT step's first launch as a second job — same actor, 2 x 128 workgroups fill the chip]
//   q_kernel<0>          target networks at (next_obs, a'); beside them (blockIdx.y = 2, 3) the online critics' forward pass at
//                        (obs, act) — it does not depend on the targets — leaving q, the h1 / h2 tiles, reward and mask per batch row
//   q_kernel<2>          d[mean (q_y - target)^2]/d(q_y), y = 1, 2, from the stored pass: starts at the loss derivatives
//   reduce_kernel        sum of the partials + Adam: critics, soft update of their target copies + the step's bookkeeping  (:163-168)
//   [finish_kernel       step counters, loss statistics: data-parallel path only]
// (Data-parallel callers run the phases separately — include/scg_sac.h — with reduce_kernel writing the gradient only and
//  adam_kernel stepping after the all-reduce.)
// What a launch leaves for the next one (partial vectors, activation tiles) is stored write-through, 1 KB of consecutive addresses per
// instruction: it is read on other XCDs, so it must reach the memory side before the kernel retires — written back that is one flush
// of megabytes behind the last workgroup.  Requests at kernel entry are ordered by need, and nothing in front of the first barrier
// waits for an index -> row gather (profiles/r06_sac_actor_grad_timeline.txt).
// Gradient reduction: every WORKGROUP owns one partial gradient vector in global memory (a workgroup usually owns one tile: plain
// stores), reduce_kernel sums the partials in a fixed order — no atomics, bitwise reproducible.
// History: until late in round 3 a tile belonged to ONE wave behind a 100 KB LDS weight image (scg_learn.hip's scheme, right for
// PPO's 2000-tile minibatches): 0.211 ms per step at batch 4096, where 128 tiles left 7/8 of the SIMDs idle behind 288-864
// dependent MFMAs each; the wide tiles run the same step in 0.114 ms (profiles/r03_sac_update_cost.json), round 6's in 0.078 ms
// (profiles/r06_sac_step_ab.txt: stored passes, merged launches, write-through hand-over, ordered requests).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/scg_sac.h"
#include "scg_adam.h"
#include "scg_mlp.h"
#include "scg_once.h"
#include "scg_rng.h"

#ifndef SCG_S_NOBS
#error "compile with -DSCG_S_NOBS= -DSCG_S_H= -DSCG_S_NU= -DSCG_S_ACT="
#endif

using namespace scg;

constexpr int NOBS = SCG_S_NOBS, HID = SCG_S_H, NU = SCG_S_NU, ACT = SCG_S_ACT;
constexpr int NQ = NOBS + NU;                 // Q-network input: (obs, act)
constexpr int NA = 2 * NU;                    // actor head: mu | log_std
constexpr int NT = HID / 32;
constexpr int WAVES = 4;
static_assert(NU >= 1 && NU <= 4 && NQ < 32 && HID % 32 == 0 && HID <= 128, "unsupported SAC shape");

static thread_local std::string g_err;
static int fail(int code, const std::string& m) { g_err = m; return code; }
extern "C" const char* scg_sac_last_error(void) { return g_err.c_str(); }
extern "C" void scg_sac_shape(int32_t* nobs, int32_t* hidden, int32_t* nu, int32_t* act) { *nobs = NOBS; *hidden = HID; *nu = NU; *act = ACT; }
#ifndef SCG_SRC_HASH
#define SCG_SRC_HASH 0ULL
#endif
#define SCG_STR2(x) #x
#define SCG_STR(x) SCG_STR2(x)
extern "C" const char* scg_sac_source_hash_tag(void) { return "SCG_SRC_HASH:" SCG_STR(SCG_SRC_HASH); }
#define HIP_TRY(e) do { hipError_t _e = (e); if (_e != hipSuccess) return fail(-2, std::string(#e) + ": " + hipGetErrorString(_e)); } while (0)

__host__ __device__ static inline MlpWeights weights_of(const float* p, const scg_mlp_layout& L) {
    return MlpWeights{p + L.W1, p + L.b1, p + L.W2, p + L.b2, p + L.W3, p + L.b3};
}

constexpr float LOG_SQRT_2PI = 0.91893853320467274f, LOG2F = 0.69314718055994531f;

// ------------------------------------------------------------------ partial gradient vector of one wave
template <int NIN, int NOUT>
struct Part {
    static constexpr int DW1 = 0;                           // [NIN + 1][H]: column NIN is db1
    static constexpr int DB2 = DW1 + (NIN + 1) * HID;
    static constexpr int DW3 = DB2 + HID;                   // [NOUT][H]
    static constexpr int DB3 = DW3 + NOUT * HID;            // [8]
    static constexpr int STAT = DB3 + 8;                    // [4]
    static constexpr int DW2 = STAT + 4;                    // [NT * NT tiles][4 g][64 lanes][4]: accumulator word q = 4 g + r of lane
    static constexpr int END = DW2 + HID * HID;
};
constexpr int PSTRIDE = (Part<NOBS, NA>::END > Part<NQ, 1>::END ? Part<NOBS, NA>::END : Part<NQ, 1>::END);

// (a branch on the wave-uniform `first`, not a select: the select form LOADS the partial word on every call — a global round trip per
//  output row in the dW3 phase of a workgroup's only tile, tools/sac_timeline.py)
__device__ __forceinline__ void padd(float* p, float v, bool first) {
    if (first) *p = v;
    else *p += v;
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 16-byte store WRITTEN THROUGH (sc0 sc1, as scg_learn.hip's partial vectors): what a launch leaves for the NEXT launch — partial gradient
// vectors, activation tiles — is read there from other XCDs, so it has to reach the memory side before this kernel may retire; written
// back, that is one flush of megabytes behind the last workgroup's last store, written through it drains while the kernel still computes.
// `base` must be wave-uniform (it becomes the buffer resource); `word` = this lane's float offset from it.
#ifndef SCG_S_STORE_AUX
#define SCG_S_STORE_AUX 17
#endif
typedef unsigned int u32x4 __attribute__((vector_size(16)));
__device__ __forceinline__ void store_wt(float* base, uint32_t word, const f32x4 v) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0xffffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, 4u * word, 0, SCG_S_STORE_AUX);
}

// Sum over lanes 0..31 of v, valid in lane 0: through 32 words of the wave's LDS (one 4-byte write per lane, eight 16-byte reads in lane 0,
// a fixed order) — a butterfly over the lanes is 5-6 dependent ds_bpermute round trips.
__device__ __forceinline__ float row_sum32(float* row, float v, int lane) {
    if (lane < 32) row[lane] = v;
    wave_sync();
    float s = 0.0f;
    if (lane == 0) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(row + 4 * g);
            s += d4.x; s += d4.y; s += d4.z; s += d4.w;
        }
    }
    return s;
}

// x[q] = input feature row(q, h) of one sample, the input being [a [NA_] | b [NB_]] (rows >= NA_ + NB_ are zero)
template <int L1Q, int NA_, int NB_>
__device__ __forceinline__ void load_x2(const float* __restrict__ a, const float* __restrict__ b, int h, float* x) {
#pragma unroll
    for (int q = 0; q < L1Q; ++q) {
        const int f0 = d_row(q, 0), f1 = d_row(q, 1);           // compile-time after unrolling
        const float v0 = f0 < NA_ ? a[f0 < NA_ ? f0 : 0] : (f0 < NA_ + NB_ ? b[f0 < NA_ + NB_ && f0 >= NA_ ? f0 - NA_ : 0] : 0.0f);
        const float v1 = f1 < NA_ ? a[f1 < NA_ ? f1 : 0] : (f1 < NA_ + NB_ ? b[f1 < NA_ + NB_ && f1 >= NA_ ? f1 - NA_ : 0] : 0.0f);
        x[q] = h ? v1 : v0;
    }
}

// ------------------------------------------------------------------ kernels
struct Common {
    const int32_t* idx; int batch; int n_part;
    const float* obs; const float* act; const float* rew; const float* next_obs; const float* mask;
    float low[4], high[4];
    const float* log_alpha;                     // device scalar (d_params + n_params)
    float gamma;
    uint32_t k0, k1; const uint32_t* counter;
};

// batch row r of this update: a replay-ring slot ~ U[0, ring size) (or the caller's, for tests)
__device__ __forceinline__ int sample_row(int r, const int32_t* __restrict__ ring_size, const int32_t* __restrict__ idx_in, uint32_t cnt,
                                          uint32_t k0, uint32_t k1) {
    if (idx_in) return idx_in[r];
    const uint32_t n = (uint32_t)max(*ring_size, 1);
    const U4 w = philox4x32_10(U4{cnt, (uint32_t)r, 0u, 0x5ac0u}, k0, k1);
    return (int32_t)int_below(w.x, n);
}

// N(0,1) draws for one batch row: Box-Muller on a Philox block (stream: 1 policy-loss action, 2 target action)
__device__ __forceinline__ void normal4(uint32_t cnt, uint32_t row, uint32_t stream, uint32_t k0, uint32_t k1, float* n) {
    const U4 w = philox4x32_10(U4{cnt, row, stream, 0x5ac1u}, k0, k1);
    const float r0 = sqrtf(-2.0f * __logf(u01<float>(w.x))), r1 = sqrtf(-2.0f * __logf(u01<float>(w.z)));
    float s0, c0, s1, c1;
    __sincosf(6.283185307179586f * u01<float>(w.y), &s0, &c0);
    __sincosf(6.283185307179586f * u01<float>(w.w), &s1, &c1);
    n[0] = r0 * c0; n[1] = r0 * s0; n[2] = r1 * c1; n[3] = r1 * s1;
}

__device__ __forceinline__ float softplusf(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

// tanh-Gaussian head (sac_utils.py:203-222 / sac.py::MLPActor.forward): out = mu | log_std
__device__ __forceinline__ void squash(const float* out, const float* eps, const float* low, const float* high, float* u, float* th, float* sig,
                                       float* a, float& logp) {
    logp = 0.0f;
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        const float ls = fminf(fmaxf(out[NU + j], -20.0f), 2.0f);
        sig[j] = expf(ls);
        u[j] = __builtin_fmaf(sig[j], eps[j], out[j]);
        th[j] = tanhf(u[j]);
        a[j] = low[j] + 0.5f * (th[j] + 1.0f) * (high[j] - low[j]);
        logp += -0.5f * eps[j] * eps[j] - ls - LOG_SQRT_2PI - 2.0f * (LOG2F - u[j] - softplusf(-2.0f * u[j]));
    }
}

// deterministic action of a batch (evaluation): a = low + 0.5 (tanh(mu) + 1)(high - low)
__global__ __launch_bounds__(64 * WAVES, 1) void actor_act_kernel(const float* __restrict__ params, const scg_mlp_layout lay,
                                                                   const float* __restrict__ obs, int m, float4 low, float4 high,
                                                                   float* __restrict__ a_out) {
    using L = MlpLds<NOBS, HID, NA>;
    extern __shared__ __align__(16) float lds[];
    mlp_fill_lds<NOBS, HID, NA>(lds, weights_of(params, lay), threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const float lo[4] = {low.x, low.y, low.z, low.w}, hi[4] = {high.x, high.y, high.z, high.w};
    const int n_tiles = (m + 31) / 32;
    for (int tile = blockIdx.x * WAVES + wave; tile < n_tiles; tile += gridDim.x * WAVES) {
        int s = tile * 32 + c;
        const bool live = s < m;
        s = live ? s : m - 1;
        float x[L::L1Q];
        load_x2<L::L1Q, NOBS, 0>(obs + (size_t)s * NOBS, nullptr, h, x);
        f32x16 h1[NT], h2[NT];
        float out[NA];
        mlp_forward_tile<NOBS, HID, NA, ACT, 20, MLP_ACT_NONE>(lds, x, h1, h2, out, lane);
        if (live && h == 0) {
#pragma unroll
            for (int j = 0; j < NU; ++j) a_out[(size_t)s * NU + j] = lo[j] + 0.5f * (tanhf(out[j]) + 1.0f) * (hi[j] - lo[j]);
        }
    }
}

// -DSCG_S_TIMING (tools/sac_timeline.py; development builds only): wave 0 of workgroup (0, 0) of actor_grad_kernel and q_kernel<2> stamps the
// shader clock at its phase boundaries into a device array the host reads back with scg_sac_timeline().
#ifdef SCG_S_TIMING
__device__ unsigned long long g_sac_tl[2][16];
#define SCG_S_STAMP(which, k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_sac_tl[which][k] = __builtin_readcyclecounter(); } while (0)
extern "C" int scg_sac_timeline(unsigned long long* h_out) {
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_sac_tl), sizeof(g_sac_tl)) == hipSuccess ? 0 : -2;
}
#else
#define SCG_S_STAMP(which, k) do {} while (0)
#endif

// ================================================================== collector (SAC.train_step's env-facing half, sac.py:273-311)
// A SAMPLED action of the policy for a batch of observations (MLPActorCritic.act(obs), sac_utils.py:258-262, deterministic = False):
// a = low + 0.5 (tanh(mu + exp(clamp(log_std, -20, 2)) eps) + 1)(high - low), eps ~ N(0, 1) from Philox (counter word = *counter,
// row, stream 3) or the caller's (tests).  One launch in place of ~12 PyTorch kernels (three GEMMs, activations, clamp, exp, randn, ...).
__global__ __launch_bounds__(64 * WAVES, 1) void actor_sample_kernel(const float* __restrict__ params, const scg_mlp_layout lay,
                                                                      const float* __restrict__ obs, int m, float4 low, float4 high,
                                                                      uint32_t k0, uint32_t k1, const uint32_t* __restrict__ counter,
                                                                      const float* __restrict__ eps_in, float* __restrict__ a_out) {
    using L = MlpLds<NOBS, HID, NA>;
    extern __shared__ __align__(16) float lds[];
    mlp_fill_lds<NOBS, HID, NA>(lds, weights_of(params, lay), threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const float lo[4] = {low.x, low.y, low.z, low.w}, hi[4] = {high.x, high.y, high.z, high.w};
    const uint32_t cnt = counter ? *counter : 0u;
    const int n_tiles = (m + 31) / 32;
    for (int tile = blockIdx.x * WAVES + wave; tile < n_tiles; tile += gridDim.x * WAVES) {
        int s = tile * 32 + c;
        const bool live = s < m;
        s = live ? s : m - 1;
        float x[L::L1Q];
        load_x2<L::L1Q, NOBS, 0>(obs + (size_t)s * NOBS, nullptr, h, x);
        f32x16 h1[NT], h2[NT];
        float out[NA], eps[4];
        mlp_forward_tile<NOBS, HID, NA, ACT, 20, MLP_ACT_NONE>(lds, x, h1, h2, out, lane);
        if (eps_in) {
#pragma unroll
            for (int j = 0; j < NU; ++j) eps[j] = eps_in[(size_t)s * NU + j];
        } else {
            normal4(cnt, (uint32_t)s, 3u, k0, k1, eps);
        }
        if (live && h == 0) {
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                const float ls = fminf(fmaxf(out[NU + j], -20.0f), 2.0f);
                const float u = __builtin_fmaf(expf(ls), eps[j], out[j]);
                a_out[(size_t)s * NU + j] = lo[j] + 0.5f * (tanhf(u) + 1.0f) * (hi[j] - lo[j]);
            }
        }
    }
}
// warm-up actions: action_space.sample() per env (sac.py:276-277), a ~ U[low, high) per dimension
__global__ __launch_bounds__(256) void uniform_action_kernel(int m, float4 low, float4 high, uint32_t k0, uint32_t k1,
                                                              const uint32_t* __restrict__ counter, float* __restrict__ a_out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= m) return;
    const float lo[4] = {low.x, low.y, low.z, low.w}, hi[4] = {high.x, high.y, high.z, high.w};
    const U4 w = philox4x32_10(U4{counter ? *counter : 0u, (uint32_t)s, 4u, 0x5ac1u}, k0, k1);
    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int j = 0; j < NU; ++j) a_out[(size_t)s * NU + j] = lo[j] + (hi[j] - lo[j]) * u01<float>(ww[j]);
}

// One vectorised env step into the replay ring (SACBuffer.push with the time-limit fix-up of sac.py:287-305): row pos + i (mod capacity)
// <- (obs the action was taken at, action, reward, next observation — the TERMINAL observation where the episode was truncated by the
// time limit —, mask = 1 if truncated else 1 - done); the persistent current-observation batch becomes the step's observation.
// One thread per (env, observation element).  The write position is read here by everybody and advanced by ring_advance_kernel.
struct RingArgs {
    float* obs; float* act; float* rew; float* next_obs; float* mask; int capacity;
    long long* pos; float* size_f; int32_t* size_i; uint32_t* counter;
};
__global__ __launch_bounds__(256) void ring_push_kernel(const RingArgs R, float* __restrict__ cur_obs, const float* __restrict__ act,
                                                         const float* __restrict__ rew, const float* __restrict__ next,
                                                         const float* __restrict__ term, const uint8_t* __restrict__ done,
                                                         const uint8_t* __restrict__ flags, int n) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * NOBS) return;
    const int i = gid / NOBS, e = gid - i * NOBS;
    const size_t slot = (size_t)((*R.pos + i) % R.capacity);
    const bool dn = done[i] != 0, trunc = dn && (flags[i] & 1);
    const float nv = next[gid];
    R.obs[slot * NOBS + e] = cur_obs[gid];
    R.next_obs[slot * NOBS + e] = trunc ? term[gid] : nv;
    cur_obs[gid] = nv;
    if (e < NU) R.act[slot * NU + e] = act[(size_t)i * NU + e];
    if (e == 0) { R.rew[slot] = rew[i]; R.mask[slot] = trunc ? 1.0f : (dn ? 0.0f : 1.0f); }
}
__global__ void ring_advance_kernel(const RingArgs R, int n) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    *R.pos = (*R.pos + n) % R.capacity;
    if (R.size_f) *R.size_f = fminf(*R.size_f + (float)n, (float)R.capacity);
    if (R.size_i) *R.size_i = min(*R.size_i + n, R.capacity);
    if (R.counter) *R.counter += 1u;
}

// ================================================================== wide tiles
// A batch of 4096 is only 128 tiles.  The NT waves of a workgroup SHARE one tile: wave w owns the hidden features
// [32 w, 32 w + 32) of both layers, the activations cross an LDS exchange between the layers, and every wave's dependent MFMA
// chain is 1 / NT of the network's.  Each weight is then used by exactly one wave, once per tile: the MFMA operands are read from
// the parameter vector straight into registers (no LDS image, no fill) — only W3 and the biases sit in the LDS.
//   forward :  L1 (own 32 features) -> h1 tile to H1X -> barrier -> L2 (A = own rows of W2, B = all h1 tiles) -> own partial of the
//              output layer to RED -> barrier -> every wave sums the NT partials (fixed order)
//   backward:  dW3 / db2 / dz2 on the own tile; dz2 tile to DZX, own h1 tile transposed to H1T -> barrier ->
//              data gradient of the OWN input-feature tile (all dz2 tiles x own columns of W2; computed transposed when it feeds
//              dW1 — scg_learn.hip's trick — plain when it feeds dq/da), dW1 | db1 slice, the dW2 tiles [all tau][rho = w]
//              (A = H1T tiles, B = own dz2^T)
// Partial gradient vectors: one per workgroup, word order of Part<> (dW2 in the accumulator's [tile][lane][q] order).
namespace wide {
constexpr int XW = 20, XT = 64 * XW;                    // an exchanged tile: 16 words per lane, padded to 20 (conflict-free 16-byte access)
// LDS of a wide-tile kernel = [Small<NOUT> of each network it evaluates][Xch]: the per-network constants and ONE set of exchange
// buffers shared by the networks a workgroup walks through one after the other (workgroup barrier in between).
template <int NOUT>
struct Small {
    static constexpr int W3 = 0;                                        // [NOUT][H]
    static constexpr int B1 = W3 + NOUT * HID, B2 = B1 + HID, B3 = B2 + HID;   // b3: [8]
    static constexpr int W1A = B3 + 8;                                  // [4][H]    W1's action columns (Q networks, dq/da)
    static constexpr int END = W1A + 4 * HID;
    static_assert(NOUT <= 8 && (END % 4) == 0, "layout");
};
struct Xch {
    static constexpr int H1X = 0;                                       // [NT][XT]  h1 tiles, accumulator layout (lane = sample)
    static constexpr int RED = H1X + NT * XT;                           // [NT][8][32] per-wave partial outputs
    static constexpr int FWD_END = RED + NT * 8 * 32;
    static constexpr int DZX = FWD_END;                                 // [NT][XT]  dz2 tiles, accumulator layout
    static constexpr int H1T = DZX + NT * XT;                           // [NT][XT]  h1 tiles transposed (lane = feature)
    static constexpr int DIN = H1T + NT * XT;                           // [NT][4][32] per-wave partial input gradients
    static constexpr int WAVE = DIN + NT * 4 * 32;                      // per wave: scr | xs | dout_l
    static constexpr int WAVE_WORDS = TR_WORDS + 34 * 32 + 8 * 32;
    static constexpr int END = WAVE + NT * WAVE_WORDS;
};

// W3 and the biases of one network -> LDS (all threads; caller barriers), in two halves: the REQUESTS (small_load) go out first in a
// kernel, the LDS writes (small_store) after the other requests of the prologue have been issued — the memory counter retires in
// order, so the barrier behind the stores then waits for this handful of loads only, not for the ~150 operand loads behind them
// (tools/sac_timeline.py: 3.6 us from kernel entry to the first barrier when the small block was requested last).
template <int NOUT>
struct SmallRegs { float w3[(NOUT * HID + 64 * NT - 1) / (64 * NT)]; float b1, b2, b3; };
template <int NOUT>
__device__ __forceinline__ void small_load(SmallRegs<NOUT>& R, const MlpWeights& w, int tid) {
    constexpr int IT = (NOUT * HID + 64 * NT - 1) / (64 * NT);
#pragma unroll
    for (int j = 0; j < IT; ++j) { const int k = tid + j * 64 * NT; R.w3[j] = k < NOUT * HID ? w.W3[k] : 0.0f; }
    static_assert(HID <= 64 * NT * 2, "one bias word per thread and layer");
    R.b1 = tid < HID ? w.b1[tid] : 0.0f;
    R.b2 = tid < HID ? w.b2[tid] : 0.0f;
    R.b3 = tid < NOUT ? w.b3[tid] : 0.0f;
}
template <int NOUT>
__device__ __forceinline__ void small_store(float* lds, const SmallRegs<NOUT>& R, int tid) {     // lds = the network's Small<NOUT> block
    using S = Small<NOUT>;
    constexpr int IT = (NOUT * HID + 64 * NT - 1) / (64 * NT);
#pragma unroll
    for (int j = 0; j < IT; ++j) { const int k = tid + j * 64 * NT; if (k < NOUT * HID) lds[S::W3 + k] = R.w3[j]; }
    if (tid < HID) { lds[S::B1 + tid] = R.b1; lds[S::B2 + tid] = R.b2; }
    if (tid < 8) lds[S::B3 + tid] = R.b3;
}

// MFMA operands of wave `wave` from the torch-layout parameters:
//   a1[q]      = W1[32 wave + i][row(q, h)]                  A operand of layer 1            (i = lane & 31, h = lane >> 5)
//   a2[tau][q] = W2[32 wave + i][32 tau + row(q, h)]         A operand of layer 2
//   bt[rho][q] = W2[32 rho + row(q, h)][32 wave + i]         operand of the data gradient of the own input tile
template <int NIN, int L1Q>
__device__ __forceinline__ void load_a1(const float* __restrict__ W1, int wave, int lane, float* a1) {
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int q = 0; q < L1Q; ++q) {
        const int f = d_row(q, 0) + 4 * h;
        a1[q] = f < NIN ? W1[(size_t)(32 * wave + i) * NIN + (f < NIN ? f : 0)] : 0.0f;
    }
}
__device__ __forceinline__ void load_a2(const float* __restrict__ W2, int wave, int lane, float (&a2)[NT][16]) {
    const int i = lane & 31, h = lane >> 5;
    const float* row = W2 + (size_t)(32 * wave + i) * HID + 4 * h;
    if ((reinterpret_cast<uintptr_t>(W2) & 15) == 0) {
#pragma unroll
        for (int tau = 0; tau < NT; ++tau)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(row + 32 * tau + 8 * g);
                a2[tau][4 * g] = v.x; a2[tau][4 * g + 1] = v.y; a2[tau][4 * g + 2] = v.z; a2[tau][4 * g + 3] = v.w;
            }
    } else {
#pragma unroll
        for (int tau = 0; tau < NT; ++tau)
#pragma unroll
            for (int q = 0; q < 16; ++q) a2[tau][q] = row[32 * tau + d_row(q, 0)];
    }
}
__device__ __forceinline__ void load_bt(const float* __restrict__ W2, int wave, int lane, float (&bt)[NT][16]) {
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int rho = 0; rho < NT; ++rho)
#pragma unroll
        for (int q = 0; q < 16; ++q) bt[rho][q] = W2[(size_t)(32 * rho + d_row(q, h)) * HID + 32 * wave + i];
}

__device__ __forceinline__ void put_tile(float* slot, const f32x16& t) {          // slot = base + lane * XW
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(slot + 4 * g) = (f32x4){t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]};
}
__device__ __forceinline__ void get_tile(const float* slot, float* t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(slot + 4 * g);
        t[4 * g] = v.x; t[4 * g + 1] = v.y; t[4 * g + 2] = v.z; t[4 * g + 3] = v.w;
    }
}

// Forward pass of the workgroup's tile: h1, h2 = this wave's feature tile of each hidden layer (accumulator layout), out = the
// network outputs of this lane's sample (every wave, both lane halves).  Two workgroup barriers.
template <int NIN, int NOUT, int ACT2>
__device__ __forceinline__ void forward(const float* sm, float* xch, const float* a1, const float (&a2)[NT][16], const float* x, int wave, int lane,
                                        f32x16& h1, f32x16& h2, float* out) {
    using S = Small<NOUT>;
    using X = Xch;
    constexpr int L1Q = 4 * ((NIN + 7) / 8);
    const int c = lane & 31, h = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(sm + S::B1 + 32 * wave + 8 * g + 4 * h);
        acc[4 * g] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
    }
#pragma unroll
    for (int q = 0; q < L1Q; ++q) acc = mfma32(a1[q], x[q], acc);
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = mlp_act<ACT>(acc[q]);
    h1 = acc;
    put_tile(xch + X::H1X + wave * XT + lane * XW, h1);
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(sm + S::B2 + 32 * wave + 8 * g + 4 * h);
        acc[4 * g] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
    }
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
        float hb[16];
        get_tile(xch + X::H1X + tau * XT + lane * XW, hb);
#pragma unroll
        for (int q = 0; q < 16; ++q) acc = mfma32(a2[tau][q], hb[q], acc);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = mlp_act<ACT2>(acc[q]);
    h2 = acc;
    // (all outputs' partial sums first, then the NOUT lane-half exchanges back to back: written per output — sum, exchange, store — the
    //  exchanges were NOUT dependent LDS round trips in a row, 8 for the actor's head)
    float so[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        float s = 0.0f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(sm + S::W3 + o * HID + 32 * wave + 8 * g + 4 * h);
            s = __builtin_fmaf(w.x, h2[4 * g], s); s = __builtin_fmaf(w.y, h2[4 * g + 1], s);
            s = __builtin_fmaf(w.z, h2[4 * g + 2], s); s = __builtin_fmaf(w.w, h2[4 * g + 3], s);
        }
        so[o] = s;
    }
    float sx[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) sx[o] = __shfl_xor(so[o], 32, 64);
    if (h == 0) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) xch[X::RED + (wave * 8 + o) * 32 + c] = so[o] + sx[o];
    }
    __syncthreads();
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        float s = sm[S::B3 + o];
#pragma unroll
        for (int w = 0; w < NT; ++w) s += xch[X::RED + (w * 8 + o) * 32 + c];
        out[o] = s;
    }
}

// the wave's own sample cache for the dW1 product: xs[column][sample], a ones row and a zeros row behind the inputs
template <int NIN, int L1Q>
__device__ __forceinline__ void cache_x(float* xs, const float* x, int c, int h) {
#pragma unroll
    for (int q = 0; q < L1Q; ++q) {
        const int f = d_row(q, 0);
        if (f + 4 < NIN) xs[(f + 4 * h) * 32 + c] = x[q];
        else if (f < NIN) { if (h == 0) xs[f * 32 + c] = x[q]; }
    }
    if (h == 0) { xs[NIN * 32 + c] = 1.0f; xs[(NIN + 1) * 32 + c] = 0.0f; }
}

// Backward pass of the workgroup's tile (see the scheme above).  h1, h2: this wave's tiles from forward(); dout: d loss / d out of
// this lane's sample (identical in every wave).  WGRAD: this wave's slices of the weight / bias gradients into the workgroup's
// partial vector P;  DIN: din[j] = d loss / d input[NIN - NU + j] of this lane's sample (every wave).  One workgroup barrier
// (two with DIN); the caller barriers before the next tile's forward().
template <int NIN, int NOUT, int ACT2, bool WGRAD, bool DIN, int TL = -1>
__device__ __forceinline__ void backward(const float* sm, float* xch, const float (&bt)[NT][16], f32x16& h1, f32x16& h2,
                                         const float* dout, int wave, int lane, float* P, bool first, float* din, const float* xin = nullptr) {
    using S = Small<NOUT>;
    using X = Xch;
    using G = Part<NIN, NOUT>;
    const int c = lane & 31, h = lane >> 5;
    float* const wl = xch + X::WAVE + wave * X::WAVE_WORDS;
    float* const scr = wl; float* const xs = wl + TR_WORDS; float* const dout_l = xs + 34 * 32;
    float zt[16];                                                       // dz2^T of the own tile (WGRAD)
    if constexpr (WGRAD) {
        if (h == 0) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) dout_l[o * 32 + c] = dout[o];
        }
        wave_sync();
        // db3: lane o of the LAST wave sums row o of the LDS copy (8 x 16-byte reads, a fixed order).  (As a butterfly over the lanes it was
        // 5 dependent ds_bpermute round trips per output, and the branches of the stores in between kept the compiler from overlapping the
        // outputs' chains: 40 in a row for the actor's head — ~2 us in front of the workgroup barrier, tools/sac_timeline.py.)
        if (wave == NT - 1 && lane < NOUT) {
            float v = 0.0f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dout_l + lane * 32 + 4 * g);
                v += d4.x; v += d4.y; v += d4.z; v += d4.w;
            }
            padd(P + G::DB3 + lane, v, first);
        }
        float t[16];
        tile_transpose(scr, h2, t, lane);                               // t[q] = h2[feature 32 wave + c][sample row(q, h)]
        float a3[NOUT];                                                 // dW3[o][f] = sum_s h2[f][s] dout[o][s]: all outputs, then the exchanges
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            float acc = 0.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 dv = *reinterpret_cast<const f32x4*>(dout_l + o * 32 + 8 * g + 4 * h);
                acc = __builtin_fmaf(t[4 * g], dv.x, acc); acc = __builtin_fmaf(t[4 * g + 1], dv.y, acc);
                acc = __builtin_fmaf(t[4 * g + 2], dv.z, acc); acc = __builtin_fmaf(t[4 * g + 3], dv.w, acc);
            }
            a3[o] = acc;
        }
        float x3[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) x3[o] = __shfl_xor(a3[o], 32, 64);
        if (h == 0) {
            float* const p3 = P + G::DW3 + 32 * wave + c;
            if (first) {
#pragma unroll
                for (int o = 0; o < NOUT; ++o) p3[o * HID] = a3[o] + x3[o];
            } else {
#pragma unroll
                for (int o = 0; o < NOUT; ++o) p3[o * HID] += a3[o] + x3[o];
            }
        }
    }
    if constexpr (TL >= 0) SCG_S_STAMP(TL, 4);
    // dz2 = (W3^T dout) * act2'(h2), in place
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float dh[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(sm + S::W3 + o * HID + 32 * wave + 8 * g + 4 * h);
            dh[0] = __builtin_fmaf(wv.x, dout[o], dh[0]); dh[1] = __builtin_fmaf(wv.y, dout[o], dh[1]);
            dh[2] = __builtin_fmaf(wv.z, dout[o], dh[2]); dh[3] = __builtin_fmaf(wv.w, dout[o], dh[3]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) h2[4 * g + r] = dh[r] * mlp_dact<ACT2>(h2[4 * g + r]);
    }
    put_tile(xch + X::DZX + wave * XT + lane * XW, h2);
    if constexpr (WGRAD) {
        tile_transpose(scr, h2, zt, lane);                              // dz2[out 32 wave + c][sample row(q, h)]
        float sb = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) sb += zt[q];
        sb += __shfl_xor(sb, 32, 64);
        if (h == 0) padd(P + G::DB2 + 32 * wave + c, sb, first);
        tile_transpose_inplace(scr, h1, lane);                          // h1[in 32 wave + c][sample row(q, h)]
        put_tile(xch + X::H1T + wave * XT + lane * XW, h1);
    }
    if constexpr (WGRAD) {
        // the tile's input rows -> the wave's sample cache (dW1's operand).  Here, not at the top of the tile: the rows are an index -> row
        // gather, two dependent memory round trips that nothing in front of this point has to wait for
        constexpr int L1Q = 4 * ((NIN + 7) / 8);
        cache_x<NIN, L1Q>(xs, xin, c, h);
    }
    if constexpr (TL >= 0) SCG_S_STAMP(TL, 5);
    __syncthreads();
    if constexpr (TL >= 0) SCG_S_STAMP(TL, 6);
    // data gradient of the own input tile: dh1[32 wave + .] = sum over rho of W2[32 rho + ., 32 wave + .]^T dz2[rho]
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
#pragma unroll
    for (int rho = 0; rho < NT; ++rho) {
        float za[16];
        get_tile(xch + X::DZX + rho * XT + lane * XW, za);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if constexpr (WGRAD) acc = mfma32(za[q], bt[rho][q], acc);  // transposed: [sample row(q', h)][feature 32 wave + c]
            else acc = mfma32(bt[rho][q], za[q], acc);                  // plain:      [feature row(q', h)][sample c]
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] *= mlp_dact<ACT>(h1[q]);       // h1 is transposed exactly when acc is
    if constexpr (TL >= 0) SCG_S_STAMP(TL, 7);
    if constexpr (DIN) {
        // d loss / d (action inputs): this wave's 32 features, then the waves' partials through the LDS
        static_assert(!WGRAD, "the input gradient is taken from the plain data gradient");
        float sj[NU], xj[NU];
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(sm + S::W1A + j * HID + 32 * wave + 8 * g + 4 * h);
                s = __builtin_fmaf(wv.x, acc[4 * g], s); s = __builtin_fmaf(wv.y, acc[4 * g + 1], s);
                s = __builtin_fmaf(wv.z, acc[4 * g + 2], s); s = __builtin_fmaf(wv.w, acc[4 * g + 3], s);
            }
            sj[j] = s;
        }
#pragma unroll
        for (int j = 0; j < NU; ++j) xj[j] = __shfl_xor(sj[j], 32, 64);
        if (h == 0) {
#pragma unroll
            for (int j = 0; j < NU; ++j) xch[X::DIN + (wave * 4 + j) * 32 + c] = sj[j] + xj[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < NT; ++w) s += xch[X::DIN + (w * 4 + j) * 32 + c];
            din[j] = s;
        }
    }
    if constexpr (WGRAD) {
        // dW1 | db1 slice: [dz1 tile (own 32 features x 32 samples)] x [x | 1]; D[feature row(q', h)][column c]
        {
            const float* const xrow = xs + (c < NIN + 1 ? c : NIN + 1) * 32 + 4 * h;
            float xb[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xrow + 8 * g);
                xb[4 * g] = v.x; xb[4 * g + 1] = v.y; xb[4 * g + 2] = v.z; xb[4 * g + 3] = v.w;
            }
            f32x16 g1;
#pragma unroll
            for (int q = 0; q < 16; ++q) g1[q] = 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) g1 = mfma32(acc[q], xb[q], g1);
            if (c <= NIN) {
                const uint32_t dw = G::DW1 + c * HID + 32 * wave + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {g1[4 * g], g1[4 * g + 1], g1[4 * g + 2], g1[4 * g + 3]};
                    if (!first) v += *reinterpret_cast<const f32x4*>(P + dw + 8 * g);
                    store_wt(P, dw + 8 * g, v);
                }
            }
        }
        if constexpr (TL >= 0) SCG_S_STAMP(TL, 8);
        // dW2 tiles (in 32 tau.., out 32 wave..) = h1^T[tau] x dz2^T[own] over this tile's samples
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            float ta[16];
            get_tile(xch + X::H1T + tau * XT + lane * XW, ta);
            f32x16 d2;
#pragma unroll
            for (int q = 0; q < 16; ++q) d2[q] = 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) d2 = mfma32(ta[q], zt[q], d2);
            const uint32_t dw = G::DW2 + ((tau * NT + wave) * 4 * 64 + lane) * 4;        // [tile][g][lane][4]: 1 KB of consecutive addresses per store
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {d2[4 * g], d2[4 * g + 1], d2[4 * g + 2], d2[4 * g + 3]};
                if (!first) v += *reinterpret_cast<const f32x4*>(P + dw + 256 * g);
                store_wt(P, dw + 256 * g, v);
            }
        }
    }
}

// ---- a wave's activation tiles across launches.  The forward pass that FEEDS a gradient kernel runs in the launch before it (the actor
// at obs: actor_fwd_kernel; the critics at (obs, act): q_kernel<0>'s online blocks), which leaves every wave's h1 / h2 tile — accumulator
// layout, read back by the same (tile, wave, lane) — in the workspace: [tile][wave][g][lane][4], 16-byte accesses, 1 KB per instruction.
// The gradient kernels then START at the loss derivatives: no operand loads for the forward products, no forward pass, no tanh-Gaussian
// algebra on their critical path (round 6's timeline of actor_grad_kernel: 4.8 + 1.5 of the 18.9 us a wave lived).
__device__ __forceinline__ void act_store(float* __restrict__ base, int tile, int wave, int lane, const f32x16& t) {
    const uint32_t word = (uint32_t)((tile * NT + wave) * 4 * 64 + lane) * 4u;
#pragma unroll
    for (int g = 0; g < 4; ++g) store_wt(base, word + 256u * g, (f32x4){t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]});
}
__device__ __forceinline__ void act_load(const float* __restrict__ base, int tile, int wave, int lane, f32x16& t) {
    const float* const p = base + (((size_t)tile * NT + wave) * 4 * 64 + lane) * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + g * 256);
        t[4 * g] = v.x; t[4 * g + 1] = v.y; t[4 * g + 2] = v.z; t[4 * g + 3] = v.w;
    }
}
constexpr int HEADW = 3 * NU;                   // per batch row, [j][B]: tanh u | sigma | clamp pass (what the policy gradient needs of the head)

// ---- kernels: gridDim.x workgroups of NT waves walk the 32-row tiles
// One actor forward job: rows `idx` (or, with idx_out, DRAWN here ~ U[0, ring size) and kept for the later launches) of `src`; noise,
// action, log pi per batch row; optionally (head / h1s / h2s) what actor_grad_kernel needs of this pass.
struct AfJob {
    const float* src;                           // obs (policy loss) or next_obs (target action)
    const int32_t* idx; int32_t* idx_out; const int32_t* ring_size; const int32_t* idx_in;
    const float* eps_in; uint32_t stream; uint32_t cnt_add;
    float* eps_out; float* a_out; float* logp_out;
    float* la_out;                              // nullable: snapshot of log_alpha as the policy loss sees it
    float* head; float* h1s; float* h2s;        // nullable
};
// blockIdx.y selects the job.  Two jobs in one launch (scg_sac_update_n): the target action of step k (next_obs, the actor step k just
// updated) and the policy-loss action of step k + 1 (obs, the same actor — the critics' step in between does not touch it): 2 x 128
// workgroups fill the chip where each launch alone left half of it idle, and step k + 1 starts at q_kernel<1>.
__global__ __launch_bounds__(64 * NT, 2) void actor_fwd_kernel(const float* __restrict__ params, const scg_mlp_layout lay, const Common Cm,
                                                                const AfJob J0, const AfJob J1) {
    constexpr int L1Q = 4 * ((NOBS + 7) / 8);
    extern __shared__ __align__(16) float lds[];
    const AfJob& J = blockIdx.y ? J1 : J0;
    const MlpWeights w = weights_of(params, lay);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const int n_tiles = Cm.batch / 32, B = Cm.batch;
    const float* const src = J.src;
    const float* const eps_in = J.eps_in;
    int32_t* const idx_out = J.idx_out;
    const uint32_t cnt = *Cm.counter + J.cnt_add;
    // request order (see small_load): the first tile's row index (gridDim.x <= n_tiles: unconditional), the small block, the layer
    // operands, the row itself.  A job that DRAWS its rows computes the index behind the operand requests: the Philox block waits for the
    // counter word, and in front of them that wait was a whole memory round trip with nothing else in flight.
    const int tile0 = blockIdx.x, r0 = tile0 * 32 + c;
    int s0 = 0;
    if (!idx_out) s0 = J.idx[r0];
    SmallRegs<NA> sr;
    small_load<NA>(sr, w, threadIdx.x);
    float a1[L1Q], a2[NT][16];
    load_a1<NOBS, L1Q>(w.W1, wave, lane, a1);
    load_a2(w.W2, wave, lane, a2);
    if (idx_out) s0 = sample_row(r0, J.ring_size, J.idx_in, cnt, Cm.k0, Cm.k1);
    float x[L1Q], eps[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    load_x2<L1Q, NOBS, 0>(src + (size_t)s0 * NOBS, nullptr, h, x);
    if (eps_in) {
#pragma unroll
        for (int j = 0; j < NU; ++j) eps[j] = eps_in[(size_t)r0 * NU + j];
    }
    small_store<NA>(lds, sr, threadIdx.x);
    float* const xch = lds + Small<NA>::END;
    if (J.la_out && blockIdx.x == 0 && threadIdx.x == 0) *J.la_out = *Cm.log_alpha;
    __syncthreads();
    for (int tile = tile0; tile < n_tiles; tile += gridDim.x) {
        const int r = tile * 32 + c;
        int s = s0;
        if (tile != tile0) {
            s = idx_out ? sample_row(r, J.ring_size, J.idx_in, cnt, Cm.k0, Cm.k1) : J.idx[r];
            load_x2<L1Q, NOBS, 0>(src + (size_t)s * NOBS, nullptr, h, x);
            if (eps_in) {
#pragma unroll
                for (int j = 0; j < NU; ++j) eps[j] = eps_in[(size_t)r * NU + j];
            }
        }
        if (idx_out && wave == 0 && h == 0) idx_out[r] = s;
        f32x16 h1, h2;
        float out[NA], u[NU], th[NU], sig[NU], a[NU], logp;
        forward<NOBS, NA, MLP_ACT_NONE>(lds, xch, a1, a2, x, wave, lane, h1, h2, out);
        if (J.h1s) { act_store(J.h1s, tile, wave, lane, h1); act_store(J.h2s, tile, wave, lane, h2); }
        if (wave == 0) {
            if (!eps_in) normal4(cnt, (uint32_t)r, J.stream, Cm.k0, Cm.k1, eps);
            squash(out, eps, Cm.low, Cm.high, u, th, sig, a, logp);
            if (h == 0) {
#pragma unroll
                for (int j = 0; j < NU; ++j) { J.eps_out[(size_t)r * NU + j] = eps[j]; J.a_out[(size_t)r * NU + j] = a[j]; }
                J.logp_out[r] = logp;
                if (J.head) {
#pragma unroll
                    for (int j = 0; j < NU; ++j) {
                        const float raw = out[NU + j];
                        J.head[(size_t)j * B + r] = th[j];
                        J.head[(size_t)(NU + j) * B + r] = sig[j];
                        J.head[(size_t)(2 * NU + j) * B + r] = (raw >= -20.0f && raw <= 2.0f) ? 1.0f : 0.0f;      // torch.clamp's gradient
                    }
                }
            }
        }
    }
}

// Q networks.
//   MODE 0: forward only, blockIdx.y = y + 2 * online:
//             online = 0: TARGET network y at (next_obs[idx], a_in[row])          -> q_out[y][row]
//             online = 1: ONLINE network y at (obs[idx], act[idx])                -> qo[y][row] and the waves' h1 / h2 tiles, for MODE 2
//           (the second kind does not depend on the first: it is the forward half of the critic loss, run here next to the targets
//            instead of behind them)
//   MODE 1: online networks at (obs[idx], a_in[row]), data gradient       -> q_out[y][row], dqda[y][row][NU]; blockIdx.y = y
//   MODE 2: backward only, blockIdx.y = y: d mean (q - y)^2 / d(theta_y) into the workgroups' partials, q = qo[y][row] and the stored
//           tiles, y = rew + gamma mask (min(qt1, qt2) - alpha logp_next)
struct QAct { float* qo; float* h1s; float* h2s; float* rew; float* mask; };     // [2][B], [2][B / 32 tiles][NT][1024] each, [B], [B]
template <int MODE>
__global__ __launch_bounds__(64 * NT, MODE == 0 ? 2 : 1) void q_kernel(const float* __restrict__ params, const float* __restrict__ params_online,
                                                        const scg_mlp_layout lay1, const scg_mlp_layout lay2,
                                                        const Common Cm, const float* __restrict__ a_in, const float* __restrict__ qt,
                                                        const float* __restrict__ logp_next, float* __restrict__ q_out,
                                                        float* __restrict__ dqda, const QAct QA, float* __restrict__ partials) {
    using S = Small<1>;
    using G = Part<NQ, 1>;
    constexpr int L1Q = 4 * ((NQ + 7) / 8);
    extern __shared__ __align__(16) float lds[];
    const int y = blockIdx.y & 1;
    const bool online = MODE == 0 && (blockIdx.y >> 1);
    const MlpWeights w = weights_of(online ? params_online : params, y ? lay2 : lay1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const int n_tiles = Cm.batch / 32, B = Cm.batch;
    const int tile0 = blockIdx.x, r0 = tile0 * 32 + c;
    const int s0 = Cm.idx[r0];                                          // (gridDim.x <= n_tiles; unconditional: see actor_grad_kernel)
    SmallRegs<1> sr;
    small_load<1>(sr, w, threadIdx.x);
    float* const xch = lds + S::END;
    if constexpr (MODE == 2) {
        // ---- backward only.  Request order: row index, small block, the tiles and the row's values, the data-gradient operand
        float bt[NT][16];
        const float* const h1s = QA.h1s + (size_t)y * B * HID;
        const float* const h2s = QA.h2s + (size_t)y * B * HID;
        f32x16 h1, h2;
        float x[L1Q], v_rew = 0.0f, v_mask = 0.0f, v_qt1 = 0.0f, v_qt2 = 0.0f, v_lpn = 0.0f, v_q = 0.0f;
        // (reward and mask by BATCH row, left by q_kernel<0>'s online blocks: nothing the loss derivative needs waits for the row index)
        auto load_row = [&](int tile, int r) {
            act_load(h1s, tile, wave, lane, h1); act_load(h2s, tile, wave, lane, h2);
            v_q = QA.qo[(size_t)y * B + r]; v_qt1 = qt[r]; v_qt2 = qt[B + r]; v_lpn = logp_next[r];
            v_rew = QA.rew[r]; v_mask = QA.mask[r];
        };
        auto load_in = [&](int s) { load_x2<L1Q, NOBS, NU>(Cm.obs + (size_t)s * NOBS, Cm.act + (size_t)s * NU, h, x); };
        load_row(tile0, r0);
        load_bt(w.W2, wave, lane, bt);                                  // (see actor_grad_kernel)
        load_in(s0);
        small_store<1>(lds, sr, threadIdx.x);
        __syncthreads();
        float* const P = partials + ((size_t)y * Cm.n_part + blockIdx.x) * PSTRIDE;
        const float alpha = expf(*Cm.log_alpha);
        const float inv_b = 1.0f / (float)B;
        float st = 0.0f;
        bool first = true;
        for (int tile = tile0; tile < n_tiles; tile += gridDim.x) {
            const int r = tile * 32 + c;
            if (tile != tile0) { load_row(tile, r); load_in(Cm.idx[r]); }
            const float target = v_rew + Cm.gamma * v_mask * (fminf(v_qt1, v_qt2) - alpha * v_lpn);
            const float e = v_q - target;
            const float dout[1] = {2.0f * e * inv_b};
            if (wave == 0 && h == 0) st += e * e * inv_b;
            backward<NQ, 1, ACT, true, false>(lds, xch, bt, h1, h2, dout, wave, lane, P, first, nullptr, x);
            first = false;
            __syncthreads();                                            // the exchange buffers are free for the next tile
        }
        if (wave == 0) {                                                // (st lives in the lanes of half 0)
            const float v = row_sum32(xch + Xch::WAVE + TR_WORDS + 34 * 32, st, lane);
            if (lane == 0) { P[G::STAT] = v; P[G::STAT + 1] = 0.0f; }
        }
    } else {
        // request order (see small_load): the first tile's row index, the small block, the forward operands, the row's values;
        // the backward operand (bt) is requested behind the barrier and arrives under the forward pass
        float w1a[(NU * HID + 64 * NT - 1) / (64 * NT)];
        if constexpr (MODE == 1) {                                      // W1A[j][f] = W1[f][NOBS + j]
#pragma unroll
            for (int j = 0; j < (NU * HID + 64 * NT - 1) / (64 * NT); ++j) {
                const int k = threadIdx.x + j * 64 * NT;
                w1a[j] = k < NU * HID ? w.W1[(size_t)(k % HID) * NQ + NOBS + k / HID] : 0.0f;
            }
        }
        float a1[L1Q], a2[NT][16];
        load_a1<NQ, L1Q>(w.W1, wave, lane, a1);
        load_a2(w.W2, wave, lane, a2);
        float x[L1Q], v_rew = 0.0f, v_mask = 0.0f;
        auto load_row = [&](int r, int s) {
            if constexpr (MODE == 0) {
                if (online) {
                    load_x2<L1Q, NOBS, NU>(Cm.obs + (size_t)s * NOBS, Cm.act + (size_t)s * NU, h, x);
                    if (y == 0) { v_rew = Cm.rew[s]; v_mask = Cm.mask[s]; }
                } else load_x2<L1Q, NOBS, NU>(Cm.next_obs + (size_t)s * NOBS, a_in + (size_t)r * NU, h, x);
            } else {
                load_x2<L1Q, NOBS, NU>(Cm.obs + (size_t)s * NOBS, a_in + (size_t)r * NU, h, x);
            }
        };
        load_row(r0, s0);
        small_store<1>(lds, sr, threadIdx.x);
        if constexpr (MODE == 1) {
#pragma unroll
            for (int j = 0; j < (NU * HID + 64 * NT - 1) / (64 * NT); ++j) {
                const int k = threadIdx.x + j * 64 * NT;
                if (k < NU * HID) lds[S::W1A + k] = w1a[j];
            }
        }
        __syncthreads();
        float bt[MODE == 1 ? NT : 1][16];
        if constexpr (MODE == 1) load_bt(w.W2, wave, lane, bt);
        for (int tile = tile0; tile < n_tiles; tile += gridDim.x) {
            const int r = tile * 32 + c;
            if (tile != tile0) load_row(r, Cm.idx[r]);
            f32x16 h1, h2;
            float out[1];
            forward<NQ, 1, ACT>(lds, xch, a1, a2, x, wave, lane, h1, h2, out);
            if constexpr (MODE == 0) {
                if (online) {
                    act_store(QA.h1s + (size_t)y * B * HID, tile, wave, lane, h1);
                    act_store(QA.h2s + (size_t)y * B * HID, tile, wave, lane, h2);
                    if (wave == 0 && h == 0) QA.qo[(size_t)y * B + r] = out[0];
                    if (wave == 1 % NT && h == 0 && y == 0) { QA.rew[r] = v_rew; QA.mask[r] = v_mask; }
                } else if (wave == 0 && h == 0) {
                    q_out[(size_t)y * B + r] = out[0];
                }
            } else {
                const float dout[1] = {1.0f};
                float din[NU];
                backward<NQ, 1, ACT, false, true>(lds, xch, bt, h1, h2, dout, wave, lane, nullptr, true, din);
                if (wave == 0 && h == 0) {
                    q_out[(size_t)y * B + r] = out[0];
#pragma unroll
                    for (int j = 0; j < NU; ++j) dqda[((size_t)y * B + r) * NU + j] = din[j];
                }
                __syncthreads();                                        // the exchange buffers are free for the next tile
            }
        }
    }
}

// actor gradient of policy_loss = mean(alpha log pi - min(q1, q2)(obs, a)), from the pass actor_fwd_kernel left behind: the waves' h1 / h2
// tiles, the head's (tanh u, sigma, clamp pass), log pi and the noise.  Request order: row index, small block (W3 for dz2), the tiles,
// the row's values, the observation (dW1's operand), the data-gradient operand.
__global__ __launch_bounds__(64 * NT, 1) void actor_grad_kernel(const float* __restrict__ params, const scg_mlp_layout lay, const Common Cm,
                                                                 const float* __restrict__ eps_all, const float* __restrict__ qpi,
                                                                 const float* __restrict__ dqda, const float* __restrict__ head,
                                                                 const float* __restrict__ logp_all, const float* __restrict__ h1s,
                                                                 const float* __restrict__ h2s, float* __restrict__ partials) {
    using S = Small<NA>;
    using G = Part<NOBS, NA>;
    constexpr int L1Q = 4 * ((NOBS + 7) / 8);
    extern __shared__ __align__(16) float lds[];
    const MlpWeights w = weights_of(params, lay);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const int n_tiles = Cm.batch / 32, B = Cm.batch;
    SCG_S_STAMP(0, 0);
    const int tile0 = blockIdx.x, r0 = tile0 * 32 + c;
    // (unconditional — gridDim.x <= n_tiles: written as `tile0 < n_tiles ? idx[r0] : 0` the row's byte offset was computed in the branch
    //  that held the load, behind an s_waitcnt vmcnt(0): one whole memory round trip at kernel entry before anything else was requested)
    const int s0 = Cm.idx[r0];
    SmallRegs<NA> sr;
    small_load<NA>(sr, w, threadIdx.x);
    float bt[NT][16];
    f32x16 h1, h2;
    float x[L1Q], eps[NU], th[NU], sig[NU], pass[NU], logp = 0.0f, q1 = 0.0f, q2 = 0.0f, dq[2][NU];
    auto load_row = [&](int tile, int r) {
        act_load(h1s, tile, wave, lane, h1); act_load(h2s, tile, wave, lane, h2);
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            eps[j] = eps_all[(size_t)r * NU + j];
            th[j] = head[(size_t)j * B + r]; sig[j] = head[(size_t)(NU + j) * B + r]; pass[j] = head[(size_t)(2 * NU + j) * B + r];
            dq[0][j] = dqda[(size_t)r * NU + j];
            dq[1][j] = dqda[((size_t)B + r) * NU + j];
        }
        q1 = qpi[r]; q2 = qpi[B + r]; logp = logp_all[r];
    };
    auto load_in = [&](int s) { load_x2<L1Q, NOBS, 0>(Cm.obs + (size_t)s * NOBS, nullptr, h, x); };
    load_row(tile0, r0);
    // (the data-gradient operand behind the tiles and the row values: 16 NT one-word loads per lane, needed only behind the tile's workgroup
    //  barrier — in front of them they filled the 63-deep memory counter and what is needed first was issued a round trip late; the
    //  observation last: it waits for the row index, and nothing needs it before the sample cache is written in backward())
    load_bt(w.W2, wave, lane, bt);
    load_in(s0);
    small_store<NA>(lds, sr, threadIdx.x);
    __syncthreads();
    SCG_S_STAMP(0, 1);
    float* const xch = lds + S::END;
    float* const P = partials + (size_t)blockIdx.x * PSTRIDE;
    const float alpha = expf(*Cm.log_alpha), inv_b = 1.0f / (float)B;
    float st_loss = 0.0f, st_logp = 0.0f;
    bool first = true;
    for (int tile = tile0; tile < n_tiles; tile += gridDim.x) {
        const int r = tile * 32 + c;
        if (tile != tile0) { load_row(tile, r); load_in(Cm.idx[r]); }
        SCG_S_STAMP(0, 2);
        float dout[NA];
        const int ysel = q2 < q1 ? 1 : 0;                                          // torch.min: gradient to the smaller (q1 on a tie)
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const float dqj = ysel ? dq[1][j] : dq[0][j];
            const float du = inv_b * (alpha * 2.0f * th[j] - dqj * 0.5f * (Cm.high[j] - Cm.low[j]) * (1.0f - th[j] * th[j]));
            dout[j] = du;
            dout[NU + j] = pass[j] * (du * sig[j] * eps[j] - inv_b * alpha);
        }
        if (wave == 0 && h == 0) { st_loss += (alpha * logp - fminf(q1, q2)) * inv_b; st_logp += logp * inv_b; }
        SCG_S_STAMP(0, 3);
        backward<NOBS, NA, MLP_ACT_NONE, true, false, 0>(lds, xch, bt, h1, h2, dout, wave, lane, P, first, nullptr, x);
        SCG_S_STAMP(0, 9);
        first = false;
        __syncthreads();
        SCG_S_STAMP(0, 10);
    }
    if (wave == 0) {                                                    // (the sums live in the lanes of half 0)
        float* const r = xch + Xch::WAVE + TR_WORDS + 34 * 32;          // wave 0's dout_l rows: free behind the last tile's barrier
        if (lane < 32) { r[lane] = st_loss; r[32 + lane] = st_logp; }
        wave_sync();
        if (lane < 2) {
            float v = 0.0f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(r + 32 * lane + 4 * g);
                v += d4.x; v += d4.y; v += d4.z; v += d4.w;
            }
            P[G::STAT + lane] = v;
        }
    }
    SCG_S_STAMP(0, 11);
}

}  // namespace wide

// Sum of the waves' partials -> flat gradient (torch parameter order); blockIdx.y = network of the launch.
template <int NIN, int NOUT>
__device__ __forceinline__ int dest_of(int k, const scg_mlp_layout& lay) {
    using G = Part<NIN, NOUT>;
    if (k < G::DB2) { const int in = k / HID, o = k % HID; return in < NIN ? lay.W1 + o * NIN + in : lay.b1 + o; }
    if (k < G::DW3) return lay.b2 + (k - G::DB2);
    if (k < G::DB3) return lay.W3 + (k - G::DW3);
    if (k < G::STAT) return (k - G::DB3) < NOUT ? lay.b3 + (k - G::DB3) : -1;
    if (k < G::DW2) return -2 - (k - G::STAT);
    const int p = k - G::DW2;                               // [tile][g][lane][4]
    const int q = 4 * ((p >> 8) & 3) + (p & 3), lane = (p >> 2) & 63, tr = p >> 10;
    const int tau = tr / NT, rho = tr % NT;
    return lay.W2 + (32 * rho + (lane & 31)) * HID + 32 * tau + d_row(q, lane >> 5);
}

struct ReduceArgs {
    const float* partials; int n_part; scg_mlp_layout lay[2]; float* grad;
    float* stat_out;            // [2 * gridDim.y]: STAT + 0, STAT + 1 of each network
    int alpha_slot;             // >= 0 (actor reduce): grad[alpha_slot] = d entropy_loss / d log_alpha = -(mean log pi + target_entropy)
    float target_entropy;
    // p != nullptr (single-GPU path): the element's torch.optim.Adam step and its soft update follow its sum at once — each
    // parameter is written by exactly one thread of one launch, so the separate adam_kernel launch and the trip of the
    // gradient through memory go away.  Data-parallel callers leave p null, all-reduce grad and run adam_kernel.
    float* p; float* m; float* v; float lr; const float* steps; float* steps_rw; int step_slot; float* target; float tau;
    int alpha_on; float lr_alpha;
    // Single-GPU bookkeeping folded into the reductions (no finish_kernel launch; no device-scope fence either — every word below is
    // touched by ONE thread of a launch in which nobody else reads it):
    //   bump_critic  (actor's launch)    the statistics word's owner pre-increments steps[1], which only the CRITICS' launch reads
    //   t_add        what a launch adds to steps[step_slot] to get Adam's t: 1, or 0 where the count was pre-incremented
    //   fin          (critics' launch)   the owner of network 0's statistics word also sums network 1's (same fixed order as that
    //                                    network's own block), advances steps[0], steps[2] and the Philox counter (read by other
    //                                    launches only) and writes the step's loss statistics
    int bump_critic; float t_add;
    struct Fin { float* steps; uint32_t* counter; float* stats; float* stats_acc; const float* actor_stat; const float* log_alpha_before;
                 int alpha_on; float target_entropy; } fin;
};
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr, float t);
template <int NIN, int NOUT>
__global__ __launch_bounds__(256) void reduce_kernel(const ReduceArgs R) {
    __shared__ float part[4][64];
    const int net = blockIdx.y;
    const int kl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kl;
    constexpr int words = Part<NIN, NOUT>::END;
    // Request order: what the word's OWNER needs for its optimiser step (parameter, moments, target copy, step count) goes out first and
    // arrives under the partial loads — asked for behind the sum it was one more memory round trip at the end of every block; then ALL
    // of this thread's partial words at once (32 at n_part = 128; `unroll 8` made that four rounds of eight).  The sum keeps its order.
    const bool owner = grp == 0 && k < words;
    int d = -1;
    float o_p = 0.0f, o_m = 0.0f, o_v = 0.0f, o_t = 0.0f, o_steps = 0.0f;
    if (owner) {
        d = dest_of<NIN, NOUT>(k, R.lay[net]);
        if (d >= 0 && R.p) {
            o_p = R.p[d]; o_m = R.m[d]; o_v = R.v[d]; o_steps = R.steps[R.step_slot];
            if (R.target) o_t = R.target[d];
        } else if (d == -2 && R.bump_critic) {
            o_steps = R.steps_rw[1];
        } else if (d == -3 && R.alpha_slot >= 0 && R.p && R.alpha_on) {                 // the temperature's step
            const int a = R.alpha_slot;
            o_p = R.p[a]; o_m = R.m[a]; o_v = R.v[a]; o_steps = R.steps[2];
        }
    }
    float s = 0.0f;
    if (k < words) {
        const float* const src = R.partials + ((size_t)net * R.n_part + grp) * PSTRIDE + k;      // partials grp, grp + 4, ...
        const int mine = (R.n_part - grp + 3) / 4;
        for (int g0 = 0; g0 < mine; g0 += 32) {
            float v[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) v[j] = g0 + j < mine ? src[(size_t)(g0 + j) * 4 * PSTRIDE] : 0.0f;
#pragma unroll
            for (int j = 0; j < 32; ++j) { if (g0 + j < mine) s += v[j]; }
        }
    }
    part[grp][kl] = s;
    __shared__ float other[512];                            // (n_part <= 512, n_part_of)
    constexpr int KS = Part<NIN, NOUT>::STAT;
    const bool fin_block = R.fin.steps && net == 0 && blockIdx.x == KS / 64;
    // the bookkeeping thread's inputs, requested up here: behind its sum they would be three dependent memory round trips at the end of
    // the launch's longest-running thread
    float f_steps0 = 0.0f, f_steps2 = 0.0f, f_pl = 0.0f, f_ml = 0.0f, f_la = 0.0f, f_acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t f_cnt = 0u;
    if (fin_block) {                                        // network 1's statistics word of every partial: one load per thread
        for (int g = threadIdx.x; g < R.n_part; g += 256) other[g] = R.partials[((size_t)R.n_part + g) * PSTRIDE + KS];
        if (grp == 0 && k == KS) {
            const ReduceArgs::Fin& F = R.fin;
            f_steps0 = F.steps[0]; f_steps2 = F.steps[2]; f_cnt = *F.counter; f_pl = F.actor_stat[0]; f_ml = F.actor_stat[1];
            f_la = *F.log_alpha_before;
            if (F.stats_acc) { f_acc[0] = F.stats_acc[0]; f_acc[1] = F.stats_acc[1]; f_acc[2] = F.stats_acc[2]; f_acc[3] = F.stats_acc[3]; }
        }
    }
    __syncthreads();
    if (!owner) return;
    s = (part[0][kl] + part[1][kl]) + (part[2][kl] + part[3][kl]);
    if (d >= 0) {
        R.grad[d] = s;
        if (R.p) {
            adam_one(o_p, s, o_m, o_v, R.lr, o_steps + R.t_add);
            R.p[d] = o_p; R.m[d] = o_m; R.v[d] = o_v;
            if (R.target) R.target[d] = polyak(o_t, o_p, R.tau);
        }
    } else if (d == -2) {
        R.stat_out[2 * net] = s;
        if (R.bump_critic) R.steps_rw[1] = o_steps + 1.0f;
        if (fin_block) {                                    // what finish_kernel does on the data-parallel path
            const ReduceArgs::Fin& F = R.fin;
            F.steps[0] = f_steps0 + 1.0f;
            if (F.alpha_on) F.steps[2] = f_steps2 + 1.0f;
            *F.counter = f_cnt + 1u;
            const float pl = f_pl, ml = f_ml;
            float o4[4];                                    // summed exactly as network 1's own block sums its word
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float o = 0.0f;
                for (int g = q; g < R.n_part; g += 4) o += other[g];
                o4[q] = o;
            }
            const float cl = s + ((o4[0] + o4[1]) + (o4[2] + o4[3]));
            const float el = F.alpha_on ? -f_la * (ml + F.target_entropy) : 0.0f;
            F.stats[0] = pl; F.stats[1] = cl; F.stats[2] = el; F.stats[3] = ml;
            if (F.stats_acc) { F.stats_acc[0] = f_acc[0] + pl; F.stats_acc[1] = f_acc[1] + cl; F.stats_acc[2] = f_acc[2] + el; F.stats_acc[3] = f_acc[3] + ml; }
        }
    } else if (d == -3) {
        R.stat_out[2 * net + 1] = s;
        // entropy_loss = -mean(log_alpha (log pi + target_entropy)) (sac_utils.py:124-126); in the gradient vector so that a
        // data-parallel all-reduce of d_grad carries it
        if (R.alpha_slot >= 0) {
            const float g = -(s + R.target_entropy);
            const int a = R.alpha_slot;
            R.grad[a] = g;
            if (R.p && R.alpha_on) {
                adam_one(o_p, g, o_m, o_v, R.lr_alpha, o_steps + 1.0f);
                R.p[a] = o_p; R.m[a] = o_m; R.v[a] = o_v;
            }
        }
    }
}

// torch.optim.Adam (betas 0.9 / 0.999, eps 1e-8) on elements [lo, hi) of the flat vectors; optionally the temperature
// (element n_params, gradient from the mean log pi) and the Polyak update of the target copy over [0, n_polyak).
struct AdamArgs {
    float* p; const float* g; float* m; float* v; int lo, hi; float lr; const float* steps; int step_slot;
    int alpha_on; int n_params; float lr_alpha; float target_entropy; const float* actor_stat;
    float* target; int n_polyak; float tau;
};
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr, float t) { adam_element(p, g, m, v, lr, t); }
__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;            // one element per thread: Adam (if in [lo, hi)), then its soft update
    if (i >= A.lo && i < A.hi) adam_one(A.p[i], A.g[i], A.m[i], A.v[i], A.lr, A.steps[A.step_slot] + 1.0f);
    if (A.alpha_on && i == 0) {
        const int k = A.n_params;                           // (gradient written by the actor's reduce_kernel)
        adam_one(A.p[k], A.g[k], A.m[k], A.v[k], A.lr_alpha, A.steps[2] + 1.0f);
    }
    if (A.target && i < A.n_polyak) A.target[i] = polyak(A.target[i], A.p[i], A.tau);
}

struct FinishArgs {
    float* steps; uint32_t* counter; float* stats; float* stats_acc; const float* actor_stat; const float* q_stat;
    const float* log_alpha_before; int alpha_on; float target_entropy;
};
// (Kept as its own one-thread launch.  Folding it into the critics' reduce_kernel as "the last workgroup to finish does the
//  bookkeeping" — a device-scope fence + one atomic per workgroup — was measured in round 4 (tools/sessions/s84.sh,
//  profiles/r04_kernel_stats_sac_iteration.csv): reduce_kernel<28, 1> 7.7 -> 64.8 us.  On this chip a device-scope release is
//  a write-back of the XCD's L2 (the 8 L2s are not coherent with each other), and 640 workgroups each paid one.)
__global__ void finish_kernel(const FinishArgs F) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    F.steps[0] += 1.0f; F.steps[1] += 1.0f;
    if (F.alpha_on) F.steps[2] += 1.0f;
    *F.counter += 1u;
    const float pl = F.actor_stat[0], ml = F.actor_stat[1];
    const float cl = F.q_stat[0] + F.q_stat[2];
    const float el = F.alpha_on ? -(*F.log_alpha_before) * (ml + F.target_entropy) : 0.0f;
    F.stats[0] = pl; F.stats[1] = cl; F.stats[2] = el; F.stats[3] = ml;
    if (F.stats_acc) { F.stats_acc[0] += pl; F.stats_acc[1] += cl; F.stats_acc[2] += el; F.stats_acc[3] += ml; }
}

// ------------------------------------------------------------------ host side
struct Ws {      // workspace carve-up (floats)
    size_t idx[2], eps, a_pi, logp, eps2, a_next, logp_next, qpi, dqda, qt, stat, la_before[2], head, ah1, ah2, qo, qh1, qh2, qrew, qmask, partials, total;
};
static Ws carve(int B, int n_part) {
    Ws w; size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o += (n + 63) / 64 * 64; return at; };
    // (two copies of the minibatch rows and of the log_alpha snapshot: scg_sac_update_n draws step k + 1's while step k still reads its own)
    w.idx[0] = take(B); w.idx[1] = take(B); w.eps = take((size_t)B * NU); w.a_pi = take((size_t)B * NU); w.logp = take(B);
    w.eps2 = take((size_t)B * NU); w.a_next = take((size_t)B * NU); w.logp_next = take(B);
    w.qpi = take(2 * (size_t)B); w.dqda = take(2 * (size_t)B * NU); w.qt = take(2 * (size_t)B);
    w.stat = take(8); w.la_before[0] = take(1); w.la_before[1] = take(1);
    w.head = take((size_t)wide::HEADW * B); w.ah1 = take((size_t)B * HID); w.ah2 = take((size_t)B * HID);
    w.qo = take(2 * (size_t)B); w.qh1 = take(2 * (size_t)B * HID); w.qh2 = take(2 * (size_t)B * HID); w.qrew = take(B); w.qmask = take(B);
    w.partials = take(2 * (size_t)n_part * PSTRIDE);
    w.total = o;
    return w;
}
static int n_part_of(int batch) { const int t = batch / 32; return t < 512 ? t : 512; }

extern "C" size_t scg_sac_workspace_bytes(int batch) {
    if (batch <= 0 || batch % 32) return 0;
    return carve(batch, n_part_of(batch)).total * sizeof(float);
}

template <typename K>
static int set_lds(K kernel, size_t bytes) {
    HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

static size_t lds_actor_bytes() { return (MlpLds<NOBS, HID, NA>::END + WAVES * (32 * ((NOBS + 3) / 4 * 4) + NA * 32 + TR_WORDS)) * sizeof(float); }
static size_t wide_lds_actor() { return (wide::Small<NA>::END + wide::Xch::END) * sizeof(float); }
static size_t wide_lds_q() { return (wide::Small<1>::END + wide::Xch::END) * sizeof(float); }
// forward-only launches need the small block, the h1 exchange and the output partials only: two workgroups per CU fit
static size_t wide_lds_actor_fwd() { return (wide::Small<NA>::END + wide::Xch::FWD_END) * sizeof(float); }
static size_t wide_lds_q_fwd() { return (wide::Small<1>::END + wide::Xch::FWD_END) * sizeof(float); }

// One-time kernel attributes (dynamic LDS above 64 KB).  Call once before capturing scg_sac_update into a HIP graph: the
// attribute calls are not stream operations.  scg_sac_update calls it itself otherwise.
extern "C" int scg_sac_prepare(void) {
    static scg::PerDeviceOnce once;         // per device (scg_once.h): call with the agent's device current
    int dev;
    if (!once.pending(&dev)) return 0;
    const size_t lds_a = lds_actor_bytes();
    if (lds_a > 160 * 1024 || wide_lds_actor() > 160 * 1024 || wide_lds_q() > 160 * 1024)
        return fail(-1, "scg_sac: network image does not fit the LDS");
    if (set_lds(actor_act_kernel, lds_a) || set_lds(actor_sample_kernel, MlpLds<NOBS, HID, NA>::END * sizeof(float))) return -2;
    if (set_lds(wide::actor_fwd_kernel, wide_lds_actor_fwd()) || set_lds(wide::actor_grad_kernel, wide_lds_actor()) ||
        set_lds(wide::q_kernel<0>, wide_lds_q_fwd()) || set_lds(wide::q_kernel<1>, wide_lds_q()) || set_lds(wide::q_kernel<2>, wide_lds_q())) return -2;
    once.commit(dev);
    return 0;
}

static int check_args(const scg_sac_args* a, const char* who) {
    if (!a || !a->d_params || !a->d_target || !a->d_grad || !a->d_m || !a->d_v || !a->d_steps || !a->d_obs || !a->d_act || !a->d_rew ||
        !a->d_next_obs || !a->d_mask || !a->d_counter || !a->d_workspace || !a->d_stats || (!a->d_ring_size && !a->d_idx_in))
        return fail(-1, std::string(who) + ": NULL argument");
    if (a->batch <= 0 || a->batch % 32) return fail(-1, std::string(who) + ": the batch size must be a positive multiple of 32");
    return 0;
}

// One gradient step (or the parts `phases` names).  parity: which copy of the minibatch rows / log_alpha snapshot the step uses;
// have_first: the step's first launch (rows drawn, a, log pi at obs) already ran as the second job of the previous step's target-action
// launch; with_next: this step's target-action launch carries that job for the NEXT step (scg_sac_update_n).
static int enqueue_step(const scg_sac_args* a, hipStream_t st, int phases, int parity, bool have_first, bool with_next) {
    const int B = a->batch, n_part = n_part_of(B);
    const Ws w = carve(B, n_part);
    float* W = (float*)a->d_workspace;
    int32_t* idx = (int32_t*)(W + w.idx[parity]);
    int32_t* idx_next = (int32_t*)(W + w.idx[parity ^ 1]);
    float* la_before = W + w.la_before[parity];
    const size_t wlds_a = wide_lds_actor(), wlds_q = wide_lds_q(), wlds_af = wide_lds_actor_fwd(), wlds_qf = wide_lds_q_fwd();
    Common Cm;
    Cm.idx = idx; Cm.batch = B; Cm.n_part = n_part; Cm.obs = a->d_obs; Cm.act = a->d_act; Cm.rew = a->d_rew; Cm.next_obs = a->d_next_obs;
    Cm.mask = a->d_mask; Cm.log_alpha = a->d_params + a->n_params; Cm.gamma = a->gamma;
    for (int j = 0; j < 4; ++j) { Cm.low[j] = a->act_low[j]; Cm.high[j] = a->act_high[j]; }
    Cm.k0 = (uint32_t)a->seed; Cm.k1 = (uint32_t)(a->seed >> 32); Cm.counter = a->d_counter;
    float* stat = W + w.stat;
    // the whole step on one GPU: the optimiser steps ride in the reduction launches; data-parallel callers (phases given one
    // by one, an all-reduce of d_grad between them) get the gradient only and step in adam_kernel
    const bool fuse = phases == SCG_SAC_ALL;
    auto optimiser = [&](ReduceArgs& R, float lr, int step_slot) {
        R.bump_critic = 0; R.t_add = 1.0f; R.steps_rw = nullptr;
        R.fin = ReduceArgs::Fin{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.0f};
        if (!fuse) { R.p = nullptr; R.m = R.v = R.target = nullptr; R.steps = nullptr; R.lr = R.tau = R.lr_alpha = 0.0f; R.step_slot = 0; R.alpha_on = 0; return; }
        R.p = a->d_params; R.m = a->d_m; R.v = a->d_v; R.lr = lr; R.steps = a->d_steps; R.steps_rw = a->d_steps; R.step_slot = step_slot;
        R.target = a->d_target; R.tau = a->tau; R.alpha_on = a->use_entropy_tuning; R.lr_alpha = a->entropy_lr;
        if (step_slot == 0) R.bump_critic = 1;              // actor's launch: pre-increment the critics' count (read by their launch only)
        else {                                              // critics' launch: the count is already this step's; + the step's bookkeeping
            R.t_add = 0.0f;
            R.fin = ReduceArgs::Fin{a->d_steps, a->d_counter, a->d_stats, a->d_stats_acc, stat, la_before, a->use_entropy_tuning, a->target_entropy};
        }
    };
    // the policy-loss job of a step: draws the minibatch rows (kept in idx), snapshots log_alpha as the policy loss sees it (entropy_loss
    // is reported with that value), a, log pi at obs + what actor_grad_kernel reads back of the pass
    auto policy_job = [&](int32_t* rows, float* la, uint32_t cnt_add) {
        return wide::AfJob{a->d_obs, nullptr, rows, a->d_ring_size, a->d_idx_in, a->d_eps_in, 1u, cnt_add, W + w.eps, W + w.a_pi, W + w.logp, la,
                           W + w.head, W + w.ah1, W + w.ah2};
    };
    const wide::QAct QA{W + w.qo, W + w.qh1, W + w.qh2, W + w.qrew, W + w.qmask};
    if (phases & SCG_SAC_ACTOR_GRAD) {
    // 1. minibatch rows, a, log pi at obs
    if (!have_first) {
        const wide::AfJob J = policy_job(idx, la_before, 0u);
        wide::actor_fwd_kernel<<<dim3(n_part, 1), dim3(64 * NT), wlds_af, st>>>(a->d_params, a->actor, Cm, J, J);
    }
    // 2. q1, q2 and dq/da at (obs, a)
    wide::q_kernel<1><<<dim3(n_part, 2), dim3(64 * NT), wlds_q, st>>>(a->d_params, nullptr, a->q1, a->q2, Cm, W + w.a_pi, nullptr, nullptr, W + w.qpi, W + w.dqda, QA, nullptr);
    // 3. actor gradient
    wide::actor_grad_kernel<<<dim3(n_part), dim3(64 * NT), wlds_a, st>>>(a->d_params, a->actor, Cm, W + w.eps, W + w.qpi, W + w.dqda, W + w.head, W + w.logp,
                                                                         W + w.ah1, W + w.ah2, W + w.partials);
    // 4. its sum (+ the actor and temperature steps and the soft update of the actor's target copy when fused)
    {
        ReduceArgs R; R.partials = W + w.partials; R.n_part = n_part; R.lay[0] = a->actor; R.lay[1] = a->actor; R.grad = a->d_grad; R.stat_out = stat;
        R.alpha_slot = a->n_params; R.target_entropy = a->target_entropy;
        optimiser(R, a->actor_lr, 0);
        reduce_kernel<NOBS, NA><<<dim3((Part<NOBS, NA>::END + 63) / 64, 1), dim3(256), 0, st>>>(R);
    }
    }
    if (phases & SCG_SAC_CRITIC_GRAD) {
    // 4'. actor (+ temperature) step after the caller's all-reduce
    if (!fuse) {
        AdamArgs A{a->d_params, a->d_grad, a->d_m, a->d_v, 0, a->n_actor, a->actor_lr, a->d_steps, 0,
                   a->use_entropy_tuning, a->n_params, a->entropy_lr, a->target_entropy, stat, nullptr, 0, 0.0f};
        adam_kernel<<<dim3((a->n_actor + 255) / 256), dim3(256), 0, st>>>(A);
    }
    // 5. a', log pi' at next_obs with the updated actor [+ the next step's launch 1: same actor, the counter word one ahead]
    {
        const wide::AfJob J{a->d_next_obs, idx, nullptr, nullptr, nullptr, a->d_eps_next_in, 2u, 0u, W + w.eps2, W + w.a_next, W + w.logp_next, nullptr,
                            nullptr, nullptr, nullptr};
        const wide::AfJob Jn = with_next ? policy_job(idx_next, W + w.la_before[parity ^ 1], 1u) : J;
        wide::actor_fwd_kernel<<<dim3(n_part, with_next ? 2 : 1), dim3(64 * NT), wlds_af, st>>>(a->d_params, a->actor, Cm, J, Jn);
    }
    // 6. target networks at (next_obs, a'); next to them the online critics' forward pass at (obs, act) for 7
    wide::q_kernel<0><<<dim3(n_part, 4), dim3(64 * NT), wlds_qf, st>>>(a->d_target, a->d_params, a->q1, a->q2, Cm, W + w.a_next, nullptr, nullptr, W + w.qt, nullptr, QA, nullptr);
    // 7. critic gradients
    wide::q_kernel<2><<<dim3(n_part, 2), dim3(64 * NT), wlds_q, st>>>(a->d_params, nullptr, a->q1, a->q2, Cm, nullptr, W + w.qt, W + w.logp_next, nullptr, nullptr, QA, W + w.partials);
    // 8. their sums (+ the critic steps and the soft update of their target copies when fused)
    {
        ReduceArgs R; R.partials = W + w.partials; R.n_part = n_part; R.lay[0] = a->q1; R.lay[1] = a->q2; R.grad = a->d_grad; R.stat_out = stat + 2;
        R.alpha_slot = -1; R.target_entropy = 0.0f;
        optimiser(R, a->critic_lr, 1);
        reduce_kernel<NQ, 1><<<dim3((Part<NQ, 1>::END + 63) / 64, 2), dim3(256), 0, st>>>(R);
    }
    }
    if (phases & SCG_SAC_FINISH) {
    // 8'. critic step + Polyak averaging of every actor-critic parameter after the caller's all-reduce
    if (!fuse) {
        AdamArgs A{a->d_params, a->d_grad, a->d_m, a->d_v, a->n_actor, a->n_params, a->critic_lr, a->d_steps, 1,
                   0, a->n_params, 0.0f, 0.0f, stat, a->d_target, a->n_params, a->tau};
        adam_kernel<<<dim3((a->n_params + 255) / 256), dim3(256), 0, st>>>(A);
    }
    // 9. step counters, loss statistics (data-parallel path; the fused single-GPU step did them inside the critics' reduction)
    if (!fuse) {
        FinishArgs F{a->d_steps, a->d_counter, a->d_stats, a->d_stats_acc, stat, stat + 2, la_before, a->use_entropy_tuning, a->target_entropy};
        finish_kernel<<<dim3(1), dim3(64), 0, st>>>(F);
    }
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int scg_sac_update(const scg_sac_args* a, void* stream) {
    if (int rc = check_args(a, "scg_sac_update")) return rc;
    if (int rc = scg_sac_prepare()) return rc;
    return enqueue_step(a, (hipStream_t)stream, a->phases == 0 ? SCG_SAC_ALL : a->phases, 0, false, false);
}

// n_steps whole gradient steps (a->phases must be 0 / SCG_SAC_ALL), results bit-identical to n_steps scg_sac_update calls, in
// 7 n_steps + 1 launches instead of 8 n_steps: the target-action launch of step k also runs step k + 1's first launch (see
// wide::actor_fwd_kernel) — nothing between the two touches the actor, the replay ring or the row draw's inputs.
extern "C" int scg_sac_update_n(const scg_sac_args* a, int n_steps, void* stream) {
    if (int rc = check_args(a, "scg_sac_update_n")) return rc;
    if (n_steps <= 0) return fail(-1, "scg_sac_update_n: n_steps must be positive");
    if (a->phases != 0 && a->phases != SCG_SAC_ALL) return fail(-1, "scg_sac_update_n: whole steps only (phases = 0)");
    if (int rc = scg_sac_prepare()) return rc;
    for (int k = 0; k < n_steps; ++k)
        if (int rc = enqueue_step(a, (hipStream_t)stream, SCG_SAC_ALL, k & 1, k > 0, k + 1 < n_steps)) return rc;
    return 0;
}

extern "C" int scg_sac_sample(const float* d_params, const scg_mlp_layout* actor, const float* act_low, const float* act_high, const float* d_obs,
                              int m, uint64_t seed, const uint32_t* d_counter, int uniform, const float* d_eps_in, float* d_act_out, void* stream) {
    if (!act_low || !act_high || !d_act_out || m <= 0 || (!uniform && (!d_params || !actor || !d_obs))) return fail(-1, "scg_sac_sample: bad argument");
    float lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
    for (int j = 0; j < NU; ++j) { lo[j] = act_low[j]; hi[j] = act_high[j]; }
    const float4 l4 = make_float4(lo[0], lo[1], lo[2], lo[3]), h4 = make_float4(hi[0], hi[1], hi[2], hi[3]);
    if (uniform) {
        uniform_action_kernel<<<dim3((m + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(m, l4, h4, (uint32_t)seed, (uint32_t)(seed >> 32), d_counter, d_act_out);
    } else {
        const size_t lds_a = MlpLds<NOBS, HID, NA>::END * sizeof(float);
        if (int rc = scg_sac_prepare()) return rc;
        const int grid = std::min(256, (m + 127) / 128);
        actor_sample_kernel<<<dim3(grid), dim3(64 * WAVES), lds_a, (hipStream_t)stream>>>(d_params, *actor, d_obs, m, l4, h4, (uint32_t)seed,
                                                                                            (uint32_t)(seed >> 32), d_counter, d_eps_in, d_act_out);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int scg_sac_push(const scg_sac_ring* ring, float* d_cur_obs, const float* d_act, const float* d_reward, const float* d_next_obs,
                            const float* d_terminal_obs, const uint8_t* d_done, const uint8_t* d_flags, int n, void* stream) {
    if (!ring || !ring->d_obs || !ring->d_act || !ring->d_rew || !ring->d_next_obs || !ring->d_mask || !ring->d_pos || !d_cur_obs || !d_act ||
        !d_reward || !d_next_obs || !d_terminal_obs || !d_done || !d_flags)
        return fail(-1, "scg_sac_push: NULL argument");
    if (n <= 0 || ring->capacity < n) return fail(-1, "scg_sac_push: replay capacity smaller than one vectorised step");
    const RingArgs R{ring->d_obs, ring->d_act, ring->d_rew, ring->d_next_obs, ring->d_mask, ring->capacity, (long long*)ring->d_pos, ring->d_size_f,
                     ring->d_size_i32, ring->d_counter};
    const int total = n * NOBS;
    ring_push_kernel<<<dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(R, d_cur_obs, d_act, d_reward, d_next_obs, d_terminal_obs, d_done, d_flags, n);
    ring_advance_kernel<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>(R, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int scg_sac_act(const float* d_params, const scg_mlp_layout* actor, const float* act_low, const float* act_high, const float* d_obs,
                           int m, float* d_act_out, void* stream) {
    if (!d_params || !actor || !act_low || !act_high || !d_obs || !d_act_out || m <= 0) return fail(-1, "scg_sac_act: bad argument");
    const size_t lds_a = MlpLds<NOBS, HID, NA>::END * sizeof(float);
    if (int rc = scg_sac_prepare()) return rc;
    float lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
    for (int j = 0; j < NU; ++j) { lo[j] = act_low[j]; hi[j] = act_high[j]; }
    const int grid = std::min(256, (m + 127) / 128);
    actor_act_kernel<<<dim3(grid), dim3(64 * WAVES), lds_a, (hipStream_t)stream>>>(d_params, *actor, d_obs, m, make_float4(lo[0], lo[1], lo[2], lo[3]),
                                                                                     make_float4(hi[0], hi[1], hi[2], hi[3]), d_act_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

// scg_learn.hip — the PPO learner's hot kernels for ONE network shape, compiled per shape like the config-specialised
// simulator builds:   hipcc -DSCG_L_NIN=<obs_dim> -DSCG_L_H=<hidden> -DSCG_L_NU=<act_dim> -DSCG_L_ACT=<0 tanh|1 relu|2 leaky>
//                     -> libscg_learn_<nin>_<h>_<nu>_<act>.so   (C ABI: include/scg_learn.h)
//
// Replaces, for the actor / critic pair of controllers/ppo/ppo_utils.py (MLPActor / MLPCritic: obs -> H -> H -> out):
//   scg_mlp_forward      MLP.forward on a batch                          (neural_networks.py:45-54)
//   scg_ppo_grad         one minibatch of PPOAgent.update up to the gradients: compute_policy_loss + compute_value_loss
//                        (ppo_utils.py:82-111) forward, backward through both networks, approx-KL — ONE launch
//   scg_adam_gated       the two Adam steps with the approx-KL gate on the actor (ppo_utils.py:126-138)
// All arithmetic is float32; the matrix products run on v_mfma_f32_32x32x2_f32 (exact f32, scg_mlp.h).
//
// scg_ppo_grad, per workgroup (4 waves, one per SIMD; blockIdx.y selects actor / critic):
//   * the network's parameters are packed into LDS once (scg_mlp.h layouts);
//   * each wave walks over 32-sample column tiles of the minibatch: forward (activations stay in registers), loss
//     derivatives, backward data gradients through the same LDS image, weight gradients:
//       - dW2 (H x H, the bulk): MFMA over sample pairs, accumulated over ALL the wave's tiles in 16 H^2 / 1024
//         accumulator registers (256 for H = 128: the AGPR half of the unified file at one wave per SIMD);
//       - dW1, dW3, biases, log-std: small — vector unit on transposed tiles, accumulated in LDS with ds_add_f32;
//   * at the end the four waves' dW2 accumulators are summed through LDS and the workgroup writes ONE partial gradient
//     vector; ppo_reduce_kernel sums the partials of all workgroups into the flat gradient buffer (+ the approx-KL slot
//     that the data-parallel all-reduce carries), deterministic: no global atomics anywhere.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/scg_learn.h"
#include "scg_adam.h"
#include "scg_mlp.h"
#include "scg_once.h"

#ifndef SCG_L_NIN
#error "compile with -DSCG_L_NIN= -DSCG_L_H= -DSCG_L_NU= -DSCG_L_ACT="
#endif

using namespace scg;

#ifndef SCG_L_PART_AUX
#define SCG_L_PART_AUX 17       // cache policy of the partial-gradient stores: sc0 | sc1 = write-through (0 = write-back, for A/B)
#endif
constexpr int NIN = SCG_L_NIN, HID = SCG_L_H, NU = SCG_L_NU, ACT = SCG_L_ACT;
constexpr int NT = HID / 32;
constexpr int XS_WORDS = (NIN + 2) * 32;                // a wave's sample cache: [input | ones | zeros][32 samples]
constexpr int WAVES = 4;                                // waves per workgroup of the gradient kernel

static thread_local std::string g_err;
static int fail(int code, const std::string& m) { g_err = m; return code; }
extern "C" const char* scg_learn_last_error(void) { return g_err.c_str(); }
extern "C" void scg_learn_shape(int32_t* nin, int32_t* hidden, int32_t* nu, int32_t* act) {
    *nin = NIN; *hidden = HID; *nu = NU; *act = ACT;
}
#ifndef SCG_SRC_HASH
#define SCG_SRC_HASH 0ULL
#endif
#define SCG_STR2(x) #x
#define SCG_STR(x) SCG_STR2(x)
extern "C" const char* scg_learn_source_hash_tag(void) { return "SCG_SRC_HASH:" SCG_STR(SCG_SRC_HASH); }

#define HIP_TRY(e) do { hipError_t _e = (e); if (_e != hipSuccess) return fail(-2, std::string(#e) + ": " + hipGetErrorString(_e)); } while (0)

__host__ __device__ static inline MlpWeights weights_of(const float* p, const scg_mlp_layout& L) {
    return MlpWeights{p + L.W1, p + L.b1, p + L.W2, p + L.b2, p + L.W3, p + L.b3};
}

// ------------------------------------------------------------------ input operands of a column tile
// x[q] = obs[sample][row(q, h)] for q < L1Q (rows >= NIN are zero).
template <int L1Q>
__device__ __forceinline__ void load_x(const float* __restrict__ obs, int sample, int h, float* x) {
#pragma unroll
    for (int g = 0; g < L1Q / 4; ++g) {
        const int r0 = 8 * g + 4 * h;
        if constexpr (NIN % 4 == 0) {
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (r0 < NIN) v = *reinterpret_cast<const f32x4*>(obs + (size_t)sample * NIN + r0);
            x[4 * g + 0] = v.x; x[4 * g + 1] = v.y; x[4 * g + 2] = v.z; x[4 * g + 3] = v.w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[4 * g + r] = (r0 + r < NIN) ? obs[(size_t)sample * NIN + r0 + r] : 0.0f;
        }
    }
}

// ------------------------------------------------------------------ batched forward (inference / tests)
// FWD_WAVES = 8: TWO waves per SIMD behind one weight image (the kernel keeps h1 / h2 = 2 NT accumulator tiles per wave: <= 256 registers).
// A tile is 32 NT (NT + L1Q / 16) dependent MFMAs followed, layer by layer, by the activation of 16 NT values per lane on the vector
// unit (tanh = v_exp + v_rcp at quarter rate: ~40 % of the tile's time with one wave per SIMD, during which the matrix pipe idles);
// a second wave's products run under it.
#ifndef SCG_L_FWD_WAVES
#define SCG_L_FWD_WAVES 8
#endif
constexpr int FWD_WAVES = SCG_L_FWD_WAVES;                  // (12 = three per SIMD, measured: see DESIGN 4.6)
constexpr int FWD_FILL = FWD_WAVES > 8 ? 512 : 64 * FWD_WAVES;          // threads that fill the image (must divide H * H)
template <int NOUT>
__global__ __launch_bounds__(64 * FWD_WAVES) void mlp_forward_kernel(const float* __restrict__ params, const scg_mlp_layout lay,
                                                          const float* __restrict__ xin, int M, float* __restrict__ out,
                                                          const uint8_t* __restrict__ row_mask) {
    using L = MlpLds<NIN, HID, NOUT>;
    extern __shared__ __align__(16) float lds[];
    const MlpWeights w = weights_of(params, lay);
    if (FWD_FILL == 64 * FWD_WAVES || (int)threadIdx.x < FWD_FILL) mlp_fill_lds<NIN, HID, NOUT, 20, FWD_FILL>(lds, w, threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int n_tiles = (M + 31) / 32;
    for (int tile = blockIdx.x * (blockDim.x >> 6) + wave; tile < n_tiles; tile += gridDim.x * (blockDim.x >> 6)) {
        int s = tile * 32 + c;
        const bool live = s < M;
        s = live ? s : M - 1;
        if (row_mask) {                 // sparse evaluation: a 32-row tile without a flagged row is skipped (its outputs are 0)
            const bool want = live && row_mask[s] != 0;
            if (__ballot(want) == 0ull) {
                if (live && h == 0) {
#pragma unroll
                    for (int k = 0; k < NOUT; ++k) out[(size_t)s * NOUT + k] = 0.0f;
                }
                continue;
            }
        }
        float x[L::L1Q];
        load_x<L::L1Q>(xin, s, h, x);
        f32x16 h1[NT], h2[NT];
        float o[NOUT];
        mlp_forward_tile<NIN, HID, NOUT, ACT>(lds, x, h1, h2, o, lane);
        if (live && h == 0) {
            const bool keep = !row_mask || row_mask[s] != 0;            // (a masked pass returns 0 on every row that is not flagged)
#pragma unroll
            for (int k = 0; k < NOUT; ++k) out[(size_t)s * NOUT + k] = keep ? o[k] : 0.0f;
        }
    }
}

// ------------------------------------------------------------------ PPO minibatch gradients
// Small-gradient block in LDS (float words) for a network with NOUT outputs.
template <int NOUT>
struct GradLds {
    static constexpr int DW1 = 0;                       // [NIN][H]
    static constexpr int DB1 = DW1 + NIN * HID;
    static constexpr int DB2 = DB1 + HID;
    static constexpr int DW3 = DB2 + HID;               // [NOUT][H]
    static constexpr int DB3 = DW3 + NOUT * HID;
    static constexpr int DLS = DB3 + 4;                 // d log-std (actor)
    static constexpr int STAT = DLS + 4;                // loss sum, approx-KL sum (x 1 / B)
    static constexpr int END = STAT + 4;
};
// Per-workgroup partial vector, identical order for both networks (NOUT <= 4 padded to its own NOUT):
//   [dW1 NIN*H ([in][out])] [db1 H] [db2 H] [dW3 NOUT*H] [db3 4] [dlogstd 4] [stats 4] [dW2 staging H*H]
template <int NOUT> constexpr int partial_small() { return GradLds<NOUT>::END; }
template <int NOUT> constexpr int partial_words() { return GradLds<NOUT>::END + HID * HID; }
constexpr int PARTIAL_STRIDE = partial_words<(NU > 1 ? NU : 1)>();      // actor's is the longer one

constexpr int W1R = GradLds<(NU > 1 ? NU : 1)>::END;                    // words of one wave's private small-gradient vector
constexpr size_t grad_lds_base_words() {
    size_t a = MlpLds<NIN, HID, NU>::END + GradLds<NU>::END;
    size_t c = MlpLds<NIN, HID, 1>::END + GradLds<1>::END;
    return (a > c ? a : c) + WAVES * XS_WORDS + WAVES * 4 * 32 + WAVES * TR_WORDS;
}
// Per-wave private copies of the whole small-gradient vector (dW1, db1, db2, dW3, db3, dlogstd, statistics): every
// accumulation is then a plain read-add-write by the owning wave, the copies are summed in wave order at the end, and the
// kernel's result is bitwise reproducible — used when they fit next to the weight image, LDS atomics otherwise.
constexpr bool private_dw1() { return (grad_lds_base_words() + (WAVES - 1) * W1R) * sizeof(float) <= 160 * 1024; }
// The one-tile form's operand exchange (end of grad_net): every wave publishes 2 NT tiles of 1024 words; they go over the dead weight image
// (whole tiles) and, behind the small-gradient block, over the dead per-wave scratch — which is made large enough for the rest here.
constexpr int PUB_TILE = 1024;
constexpr size_t scratch_words() { return WAVES * XS_WORDS + WAVES * 4 * 32 + WAVES * TR_WORDS + (private_dw1() ? (WAVES - 1) * W1R : 0); }
template <int NOUT>
constexpr size_t region_b_words() {
    constexpr size_t in_image = MlpLds<NIN, HID, NOUT>::END / PUB_TILE, need = 2 * WAVES * (HID / 32);
    constexpr size_t rest = need > in_image ? (need - in_image) * PUB_TILE : 0;
    return rest > scratch_words() ? rest : scratch_words();
}
static_assert(NIN < 32, "the dW1 product appends a column of ones: NIN + 1 <= 32");

struct GradArgs {
    const float* params; scg_mlp_layout actor, critic; int logstd_off;
    const float* obs; const float* act; const float* logp_old; const float* adv; const float* ret; const float* v_old;
    const int32_t* idx; int batch;
    float clip_param; int use_clipped_value;
    float* partials;                                    // [gridDim.x][2][PARTIAL_STRIDE]
};

// -DSCG_L_TIMING: the first wave of workgroup (0, actor) stamps the shader clock at its phase boundaries into the 32 words
// behind the partial vectors (slots 0..5: kernel phases; 8..: phase boundaries inside the LAST tile of that wave, i.e. one warm
// tile; tools/learn_cost.py --timeline prints them); the workspace is 256 bytes longer in that build.
#ifdef SCG_L_TIMING
#define SCG_L_SLOTS reinterpret_cast<unsigned long long*>(A.partials + (size_t)gridDim.x * 2 * PARTIAL_STRIDE)
#define SCG_L_STAMP(k) do { if (ACTOR && blockIdx.x == 0 && threadIdx.x == 0) SCG_L_SLOTS[k] = __builtin_readcyclecounter(); } while (0)
#define SCG_L_TSTAMP(k) do { tst[k] = __builtin_readcyclecounter(); } while (0)     // kept in registers, stored after the tile
#else
#define SCG_L_STAMP(k) do {} while (0)
#define SCG_L_TSTAMP(k) do {} while (0)
#endif

// += into the wave's small-gradient vector: its own copy (plain) or the shared one (LDS atomic)
__device__ __forceinline__ void gl_add(float* p, float v) {
    if constexpr (private_dw1()) *p += v;
    else atomicAdd(p, v);
}

// ONE = at most one tile per wave (tiles <= 4 x workgroups: what the shipped minibatch sizes give — 16 256 rows = 508 tiles on 127
// workgroups per network).  The dW2 "accumulators" then accumulate nothing: each of the NT^2 tile products is a single 16-MFMA chain, and
// holding all of them in 256 AccVGPRs until the cross-wave sum (a) left everything else of the tile 256 registers — 35 spilled words per
// lane, each reload a scratch round trip behind an s_waitcnt vmcnt(0) — and (b) made that sum a separate 5.6 us pass over 64 KB of LDS.
// With ONE the products are formed behind the barrier that frees the weight image, by OPERAND exchange: every wave publishes its tile's
// transposed h1 / dz2 tiles in the LDS, and wave w forms tile row tau = w of dW2 over all the workgroup's samples in its MFMA accumulators
// (publishers in a fixed order) and stores it straight to the partial vector.  (First cut of the round: running sums through a staging
// area in four rounds — the accumulating form's own sum with the products moved into it, bit-identical to it: 11.2 us for 6.8 of MFMA.)
template <int NOUT, bool ACTOR, bool ONE>
__device__ __forceinline__ void grad_net(const GradArgs& A, float* lds) {
    using L = MlpLds<NIN, HID, NOUT>;
    using G = GradLds<NOUT>;
    constexpr int L1Q = L::L1Q;
    float* const gl = lds + L::END;                                     // small gradients
    float* const xs_all = gl + G::END;                                  // [WAVES][NIN + 2][32]
    float* const dout_all = xs_all + WAVES * XS_WORDS;                    // [WAVES][NOUT][32]
    float* const scr_all = dout_all + WAVES * 4 * 32;                   // [WAVES][TR_WORDS]
    float* const w1_all = scr_all + WAVES * TR_WORDS;                   // [WAVES - 1][W1R] (private_dw1() only)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const MlpWeights w = weights_of(A.params, ACTOR ? A.actor : A.critic);
    SCG_L_STAMP(0);
    // The first tile's row index is requested BEFORE the weight image is filled: index -> row gather is two dependent memory round
    // trips, and the first of them otherwise starts behind the fill's barrier (one tile per wave at the shipped minibatch size: every
    // call paid it on its critical path).  (The rows themselves requested up here as well cost 40 more spilled registers: not kept.)
    const int n_tiles = A.batch / 32;
    const int tile0 = blockIdx.x * WAVES + wave;
    const int s_pre = A.idx[tile0 < n_tiles ? tile0 * 32 + c : c];      // (unconditional: a `cond ? load : 0` keeps its use in the load's branch)
    // ONE: the first (only) tile's rows and per-sample scalars are requested INSIDE the fill, once its first batch of requests is out —
    // they arrive under the rest of the fill instead of costing a round trip behind its barrier (the accumulating form has no registers
    // for them: 40 spilled words when it was tried there)
    float x_pre[L1Q], act_pre[NOUT], a_pre = 0.0f, b_pre = 0.0f;
    if constexpr (ONE) {
        mlp_fill_lds<NIN, HID, NOUT>(lds, w, tid, [&]() {
            load_x<L1Q>(A.obs, s_pre, h, x_pre);
            if constexpr (ACTOR) {
#pragma unroll
                for (int a = 0; a < NOUT; ++a) act_pre[a] = A.act[(size_t)s_pre * NOUT + a];
                a_pre = A.logp_old[s_pre]; b_pre = A.adv[s_pre];
            } else {
                a_pre = A.ret[s_pre];
                if (A.use_clipped_value) b_pre = A.v_old[s_pre];
            }
        });
    } else {
        mlp_fill_lds<NIN, HID, NOUT>(lds, w, tid);
    }
    for (int k = tid; k < G::END; k += blockDim.x) gl[k] = 0.0f;
    if constexpr (private_dw1()) {
        for (int k = tid; k < (WAVES - 1) * W1R; k += blockDim.x) w1_all[k] = 0.0f;
    }
    // rows NIN (ones: the bias column of the dW1 product) and NIN + 1 (zeros: what the lanes beyond it read) of every sample cache
    for (int k = tid; k < WAVES * 64; k += blockDim.x) xs_all[(k >> 6) * XS_WORDS + NIN * 32 + (k & 63)] = (k & 63) < 32 ? 1.0f : 0.0f;
    __syncthreads();
    SCG_L_STAMP(1);
    float* const xs = xs_all + wave * XS_WORDS;
    float* const dout_l = dout_all + wave * 4 * 32;
    float* const scr = scr_all + wave * TR_WORDS;
    // this wave's small-gradient vector; wave 0 (and every wave, when the private copies do not fit the LDS) uses the shared one
    float* const glw = (private_dw1() && wave > 0) ? w1_all + (wave - 1) * W1R : gl;
    float* const w1 = glw + G::DW1;                                     // dW1 | db1: [input c <= NIN][feature]
    float logstd[NOUT], inv_std[NOUT];
    if constexpr (ACTOR) {
#pragma unroll
        for (int a = 0; a < NOUT; ++a) { logstd[a] = A.params[A.logstd_off + a]; inv_std[a] = __expf(-logstd[a]); }
    }
    const float inv_b = 1.0f / (float)A.batch;
    float st_loss = 0.0f, st_kl = 0.0f, dls[NOUT];
#pragma unroll
    for (int a = 0; a < NOUT; ++a) dls[a] = 0.0f;
    f32x16 dW2[ONE ? 1 : NT][ONE ? 1 : NT];                             // [tau (in tile)][rho (out tile)]
    if constexpr (!ONE) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < NT; ++r)
#pragma unroll
                for (int q = 0; q < 16; ++q) dW2[t][r][q] = 0.0f;
    }
    f32x16 h1[NT], h2[NT];                      // (ONE: the tile's transposed h1 / dz2 tiles outlive the loop — operands of the products below)
    if constexpr (ONE) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) { h1[t][q] = 0.0f; h2[t][q] = 0.0f; }
    }

    for (int tile = tile0; tile < n_tiles; tile += gridDim.x * WAVES) {
#ifdef SCG_L_TIMING
        unsigned long long tst[10];
#endif
        SCG_L_TSTAMP(0);
        float x[L1Q];
        // the per-sample scalars of the loss are requested here, a forward pass ahead of their use (asked for where they are
        // used, their round trip to memory was 2.3 us of a 37 us tile)
        float s_act[NOUT], s_a = 0.0f, s_b = 0.0f;
        if constexpr (ONE) {
#pragma unroll
            for (int q = 0; q < L1Q; ++q) x[q] = x_pre[q];
#pragma unroll
            for (int a = 0; a < NOUT; ++a) s_act[a] = act_pre[a];
            s_a = a_pre; s_b = b_pre;
        } else {
            const int s = tile == tile0 ? s_pre : A.idx[tile * 32 + c];
            load_x<L1Q>(A.obs, s, h, x);
            if constexpr (ACTOR) {
#pragma unroll
                for (int a = 0; a < NOUT; ++a) s_act[a] = A.act[(size_t)s * NOUT + a];
                s_a = A.logp_old[s]; s_b = A.adv[s];
            } else {
                s_a = A.ret[s];
                if (A.use_clipped_value) s_b = A.v_old[s];
            }
        }
        // sample cache for dW1: xs[input column][sample c]
#pragma unroll
        for (int q = 0; q < L1Q; ++q) {
            const int f = d_row(q, 0);                                  // this lane holds column f + 4 h
            if (f + 4 < NIN) xs[(f + 4 * h) * 32 + c] = x[q];
            else if (f < NIN) { if (h == 0) xs[f * 32 + c] = x[q]; }
        }
        SCG_L_TSTAMP(1);
        float out[NOUT], dout[NOUT];
#ifdef SCG_L_TIMING
        mlp_forward_tile<NIN, HID, NOUT, ACT, 20, ACT, ONE>(lds, x, h1, h2, out, lane, tst + 8);
#else
        mlp_forward_tile<NIN, HID, NOUT, ACT, 20, ACT, ONE>(lds, x, h1, h2, out, lane);
#endif
        SCG_L_TSTAMP(2);
        // ---- loss derivatives w.r.t. the network outputs (both lane halves compute the same numbers)
        if constexpr (ACTOR) {
            // compute_policy_loss (ppo_utils.py:82-96): Normal(mean, exp(logstd)).log_prob(act).sum(-1), clipped surrogate
            float logp = 0.0f, z[NOUT];
#pragma unroll
            for (int a = 0; a < NOUT; ++a) {
                z[a] = (s_act[a] - out[a]) * inv_std[a];
                logp += -0.5f * z[a] * z[a] - logstd[a] - 0.91893853320467274f;
            }
            const float lp_old = s_a, adv = s_b;
            const float ratio = __expf(logp - lp_old);
            const float lo = 1.0f - A.clip_param, hi = 1.0f + A.clip_param;
            const float rc = fminf(fmaxf(ratio, lo), hi);
            const float s1 = ratio * adv, s2 = rc * adv;
            // d min(s1, s2) / d ratio: inside the clip range both terms move (torch splits the tie, the sum is adv);
            // outside it only the unclipped term has a gradient, and only if it is the smaller one
            const bool inside = ratio >= lo && ratio <= hi;
            const float dsur = (inside || s1 < s2) ? adv : 0.0f;
            const float wl = -dsur * ratio * inv_b;                     // d policy_loss / d logp
#pragma unroll
            for (int a = 0; a < NOUT; ++a) {
                dout[a] = wl * z[a] * inv_std[a];                       // d logp / d mean = z / sigma
                if (h == 0) dls[a] += wl * (z[a] * z[a] - 1.0f);        // d logp / d logstd
            }
            if (h == 0) { st_loss += -fminf(s1, s2) * inv_b; st_kl += (lp_old - logp) * inv_b; }
        } else {
            // compute_value_loss (ppo_utils.py:98-111)
            const float v = out[0], ret = s_a;
            float dv = v - ret, l = dv * dv;
            if (A.use_clipped_value) {
                const float vo = s_b;
                const float d = v - vo;
                const float vc = vo + fminf(fmaxf(d, -A.clip_param), A.clip_param);
                const float e2 = vc - ret, l2 = e2 * e2;
                const float pass = (d >= -A.clip_param && d <= A.clip_param) ? 1.0f : 0.0f;
                if (l2 > l) { dv = e2 * pass; l = l2; }
                else if (l2 == l) dv = 0.5f * (dv + e2 * pass);
            }
            dout[0] = dv * inv_b;
            if (h == 0) st_loss += 0.5f * l * inv_b;
        }
        if (h == 0) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) dout_l[o * 32 + c] = dout[o];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // db3: lane o sums row o of the LDS copy (eight 16-byte reads, a fixed order) — as a butterfly over the lanes it was five dependent
        // ds_bpermute round trips per output, serialised further by the branches of the stores between the outputs
        if (lane < NOUT) {
            float v = 0.0f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dout_l + lane * 32 + 4 * g);
                v += d4.x; v += d4.y; v += d4.z; v += d4.w;
            }
            gl_add(glw + G::DB3 + lane, v);
        }
        SCG_L_TSTAMP(3);
#ifndef DBG_NO_DW3
        // ---- output layer backward: dW3 from the transposed h2 tiles, then dz2 = (W3^T dout) * act'(h2) in place
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            float t[16];
            tile_transpose(scr, h2[tau], t, lane);                      // t[q] = h2[feature 32 tau + c][sample row(q, h)]
            float a3[NOUT], x3[NOUT];
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
#pragma unroll
            for (int o = 0; o < NOUT; ++o) x3[o] = __shfl_xor(a3[o], 32, 64);
            if (h == 0) {
#pragma unroll
                for (int o = 0; o < NOUT; ++o) gl_add(glw + G::DW3 + o * HID + 32 * tau + c, a3[o] + x3[o]);
            }
        }
#endif
        SCG_L_TSTAMP(4);
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float dh[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(lds + L::W3 + o * HID + 32 * tau + 8 * g + 4 * h);
                    dh[0] = __builtin_fmaf(wv.x, dout[o], dh[0]); dh[1] = __builtin_fmaf(wv.y, dout[o], dh[1]);
                    dh[2] = __builtin_fmaf(wv.z, dout[o], dh[2]); dh[3] = __builtin_fmaf(wv.w, dout[o], dh[3]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) h2[tau][4 * g + r] = dh[r] * mlp_dact<ACT>(h2[tau][4 * g + r]);
            }
        }
        // (h2 now holds dz2, still in the accumulator layout: lane = sample)
        // h1 is needed from here on only with the FEATURE on the lane (the activation derivative of the transposed data
        // gradient below, the A operand of dW2): one in-place transpose per tile
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) tile_transpose_inplace(scr, h1[tau], lane);
        SCG_L_TSTAMP(5);
        __builtin_amdgcn_sched_barrier(0);
        // ---- data gradient through layer 2, one input tile at a time, computed TRANSPOSED: with dz2 as the A operand and the
        //      weights as B the product comes out as
        //        dh1^T[sample row(q', h)][feature 32 tau' + i'] = sum over (rho', q) of dz2[rho'][q] W2[32 rho' + row(q, h)][32 tau' + i'],
        //      i.e. with the feature on the lane — already the A operand of the dW1 product (no transpose of the result; the
        //      first form computed dh1 with the sample on the lane and sent every tile through the scratch).
        //      dz1 = dh1 * act'(h1); each dz1 tile is consumed at once (dW1 and db1 with the cached inputs) and never stored
        {
            const int ip = lane & 31;
            const int qi = 4 * (ip >> 3) + (ip & 3), hi2 = (ip >> 2) & 1;
            const float* const xrow = xs + (c < NIN + 1 ? c : NIN + 1) * 32 + 4 * h;   // [x | 1 | 0][column c][sample row(., h)]
            // (the weight operand of block (tp, rp) — 16 gathered words — is read one block AHEAD, in front of the previous block's products:
            //  read where it is used, every 16-MFMA chain began with an exposed LDS round trip.  ONE only: the accumulating form has no
            //  registers for the second buffer.)
            float a_nx[16];
            auto gather = [&](int tp, int rp, float* a) {
                const float* base = lds + L::W2F + (rp * NT + tp) * L::TILE2 + 32 * hi2 * L::S + qi;
#pragma unroll
                for (int q = 0; q < 16; ++q) a[q] = base[d_row(q, h) * L::S];
            };
            if constexpr (ONE) gather(0, 0, a_nx);
#pragma unroll
            for (int tp = 0; tp < NT; ++tp) {
                f32x16 acc;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
#pragma unroll
                for (int rp = 0; rp < NT; ++rp) {
                    float a[16];
                    if constexpr (ONE) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) a[q] = a_nx[q];
                        if (rp + 1 < NT) gather(tp, rp + 1, a_nx);
                        else if (tp + 1 < NT) gather(tp + 1, 0, a_nx);
                    } else {
                        gather(tp, rp, a);
                    }
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc = mfma32(h2[rp][q], a[q], acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] *= mlp_dact<ACT>(h1[tp][q]);
#ifndef DBG_NO_DW1
                // dW1 and db1 on the matrix cores as well: [dz1 tile (32 features x 32 samples)] x [x | 1] (32 samples x
                // (NIN + 1) columns).  A = the dz1 tile as it stands, B = the cached inputs with a column of ones appended,
                // D[feature][c]: c < NIN -> dW1[feature][c], c == NIN -> db1[feature].  The wave's running sums are the C input of
                // the first product and the result is stored back: no read-add-write.
                // (The vector-unit form — 16 sample pairs x NIN FMAs per lane + 13 LDS atomics per input tile — was the most
                //  expensive section of the kernel: 18 of 49 us per tile for the smallest of the three weight matrices.)
                {
                    float* const dst = w1 + (c <= NIN ? c : NIN) * HID + 32 * tp + 4 * h;
                    f32x16 g1;
                    if constexpr (private_dw1()) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(dst + 8 * g);
                            g1[4 * g] = v.x; g1[4 * g + 1] = v.y; g1[4 * g + 2] = v.z; g1[4 * g + 3] = v.w;
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 16; ++q) g1[q] = 0.0f;
                    }
                    float xb[16];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(xrow + 8 * g);
                        xb[4 * g] = v.x; xb[4 * g + 1] = v.y; xb[4 * g + 2] = v.z; xb[4 * g + 3] = v.w;
                    }
#pragma unroll
                    for (int q = 0; q < 16; ++q) g1 = mfma32(acc[q], xb[q], g1);
                    if (c <= NIN) {
                        if constexpr (private_dw1()) {                  // this wave's own words
#pragma unroll
                            for (int g = 0; g < 4; ++g)
                                *reinterpret_cast<f32x4*>(dst + 8 * g) = (f32x4){g1[4 * g], g1[4 * g + 1], g1[4 * g + 2], g1[4 * g + 3]};
                        } else {
#pragma unroll
                            for (int q = 0; q < 16; ++q) atomicAdd(dst + d_row(q, 0), g1[q]);
                        }
                    }
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        SCG_L_TSTAMP(6);
#ifndef DBG_NO_DW2
        // ---- dW2 += h1 dz2^T over this tile's 32 samples (MFMA over sample pairs), db2: dz2 goes through the scratch as h1 did
        //      (8 in-place transposes for this product; transposing h1[tau] inside the rho loop made it 20)
        float sb[NT], xb2[NT];
#pragma unroll
        for (int rho = 0; rho < NT; ++rho) {
            tile_transpose_inplace(scr, h2[rho], lane);                 // dz2[out 32 rho + c][sample row(q, h)]
            float v = 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) v += h2[rho][q];
            sb[rho] = v;
        }
#pragma unroll
        for (int rho = 0; rho < NT; ++rho) xb2[rho] = __shfl_xor(sb[rho], 32, 64);
        if (h == 0) {
#pragma unroll
            for (int rho = 0; rho < NT; ++rho) gl_add(glw + G::DB2 + 32 * rho + c, sb[rho] + xb2[rho]);
        }
        if constexpr (!ONE) {
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) {
#pragma unroll
                for (int rho = 0; rho < NT; ++rho) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) dW2[tau][rho] = mfma32(h1[tau][q], h2[rho][q], dW2[tau][rho]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#endif
        SCG_L_TSTAMP(7);
#ifdef SCG_L_TIMING
        if (ACTOR && blockIdx.x == 0 && threadIdx.x == 0 && tile + (int)gridDim.x * WAVES >= n_tiles) {
#pragma unroll
            for (int k = 0; k < 8; ++k) SCG_L_SLOTS[8 + k] = tst[k];
            SCG_L_SLOTS[24] = tst[8]; SCG_L_SLOTS[25] = tst[9];
        }
#endif
    }
    SCG_L_STAMP(2);
    // ---- workgroup reduction and the partial vector
    // per-lane running sums (lanes of half 0) -> the wave's totals: rows of the wave's transpose scratch, lane k sums row k (fixed order) —
    // butterflies over the lanes were up to 4 x 5 dependent ds_bpermute round trips at the end of every call
    {
        static_assert((NOUT + 2) * 32 <= TR_WORDS, "rows in the transpose scratch");
        if (lane < 32) {
#pragma unroll
            for (int a = 0; a < NOUT; ++a) scr[a * 32 + lane] = ACTOR ? dls[a] : 0.0f;
            scr[NOUT * 32 + lane] = st_loss; scr[(NOUT + 1) * 32 + lane] = st_kl;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < NOUT + 2) {
            float v = 0.0f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(scr + lane * 32 + 4 * g);
                v += d4.x; v += d4.y; v += d4.z; v += d4.w;
            }
            if (lane < NOUT) { if (ACTOR) gl_add(glw + G::DLS + lane, v); }
            else gl_add(glw + G::STAT + (lane - NOUT), v);
        }
        // entropy_loss = -sum_a (0.5 + 0.5 log(2 pi) + logstd_a) of the parameters THIS launch read: slot STAT + 2 of one partial vector
        // (every other workgroup contributes 0), so that the reduction needs no look at the parameters — which the fused reduction + Adam
        // kernel updates while it sums
        if constexpr (ACTOR) {
            if (lane == 0 && wave == 0 && blockIdx.x == 0) {
                float ent = 0.0f;
#pragma unroll
                for (int a = 0; a < NOUT; ++a) ent -= 1.4189385332046727f + logstd[a];
                gl_add(glw + G::STAT + 2, ent);
            }
        }
    }
    __syncthreads();                                                    // every wave is done with the weight image
    if constexpr (private_dw1()) {
        for (int k = tid; k < G::END; k += blockDim.x) {
            float v = gl[k];
#pragma unroll
            for (int wv = 0; wv < WAVES - 1; ++wv) v += w1_all[wv * W1R + k];
            gl[k] = v;
        }
    }
    SCG_L_STAMP(3);
    // The waves' dW2 accumulators are summed through the (now free) weight image in a FIXED order with every wave busy in every
    // round: in round r wave w owns the tile rows tau = (w + r) mod WAVES — round 0 writes, the later rounds read-add-write,
    // 16-byte LDS accesses — so row tau is summed in the order wave tau, tau - 1, ...  (One wave at a time, wave 0 to 3, made four
    // passes over the 64 KB instead of one: 9.3 us; 256 ds_add_f32 per lane from four waves onto the same words took 86 us and
    // summed in arrival order.)  Word order: [tile][g][lane][4] (accumulator word q = 4 g + r): every 16-byte access of a wave, in the LDS and
    // in the partial vector, covers 1 KB of consecutive addresses.
    float* const stg = lds + L::W2F;                                    // H * H words
    static_assert(NT <= WAVES, "round 0 must write every tile row");
    static_assert(G::END % 4 == 0 && PARTIAL_STRIDE % 4 == 0 && (HID * HID) % 4 == 0, "16-byte partial stores");
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc((void*)A.partials, 0, 0xffffffff, 0x00020000);
    const uint32_t pbase = (uint32_t)(((size_t)blockIdx.x * 2 + (ACTOR ? 0 : 1)) * PARTIAL_STRIDE * sizeof(float));
    typedef unsigned int u32x4 __attribute__((vector_size(16)));
    if constexpr (ONE) {
        // ONE: operand exchange instead of partial-sum exchange.  Every wave publishes its tile's transposed h1 / dz2 tiles (2 NT x 4 KB) in
        // the LDS — the weight image and the per-wave scratch are dead by now — and wave w then forms tile row tau = w of dW2 over ALL the
        // workgroup's samples: for each publisher t, A = h1^T_t[tau], B = dz2^T_t[rho], accumulated in the MFMA accumulators (publishers in
        // the order 0..3: a fixed order) and stored straight to the partial vector.  One barrier where the running-sum form (rounds of
        // read-add-write through a staging area) had three, and no LDS round trip per product on the accumulators' path: 11.2 -> ~8 us.
        // (Not bit-identical to the accumulating form any more: the sum over a wave's tile partners happens inside the accumulator.)
        constexpr int TW = 1024;                                        // words of a published tile: [g][lane][4]
        constexpr int A_TILES = (int)(L::END / TW);                     // tiles that fit the dead weight image ...
        float* const reg_b = xs_all;                                    // ... the rest goes over the dead per-wave scratch (xs, dout, scr, private copies)
        static_assert(TW == PUB_TILE && (2 * WAVES * NT <= A_TILES || (size_t)(2 * WAVES * NT - A_TILES) * TW <= region_b_words<NOUT>()),
                      "published tiles do not fit the LDS");
        auto slot = [&](int wv, int kind, int t) -> float* {
            const int sidx = (wv * 2 + kind) * NT + t;
            return sidx < A_TILES ? lds + sidx * TW : reg_b + (sidx - A_TILES) * TW;
        };
        __syncthreads();                                                // the private small-gradient copies have been summed: the scratch is dead
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float* const pa = slot(wave, 0, t) + lane * 4;
            float* const pb = slot(wave, 1, t) + lane * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<f32x4*>(pa + 256 * g) = (f32x4){h1[t][4 * g], h1[t][4 * g + 1], h1[t][4 * g + 2], h1[t][4 * g + 3]};
                *reinterpret_cast<f32x4*>(pb + 256 * g) = (f32x4){h2[t][4 * g], h2[t][4 * g + 1], h2[t][4 * g + 2], h2[t][4 * g + 3]};
            }
        }
        __syncthreads();
        if (wave < NT) {
            const int tau = wave;
            f32x16 d2[NT];
#pragma unroll
            for (int rho = 0; rho < NT; ++rho)
#pragma unroll
                for (int q = 0; q < 16; ++q) d2[rho][q] = 0.0f;
            // (the operands of block (publisher, rho) are read one block AHEAD of their 16 products: 4 + 4 16-byte LDS reads under the
            //  previous block's chain instead of in front of their own)
            auto rd = [&](const float* base, float* o) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(base + 256 * g);
                    o[4 * g] = v.x; o[4 * g + 1] = v.y; o[4 * g + 2] = v.z; o[4 * g + 3] = v.w;
                }
            };
            auto a_of = [&](int pw) -> const float* {
                const float* pa = slot(pw, 0, 0) + lane * 4;
#pragma unroll
                for (int t = 1; t < NT; ++t) {
                    if (tau == t) pa = slot(pw, 0, t) + lane * 4;       // wave-uniform selection
                }
                return pa;
            };
            float a_nx[16], b_nx[16];
            rd(a_of(0), a_nx);
            rd(slot(0, 1, 0) + lane * 4, b_nx);
#pragma unroll
            for (int pw = 0; pw < WAVES; ++pw) {                        // publishers in a fixed order
                float a[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) a[q] = a_nx[q];
                if (pw + 1 < WAVES) rd(a_of(pw + 1), a_nx);
#pragma unroll
                for (int rho = 0; rho < NT; ++rho) {
                    float bq[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) bq[q] = b_nx[q];
                    if (rho + 1 < NT) rd(slot(pw, 1, rho + 1) + lane * 4, b_nx);
                    else if (pw + 1 < WAVES) rd(slot(pw + 1, 1, 0) + lane * 4, b_nx);
#pragma unroll
                    for (int q = 0; q < 16; ++q) d2[rho] = mfma32(a[q], bq[q], d2[rho]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int rho = 0; rho < NT; ++rho) {
                const int word = ((tau * NT + rho) * 4 * 64 + lane) * 4;              // [tile][g][lane][4]
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {d2[rho][4 * g], d2[rho][4 * g + 1], d2[rho][4 * g + 2], d2[rho][4 * g + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), pr, pbase + 4u * (uint32_t)(G::END + word + 256 * g), 0, SCG_L_PART_AUX);
                }
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < WAVES; ++r) {
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) {
                if (tau != ((wave + r) % WAVES)) continue;
#pragma unroll
                for (int rho = 0; rho < NT; ++rho) {
                    float* const p = stg + ((tau * NT + rho) * 4 * 64 + lane) * 4;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = {dW2[tau][rho][4 * g], dW2[tau][rho][4 * g + 1], dW2[tau][rho][4 * g + 2], dW2[tau][rho][4 * g + 3]};
                        if (r > 0) v += *reinterpret_cast<const f32x4*>(p + 256 * g);
                        *reinterpret_cast<f32x4*>(p + 256 * g) = v;
                    }
                }
            }
            __syncthreads();
        }
    }
    SCG_L_STAMP(4);
    // The workgroup's partial vector (74 KB for H = 128; 254 workgroups: 19 MB per call) leaves as 16-byte stores WRITTEN THROUGH
    // (sc0 sc1, SCG_L_PART_AUX): the reduction kernel that follows reads it from other XCDs, so it has to reach the memory side
    // before this kernel may retire — written back, that is one flush behind the LAST workgroup's last store; written through,
    // the partials of workgroups that finish early (the tile counts differ by one between waves) are out already.
    for (int k = 4 * tid; k < G::END; k += 4 * blockDim.x)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(gl + k)), pr, pbase + 4u * (uint32_t)k, 0, SCG_L_PART_AUX);
    if constexpr (!ONE) {
        for (int k = 4 * tid; k < HID * HID; k += 4 * blockDim.x)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(stg + k)), pr, pbase + 4u * (uint32_t)(G::END + k), 0,
                                                   SCG_L_PART_AUX);
    }
    SCG_L_STAMP(5);
}

constexpr size_t grad_lds_words() {
    size_t a = MlpLds<NIN, HID, NU>::END + GradLds<NU>::END + region_b_words<NU>();
    size_t c = MlpLds<NIN, HID, 1>::END + GradLds<1>::END + region_b_words<1>();
    return a > c ? a : c;
}
static_assert(grad_lds_words() * sizeof(float) <= 160 * 1024, "the gradient kernel's LDS does not fit");

template <bool ONE>
__global__ __launch_bounds__(64 * WAVES, 1) void ppo_grad_kernel(const GradArgs A) {
    extern __shared__ __align__(16) float lds[];
    if (blockIdx.y == 0) grad_net<NU, true, ONE>(A, lds);
    else grad_net<1, false, ONE>(A, lds);
}
// A/B and test hook: the accumulating form at any size (initially: $SCG_LEARN_MULTI_TILE set).  The forms agree up to dW2's summation order.
static int g_force_accumulating = getenv("SCG_LEARN_MULTI_TILE") != nullptr;
extern "C" void scg_learn_force_accumulating_form(int on) { g_force_accumulating = on != 0; }
// enqueue the gradient kernel: the one-tile form when no wave has more than one tile
static int launch_grad(const GradArgs& G, int n_workgroups, hipStream_t st) {
    const size_t bytes = grad_lds_words() * sizeof(float);
    static scg::PerDeviceOnce set_g;
    int dev;
    if (set_g.pending(&dev)) {
        HIP_TRY(hipFuncSetAttribute((const void*)ppo_grad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        HIP_TRY(hipFuncSetAttribute((const void*)ppo_grad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        set_g.commit(dev);
    }
    if (G.batch / 32 <= n_workgroups * WAVES && !g_force_accumulating) ppo_grad_kernel<true><<<dim3(n_workgroups, 2), dim3(64 * WAVES), bytes, st>>>(G);
    else ppo_grad_kernel<false><<<dim3(n_workgroups, 2), dim3(64 * WAVES), bytes, st>>>(G);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Sum of the workgroup partials -> flat gradient vector (torch parameter order) + approx-KL slot + loss statistics.
struct ReduceArgs {
    const float* partials; int n_wg;
    scg_mlp_layout actor, critic; int logstd_off; int n_params;
    float entropy_coef;
    const float* params;
    float* grad;            // [n_params + 1]: gradients, then approx_kl
    float* stats;           // [4]: policy_loss, value_loss, entropy_loss, approx_kl of this minibatch
};

template <int NOUT>
__device__ __forceinline__ int dest_of(int k, const scg_mlp_layout& lay, int logstd_off, bool actor) {
    using G = GradLds<NOUT>;
    if (k < G::DB1) { const int in = k / HID, o = k % HID; return lay.W1 + o * NIN + in; }
    if (k < G::DB2) return lay.b1 + (k - G::DB1);
    if (k < G::DW3) return lay.b2 + (k - G::DB2);
    if (k < G::DB3) return lay.W3 + (k - G::DW3);
    if (k < G::DLS) return (k - G::DB3) < NOUT ? lay.b3 + (k - G::DB3) : -1;
    if (k < G::STAT) return (actor && (k - G::DLS) < NOUT) ? logstd_off + (k - G::DLS) : -1;
    if (k < G::END) return -2 - (k - G::STAT);                          // statistics slots
    const int p = k - G::END;                                           // dW2 staging order -> W2[out][in]
    const int q = 4 * ((p >> 8) & 3) + (p & 3), lane = (p >> 2) & 63, tr = p >> 10;       // word order [tile][g][lane][4], q = 4 g + r
    const int tau = tr / NT, rho = tr % NT;
    return lay.W2 + (32 * rho + (lane & 31)) * HID + 32 * tau + d_row(q, lane >> 5);
}

// 64 partial-vector words per block, the workgroups' partials split over the block's four waves (each wave keeps several
// independent loads in flight; one wave walking all 128 partials of a word was a 128-deep dependent-latency chain), then a
// fixed-order sum of the four: deterministic.
__global__ __launch_bounds__(256) void ppo_reduce_kernel(const ReduceArgs R) {
    __shared__ float part[4][64];
    const int net = blockIdx.y;
    const int kl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kl;
    const int words = net == 0 ? partial_words<NU>() : partial_words<1>();
    float s = 0.0f;
    if (k < words) {
#pragma unroll 8
        for (int g = grp; g < R.n_wg; g += 4) s += R.partials[((size_t)g * 2 + net) * PARTIAL_STRIDE + k];
    }
    part[grp][kl] = s;
    __syncthreads();
    if (grp != 0 || k >= words) return;
    s = (part[0][kl] + part[1][kl]) + (part[2][kl] + part[3][kl]);
    const int d = net == 0 ? dest_of<NU>(k, R.actor, R.logstd_off, true) : dest_of<1>(k, R.critic, 0, false);
    if (d >= 0) {
        if (net == 0 && d >= R.logstd_off && d < R.logstd_off + NU) s -= R.entropy_coef;   // d (c_ent * entropy_loss) / d logstd
        R.grad[d] = s;
    } else if (d == -2) {                                   // loss sum
        R.stats[net == 0 ? 0 : 1] = s;
    } else if (d == -3 && net == 0) {                       // approx-KL: rides in the gradient buffer's last slot
        R.grad[R.n_params] = s;
        R.stats[3] = s;
    } else if (d == -4 && net == 0) {                       // entropy_loss (written by one workgroup of the gradient kernel)
        R.stats[2] = s;
    }
}

// Single-GPU optimiser step: the reduction above AND the two gated Adam steps of adam_gated_kernel in ONE launch — every parameter's
// gradient is summed by exactly one thread, which steps that parameter at once (no trip of the gradient through memory, no second
// launch, no device-scope fence: the step counts are double-buffered, read from `steps_in` by everybody, written to `steps_out` by
// the one thread that owns the approx-KL word).  The gate needs the minibatch's approx-KL before the first actor element steps:
// every block sums that ONE word of the workgroups' partials itself, all in the same fixed order, and the owner of the word
// reports exactly that value.  Data-parallel callers keep scg_ppo_grad -> all-reduce -> scg_adam_gated.
struct StepArgs {
    ReduceArgs R;
    float* p; float* m; float* v; float lr_actor, lr_critic;
    const float* steps_in; float* steps_out; float target_kl; float* stats_acc;
};
__global__ __launch_bounds__(256) void ppo_reduce_adam_kernel(const StepArgs S) {
    __shared__ float part[4][64];
    __shared__ float klw[4];
    const ReduceArgs& R = S.R;
    const int net = blockIdx.y;
    const int kl_ = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kl_;
    const int words = net == 0 ? partial_words<NU>() : partial_words<1>();
    // Request order (round 6): the approx-KL word of this thread's workgroup, what the word's OWNER needs for its Adam step (moments,
    // parameter, step count — asked for behind the sum they were one more memory round trip at the end of every block), then ALL of
    // this thread's partial words at once (32 at 127 workgroups; `unroll 8` made that four rounds of eight).  Every sum keeps its order.
    constexpr int KW = GradLds<NU>::STAT + 1;               // the actor's approx-KL word
    float klv = 0.0f;
    if ((int)threadIdx.x < R.n_wg) klv = 0.0f + R.partials[((size_t)threadIdx.x * 2 + 0) * PARTIAL_STRIDE + KW];
    const bool owner = grp == 0 && k < words;
    const bool critic = net == 1;
    int d = -1;
    float o_p = 0.0f, o_m = 0.0f, o_v = 0.0f, o_t = 0.0f;
    if (owner) {
        d = net == 0 ? dest_of<NU>(k, R.actor, R.logstd_off, true) : dest_of<1>(k, R.critic, 0, false);
        if (d >= 0) { o_p = S.p[d]; o_m = S.m[d]; o_v = S.v[d]; o_t = S.steps_in[critic ? 1 : 0]; }
    }
    float s = 0.0f;
    if (k < words) {
        const float* const src = R.partials + ((size_t)grp * 2 + net) * PARTIAL_STRIDE + k;       // workgroups grp, grp + 4, ...
        const int mine = (R.n_wg - grp + 3) / 4;
        for (int g0 = 0; g0 < mine; g0 += 32) {
            float v[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) v[j] = g0 + j < mine ? src[(size_t)(g0 + j) * 8 * PARTIAL_STRIDE] : 0.0f;
#pragma unroll
            for (int j = 0; j < 32; ++j) { if (g0 + j < mine) s += v[j]; }
        }
    }
    {
        float v = klv;
        for (int g = threadIdx.x + 256; g < R.n_wg; g += 256) v += R.partials[((size_t)g * 2 + 0) * PARTIAL_STRIDE + KW];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if (kl_ == 0) klw[grp] = v;
    }
    part[grp][kl_] = s;
    __syncthreads();
    if (!owner) return;
    s = (part[0][kl_] + part[1][kl_]) + (part[2][kl_] + part[3][kl_]);
    const float kl = (klw[0] + klw[1]) + (klw[2] + klw[3]);
    const bool gate = S.target_kl <= 0.0f || kl <= 1.5f * S.target_kl;
    if (d >= 0) {
        if (net == 0 && d >= R.logstd_off && d < R.logstd_off + NU) s -= R.entropy_coef;   // d (c_ent * entropy_loss) / d logstd
        R.grad[d] = s;
        if (critic || gate) {
            adam_element(o_p, s, o_m, o_v, critic ? S.lr_critic : S.lr_actor, o_t + 1.0f);
            S.m[d] = o_m; S.v[d] = o_v; S.p[d] = o_p;
        }
    } else if (d == -2) {                                   // loss sum
        R.stats[net == 0 ? 0 : 1] = s;
        if (S.stats_acc) S.stats_acc[net == 0 ? 0 : 1] += s;
    } else if (d == -3 && net == 0) {                       // approx-KL (+ the step's bookkeeping: this thread is unique in the launch)
        R.grad[R.n_params] = kl;
        R.stats[3] = kl;
        S.steps_out[0] = S.steps_in[0] + (gate ? 1.0f : 0.0f);
        S.steps_out[1] = S.steps_in[1] + 1.0f;
        if (S.stats_acc) { S.stats_acc[3] += kl; S.stats_acc[4] += gate ? 1.0f : 0.0f; }
    } else if (d == -4 && net == 0) {                       // entropy_loss
        R.stats[2] = s;
        if (S.stats_acc) S.stats_acc[2] += s;
    }
}

// Two Adam optimisers on the flat buffers (torch.optim.Adam defaults: betas 0.9 / 0.999, eps 1e-8, no weight decay),
// the actor's step gated by approx_kl <= 1.5 target_kl (ppo_utils.py:126-131).
struct AdamArgs {
    float* p; const float* g; float* m; float* v; int n; int n_actor; float lr_actor, lr_critic;
    float* steps;           // [2] actor, critic step counts (float, as the graphed torch version kept them)
    float target_kl;
    float* stats_acc; const float* stats;       // running sums over the update: 3 losses, kl, actor steps taken
    unsigned int* done;     // block counter (0 between launches): the last block to finish advances the step counts
    float gscale;           // every read of `g` (gradients and the approx-KL slot) is scaled by this: 1 / world after a SUM all-reduce
};

// Every thread reads the step counts before its block signs off; the LAST block to sign off advances them (and the running
// statistics) and re-arms the counter — one launch instead of an update kernel plus a one-thread bookkeeping kernel.
__global__ __launch_bounds__(256) void adam_gated_kernel(const AdamArgs A) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const float kl = A.g[A.n] * A.gscale;
    const bool gate = A.target_kl <= 0.0f || kl <= 1.5f * A.target_kl;
    const float t_actor = A.steps[0] + 1.0f, t_critic = A.steps[1] + 1.0f;
    if (e < A.n) {
        const bool critic = e >= A.n_actor;
        if (critic || gate) {
            const float g = __fmul_rn(A.g[e], A.gscale);
            float p = A.p[e], m = A.m[e], v = A.v[e];
            adam_element(p, g, m, v, critic ? A.lr_critic : A.lr_actor, critic ? t_critic : t_actor);
            A.m[e] = m; A.v[e] = v; A.p[e] = p;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(A.done, 1u) == gridDim.x - 1) {
            A.steps[0] = t_actor - (gate ? 0.0f : 1.0f);
            A.steps[1] = t_critic;
            if (A.stats_acc) {
                A.stats_acc[0] += A.stats[0]; A.stats_acc[1] += A.stats[1]; A.stats_acc[2] += A.stats[2]; A.stats_acc[3] += A.stats[3];
                A.stats_acc[4] += gate ? 1.0f : 0.0f;
            }
            __threadfence();
            *A.done = 0u;
        }
    }
}

// ------------------------------------------------------------------ minibatch permutation
// out[i] = pi(i), i < count, pi a keyed pseudo-random permutation of [0, n): a 6-round Feistel network on the 2 * half
// bits that cover n, cycle-walked back into range.  Replaces torch.randperm in the update loop (SubsetRandomSampler of
// ppo_utils.py:358-371): a sort of 2^19 random keys is ~36 kernel launches per epoch, this is one.
__device__ __forceinline__ uint32_t perm_mix(uint32_t x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
// KEYED: the 64-bit key is derived on the device from a two-word state {base key, epochs drawn so far} the caller keeps in device
// memory — key = base + 0x9E3779B97F4A7C15 (drawn + epoch_offset + 1) — so that a launch captured in a HIP graph shuffles
// differently at every replay (the caller advances `drawn` with a stream-ordered add after its epochs).
template <bool KEYED>
__global__ __launch_bounds__(256) void permutation_kernel(int32_t* __restrict__ out, uint32_t n, uint32_t count, int half,
                                                           uint32_t k0, uint32_t k1, const uint64_t* __restrict__ key_state,
                                                           uint32_t epoch_offset) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    if constexpr (KEYED) {
        const uint64_t key = key_state[0] + 0x9E3779B97F4A7C15ull * (key_state[1] + (uint64_t)epoch_offset + 1ull);
        k0 = (uint32_t)key; k1 = (uint32_t)(key >> 32);
    }
    const uint32_t mask = (1u << half) - 1u;
    uint32_t x = i;
    do {
        uint32_t l = x >> half, r = x & mask;
#pragma unroll
        for (int round = 0; round < 6; ++round) {
            const uint32_t f = perm_mix(r ^ (round & 1 ? k1 : k0) ^ (0x9e3779b9u * (uint32_t)(round + 1))) & mask;
            const uint32_t nl = r;
            r = l ^ f;
            l = nl;
        }
        x = (l << half) | r;
    } while (x >= n);
    out[i] = (int32_t)x;
}

// ------------------------------------------------------------------ the collector's post-processing
// Between the rollout and the update PPO.train_step (controllers/ppo/ppo.py:276-300) derives, elementwise over the [T][N] rollout, the
// time-limit flags, masks, bootstrap terms, the advantage moments and the normalised advantages — ~30 PyTorch kernels of 4-11 us each
// (5 % of an iteration at 65 536 envs).  Here: one kernel in front of the bootstrap critic pass, three behind scg_gae.
//   prepare:    trunc = done & (flags & 1)  (time truncation is not termination, ppo.py:276-283), mask = 1 - done, rew_out = rew (scg_gae
//               adds gamma * terminal_v to it in place, like the reference does to its buffer), v_out = v_all[:T]
//   moments:    block partials of sum(adv), sum(adv^2) and of the first four columns of the per-env episode accumulators, which are zeroed
//   finish:     fixed-order sum of the partials -> moments = {sum, sum of squares, count}; the episode totals are added to the running ones
//   normalise:  adv <- (adv - mean) / (std + 1e-6), population std (ppo.py:300), from the moments (after the caller's all-reduce, if any)
constexpr int RET_BLOCKS = 256;
__global__ __launch_bounds__(256) void returns_prepare_kernel(const uint8_t* __restrict__ done, const uint8_t* __restrict__ flags,
                                                               const float* __restrict__ rew, const float* __restrict__ v_all, int M,
                                                               uint8_t* __restrict__ trunc, float* __restrict__ mask,
                                                               float* __restrict__ rew_out, float* __restrict__ v_out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
        const uint8_t d = done[i];
        trunc[i] = (uint8_t)((flags[i] & 1) & d);
        mask[i] = 1.0f - (float)d;
        rew_out[i] = rew[i];
        v_out[i] = v_all[i];
    }
}
__device__ __forceinline__ float block_sum256(float v, float* red) {       // fixed order: butterfly inside the wave, waves 0..3 in order
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return s;
}
__global__ __launch_bounds__(256) void returns_moments_kernel(const float* __restrict__ adv, int M, float* __restrict__ episode_acc, int N,
                                                               float* __restrict__ partials) {
    __shared__ float red[4];
    float s1 = 0.0f, s2 = 0.0f, e[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) { const float a = adv[i]; s1 += a; s2 = __builtin_fmaf(a, a, s2); }
    if (episode_acc) {
        for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
            f32x4* const row = reinterpret_cast<f32x4*>(episode_acc + (size_t)n * 8);
            const f32x4 a = row[0];
            e[0] += a.x; e[1] += a.y; e[2] += a.z; e[3] += a.w;
            row[0] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; row[1] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        }
    }
    float out[6] = {block_sum256(s1, red), block_sum256(s2, red), block_sum256(e[0], red), block_sum256(e[1], red), block_sum256(e[2], red),
                    block_sum256(e[3], red)};
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) partials[blockIdx.x * 8 + k] = out[k];
    }
}
__global__ __launch_bounds__(256) void returns_finish_kernel(const float* __restrict__ partials, int n_blocks, float count, float* __restrict__ moments,
                                                              float* __restrict__ ep_totals) {
    // thread b holds block b's six partial sums; six fixed-order block sums (one thread walking the 256 partials of a column was 256
    // dependent loads: 32 us)
    __shared__ float red[4];
    const int b = threadIdx.x;
    float v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = b < n_blocks ? partials[b * 8 + k] : 0.0f;
    float s[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] = block_sum256(v[k], red);
    if (b == 0) {
        moments[0] = s[0]; moments[1] = s[1]; moments[2] = count;
        if (ep_totals) { ep_totals[0] += s[2]; ep_totals[1] += s[3]; ep_totals[2] += s[4]; ep_totals[3] += s[5]; }
    }
}
__global__ __launch_bounds__(256) void returns_normalise_kernel(const float* __restrict__ adv, const float* __restrict__ moments, int M,
                                                                 float* __restrict__ out) {
    const float mean = moments[0] / moments[2];
    const float var = fmaxf(moments[1] / moments[2] - mean * mean, 0.0f);
    const float den = sqrtf(var) + 1e-6f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) out[i] = (adv[i] - mean) / den;
}

extern "C" int scg_ppo_returns_prepare(const uint8_t* d_done, const uint8_t* d_flags, const float* d_rew, const float* d_v_all, int T, int N,
                                       uint8_t* d_trunc, float* d_mask, float* d_rew_out, float* d_v_out, void* stream) {
    if (!d_done || !d_flags || !d_rew || !d_v_all || !d_trunc || !d_mask || !d_rew_out || !d_v_out || T <= 0 || N <= 0)
        return fail(-1, "scg_ppo_returns_prepare: bad argument");
    const long long M = (long long)T * N;
    if (M > 0x7fffffffLL) return fail(-1, "scg_ppo_returns_prepare: T x N too large");
    const int grid = (int)std::min<long long>((M + 255) / 256, 4096);
    returns_prepare_kernel<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(d_done, d_flags, d_rew, d_v_all, (int)M, d_trunc, d_mask, d_rew_out, d_v_out);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" size_t scg_ppo_returns_scratch_bytes(void) { return (size_t)RET_BLOCKS * 8 * sizeof(float); }
extern "C" int scg_ppo_returns_moments(const float* d_adv, int T, int N, float* d_episode_acc, float* d_scratch, float* d_moments,
                                       float* d_episode_totals, void* stream) {
    if (!d_adv || !d_scratch || !d_moments || T <= 0 || N <= 0) return fail(-1, "scg_ppo_returns_moments: bad argument");
    const long long M = (long long)T * N;
    if (M > 0x7fffffffLL) return fail(-1, "scg_ppo_returns_moments: T x N too large");
    const int grid = (int)std::min<long long>((M + 255) / 256, RET_BLOCKS);
    returns_moments_kernel<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(d_adv, (int)M, d_episode_acc, N, d_scratch);
    static_assert(RET_BLOCKS <= 256, "one partial row per thread of the finishing block");
    returns_finish_kernel<<<dim3(1), dim3(256), 0, (hipStream_t)stream>>>(d_scratch, grid, (float)M, d_moments, d_episode_acc ? d_episode_totals : nullptr);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int scg_ppo_returns_normalise(const float* d_adv, const float* d_moments, int T, int N, float* d_out, void* stream) {
    if (!d_adv || !d_moments || !d_out || T <= 0 || N <= 0) return fail(-1, "scg_ppo_returns_normalise: bad argument");
    const long long M = (long long)T * N;
    if (M > 0x7fffffffLL) return fail(-1, "scg_ppo_returns_normalise: T x N too large");
    const int grid = (int)std::min<long long>((M + 255) / 256, 4096);
    returns_normalise_kernel<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(d_adv, d_moments, (int)M, d_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ C ABI
extern "C" int scg_random_permutation(int32_t* d_out, int n, int count, uint64_t key, void* stream) {
    if (!d_out || n <= 0 || count < 0 || count > n) return fail(-1, "scg_random_permutation: bad argument");
    if (count == 0) return 0;
    int bits = 1;
    while (bits < 31 && (1u << bits) < (uint32_t)n) ++bits;
    const int half = (bits + 1) / 2;
    permutation_kernel<false><<<dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(
        d_out, (uint32_t)n, (uint32_t)count, half, (uint32_t)key, (uint32_t)(key >> 32), nullptr, 0u);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int scg_random_permutation_keyed(int32_t* d_out, int n, int count, const uint64_t* d_key_state, uint32_t epoch_offset, void* stream) {
    if (!d_out || !d_key_state || n <= 0 || count < 0 || count > n) return fail(-1, "scg_random_permutation_keyed: bad argument");
    if (count == 0) return 0;
    int bits = 1;
    while (bits < 31 && (1u << bits) < (uint32_t)n) ++bits;
    const int half = (bits + 1) / 2;
    permutation_kernel<true><<<dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(
        d_out, (uint32_t)n, (uint32_t)count, half, 0u, 0u, d_key_state, epoch_offset);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int scg_mlp_forward(const float* d_params, const scg_mlp_layout* layout, int nout, const float* d_x, int m,
                               float* d_out, const uint8_t* d_row_mask, void* stream) {
    if (!d_params || !layout || !d_x || !d_out || m <= 0) return fail(-1, "scg_mlp_forward: bad argument");
    // Two CUs are left free (the tiles are walked with a grid stride, any width is correct): a workgroup holds most of a CU's LDS, and the
    // asynchronous evaluation (ppo.AsyncEvaluator: two long-running workgroups on a side stream) would otherwise make the last two
    // workgroups of a chip-wide launch queue behind it — the whole pass then takes up to twice as long (ppo_grad_kernel's 127 + 127
    // workgroups leave the same two CUs).
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 2) n_cu = v;
        else n_cu = 256;
    }
    const int grid = std::min(std::max(n_cu - 2, 1), (m + 32 * FWD_WAVES - 1) / (32 * FWD_WAVES));
    hipStream_t st = (hipStream_t)stream;
    if (nout == NU) {
        const size_t bytes = MlpLds<NIN, HID, NU>::END * sizeof(float);
        static scg::PerDeviceOnce set_a;    // (once per device: the attribute call costs more host time than the launch)
        int dev;
        if (set_a.pending(&dev)) { HIP_TRY(hipFuncSetAttribute((const void*)mlp_forward_kernel<NU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)); set_a.commit(dev); }
        mlp_forward_kernel<NU><<<dim3(grid), dim3(64 * FWD_WAVES), bytes, st>>>(d_params, *layout, d_x, m, d_out, d_row_mask);
    } else if (nout == 1) {
        const size_t bytes = MlpLds<NIN, HID, 1>::END * sizeof(float);
        static scg::PerDeviceOnce set_c;
        int dev;
        if (set_c.pending(&dev)) { HIP_TRY(hipFuncSetAttribute((const void*)mlp_forward_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)); set_c.commit(dev); }
        mlp_forward_kernel<1><<<dim3(grid), dim3(64 * FWD_WAVES), bytes, st>>>(d_params, *layout, d_x, m, d_out, d_row_mask);
    } else {
        return fail(-1, "scg_mlp_forward: this library serves nout = act_dim or 1");
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" size_t scg_ppo_grad_workspace_bytes(int n_workgroups) {
    size_t b = (size_t)n_workgroups * 2 * PARTIAL_STRIDE * sizeof(float);
#ifdef SCG_L_TIMING
    b += 256;
#endif
    return b;
}

extern "C" int scg_ppo_grad(const scg_ppo_grad_args* a, void* stream) {
    if (!a || !a->d_params || !a->d_obs || !a->d_act || !a->d_logp_old || !a->d_adv || !a->d_ret || !a->d_v_old || !a->d_idx ||
        !a->d_workspace || !a->d_grad || !a->d_stats)
        return fail(-1, "scg_ppo_grad: NULL argument");
    if (a->batch <= 0 || a->batch % 32 != 0) return fail(-1, "scg_ppo_grad: the minibatch size must be a positive multiple of 32");
    if (a->n_workgroups <= 0) return fail(-1, "scg_ppo_grad: n_workgroups must be positive");
    hipStream_t st = (hipStream_t)stream;
    GradArgs G;
    G.params = a->d_params; G.actor = a->actor; G.critic = a->critic; G.logstd_off = a->logstd_off;
    G.obs = a->d_obs; G.act = a->d_act; G.logp_old = a->d_logp_old; G.adv = a->d_adv; G.ret = a->d_ret; G.v_old = a->d_v_old;
    G.idx = a->d_idx; G.batch = a->batch; G.clip_param = a->clip_param; G.use_clipped_value = a->use_clipped_value;
    G.partials = (float*)a->d_workspace;
    if (int rc = launch_grad(G, a->n_workgroups, st)) return rc;
    ReduceArgs R;
    R.partials = G.partials; R.n_wg = a->n_workgroups; R.actor = a->actor; R.critic = a->critic; R.logstd_off = a->logstd_off;
    R.n_params = a->n_params; R.entropy_coef = a->entropy_coef; R.params = a->d_params; R.grad = a->d_grad; R.stats = a->d_stats;
    ppo_reduce_kernel<<<dim3((PARTIAL_STRIDE + 63) / 64, 2), dim3(256), 0, st>>>(R);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int scg_ppo_step(const scg_ppo_grad_args* a, float* d_m, float* d_v, float lr_actor, float lr_critic, const float* d_steps_in,
                            float* d_steps_out, float target_kl, float* d_stats_acc, void* stream) {
    if (!a || !a->d_params || !a->d_obs || !a->d_act || !a->d_logp_old || !a->d_adv || !a->d_ret || !a->d_v_old || !a->d_idx ||
        !a->d_workspace || !a->d_grad || !a->d_stats || !d_m || !d_v || !d_steps_in || !d_steps_out || d_steps_in == d_steps_out)
        return fail(-1, "scg_ppo_step: NULL argument (or steps_in == steps_out: the step counts are double-buffered)");
    if (a->batch <= 0 || a->batch % 32 != 0) return fail(-1, "scg_ppo_step: the minibatch size must be a positive multiple of 32");
    if (a->n_workgroups <= 0) return fail(-1, "scg_ppo_step: n_workgroups must be positive");
    hipStream_t st = (hipStream_t)stream;
    GradArgs G;
    G.params = a->d_params; G.actor = a->actor; G.critic = a->critic; G.logstd_off = a->logstd_off;
    G.obs = a->d_obs; G.act = a->d_act; G.logp_old = a->d_logp_old; G.adv = a->d_adv; G.ret = a->d_ret; G.v_old = a->d_v_old;
    G.idx = a->d_idx; G.batch = a->batch; G.clip_param = a->clip_param; G.use_clipped_value = a->use_clipped_value;
    G.partials = (float*)a->d_workspace;
    if (int rc = launch_grad(G, a->n_workgroups, st)) return rc;
    StepArgs S;
    S.R.partials = G.partials; S.R.n_wg = a->n_workgroups; S.R.actor = a->actor; S.R.critic = a->critic; S.R.logstd_off = a->logstd_off;
    S.R.n_params = a->n_params; S.R.entropy_coef = a->entropy_coef; S.R.params = a->d_params; S.R.grad = a->d_grad; S.R.stats = a->d_stats;
    S.p = const_cast<float*>(a->d_params); S.m = d_m; S.v = d_v; S.lr_actor = lr_actor; S.lr_critic = lr_critic;
    S.steps_in = d_steps_in; S.steps_out = d_steps_out; S.target_kl = target_kl; S.stats_acc = d_stats_acc;
    ppo_reduce_adam_kernel<<<dim3((PARTIAL_STRIDE + 63) / 64, 2), dim3(256), 0, st>>>(S);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int scg_adam_gated_scaled(float* d_p, const float* d_g, float* d_m, float* d_v, int n, int n_actor, float lr_actor,
                                     float lr_critic, float* d_steps, float target_kl, float* d_stats_acc, const float* d_stats,
                                     uint32_t* d_block_counter, float grad_scale, void* stream) {
    if (!d_p || !d_g || !d_m || !d_v || !d_steps || !d_block_counter || n <= 0) return fail(-1, "scg_adam_gated: bad argument");
    AdamArgs A{d_p, d_g, d_m, d_v, n, n_actor, lr_actor, lr_critic, d_steps, target_kl, d_stats_acc, d_stats, d_block_counter, grad_scale};
    adam_gated_kernel<<<dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(A);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int scg_adam_gated(float* d_p, const float* d_g, float* d_m, float* d_v, int n, int n_actor, float lr_actor,
                              float lr_critic, float* d_steps, float target_kl, float* d_stats_acc, const float* d_stats,
                              uint32_t* d_block_counter, void* stream) {
    return scg_adam_gated_scaled(d_p, d_g, d_m, d_v, n, n_actor, lr_actor, lr_critic, d_steps, target_kl, d_stats_acc, d_stats,
                                 d_block_counter, 1.0f, stream);
}

// scg_mlp.h — two-hidden-layer MLPs (the reference's actor / critic: math_and_models/neural_networks.py:18-54,
// controllers/ppo/ppo_utils.py:149-199) on the CDNA4 matrix cores, exact float32.
//
// Why MFMA here and nowhere in the simulator: obs -> H -> H -> out for a batch of samples IS a dense contraction.
// gfx950 has f32-input MFMA (v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] B[2x32], 64 cycles, bit-identical to an fmaf
// chain) at the f32 vector peak but with one operand register per lane per 2048 FMAs instead of two per FMA, so a wave
// keeps the whole layer in registers.
//
// Orientation: every layer is computed TRANSPOSED, Y^T[features x samples] = W[out x in] . X^T[in x samples]:
//   A operand = weights  (lane l: A[i = l & 31][k = l >> 5]),
//   B operand = activations (lane l: B[k = l >> 5][j = l & 31], j = sample column),
//   D tile 32 features x 32 samples: lane (c = l & 31, h = l >> 5), register q holds feature row(q, h) = (q & 3) +
//   8 (q >> 2) + 4 h of sample c.
// The point of this orientation: register q of a D tile is, as it stands, the B operand of step (tile, q) of the NEXT
// layer (the two lane halves supply the two k rows row(q,0), row(q,1)) — activations never leave the registers and are
// never shuffled; only the weights are laid out to match (packed once per workgroup into LDS):
//   Wf[rho][tau][lane (i, h)][q] = W[out = 32 rho + i][in = 32 tau + row(q, h)]   (a lane's 16 operands of a tile are
//   contiguous: four ds_read_b128; the lane stride is 20 words so that those reads are bank-conflict free).
// The same array serves the backward data-gradient product dH = W^T dZ through a gather (lane (i', h') reads word q(i') of
// lane row(q', h') + 32 h(i')); with the 20-word stride that gather is a 2-way bank conflict.
// Weight gradients contract over SAMPLES, which sit on the lanes here: tiles are transposed through a wave-private
// 32 x 33 LDS scratch (lane = feature, registers = sample pairs) and fed to the same instruction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { MLP_ACT_TANH = 0, MLP_ACT_RELU = 1, MLP_ACT_LEAKY_RELU = 2 };

__device__ __forceinline__ constexpr int d_row(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

template <int ACT>
__device__ __forceinline__ float mlp_act(float x) {
    if constexpr (ACT == MLP_ACT_TANH) {
        // tanh x = 1 - 2 / (exp(2x) + 1): saturates correctly at both ends, absolute error ~1e-7
        const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);      // exp(2x)
        return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
    } else if constexpr (ACT == MLP_ACT_RELU) {
        return fmaxf(x, 0.0f);
    } else {
        return x > 0.0f ? x : 0.01f * x;                                        // F.leaky_relu default slope
    }
}
// derivative from the OUTPUT y = act(x) (what autograd keeps for tanh; equivalent for the piecewise-linear ones)
template <int ACT>
__device__ __forceinline__ float mlp_dact(float y) {
    if constexpr (ACT == MLP_ACT_TANH) return 1.0f - y * y;
    else if constexpr (ACT == MLP_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
    else return y > 0.0f ? 1.0f : 0.01f;
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// One MLP's parameters in torch's nn.Linear layout (weight [out][in] row-major).
struct MlpWeights {
    const float* W1; const float* b1;      // [H][NIN], [H]
    const float* W2; const float* b2;      // [H][H], [H]
    const float* W3; const float* b3;      // [NOUT][H], [NOUT]
};

// LDS image of one MLP (float words).  NIN <= 32, H multiple of 32, NOUT <= 8 (SAC's actor head: mean | log-std).
// SS: lane stride (words) of the layer-2 image — 20 (= 16 + 4: conflict-free b128 reads, 2-way backward gather) for the
// learner, 16 for forward-only users that must fit a simulator's own LDS next to it (4-way conflicts on 4 reads per
// 16 MFMAs: invisible).
template <int NIN, int H, int NOUT, int SS = 20>
struct MlpLds {
    static constexpr int NT = H / 32;                           // feature tiles per hidden layer
    static constexpr int L1Q = 4 * ((NIN + 7) / 8);             // layer-1 steps: registers q < L1Q cover input rows < NIN
    static constexpr int S = SS;
    static constexpr int TILE2 = 64 * S;                        // words per (rho, tau) tile
    static constexpr int W1F = 0;                               // [NT][L1Q][64]
    static constexpr int W2F = W1F + NT * L1Q * 64;             // [NT][NT][64 lanes][S]: word q of lane (i, h)
    static constexpr int W3 = W2F + NT * NT * TILE2;            // [NOUT][H]
    static constexpr int B1 = W3 + NOUT * H;
    static constexpr int B2 = B1 + H;
    static constexpr int B3 = B2 + H;
    static constexpr int END = (B3 + NOUT + 3) / 4 * 4;         // 16-byte granules
    static_assert(NIN <= 32 && H % 32 == 0 && NOUT <= 8, "unsupported MLP shape");
};

// loads in flight per thread and chunk of the W2 fill: the largest divisor of `it` that is <= 32
constexpr int fill_chunk(int it) {
    int c = it < 32 ? it : 32;
    while (it % c) --c;
    return c;
}

// Cooperative fill of the LDS image from torch-layout parameters (all NTHR threads of the workgroup; caller barriers after).
// Global reads run along the rows of W (coalesced), the permutation is applied on the LDS side.
// Request schedule (round 6): the fill is memory round trips, so they are made to overlap — W1, W3, the biases and the FIRST chunk of W2
// are all requested before the first LDS store; every later W2 chunk is requested before the previous one is stored.  (History: written
// element by element — load, permute, store — the fill exposed one round trip per element, ~1 us x 64 at one wave per SIMD; with each
// PHASE issuing its loads before its stores it was still five dependent round trips, 3.8 of the gradient kernel's 46 us per call.)
// Every load is unconditional on a clamped address, the zero applied afterwards: as `cond ? load : 0` each load sat in its own
// exec-masked branch, and in a kernel that spills a scratch reload + s_waitcnt vmcnt(0) in every branch serialised them.
// `between()` runs once, after the first batch of requests has been issued and before the first LDS store: the caller's own early requests
// (a first tile's rows, whose index load it issued before the call) go there and arrive under the rest of the fill.
struct MlpFillNothing { __device__ __forceinline__ void operator()() const {} };
template <int NIN, int H, int NOUT, int SS = 20, int NTHR = 256, typename Between = MlpFillNothing>
__device__ __forceinline__ void mlp_fill_lds(float* lds, const MlpWeights& w, int tid, Between between = Between()) {
    using L = MlpLds<NIN, H, NOUT, SS>;
    // ---- requests: W1F[rho][q][lane(i,h)] = W1[32 rho + i][row(q,h)]
    constexpr int N1 = L::NT * L::L1Q * 64, IT1 = (N1 + NTHR - 1) / NTHR;
    float v1[IT1];
#pragma unroll
    for (int it = 0; it < IT1; ++it) {
        const int k = tid + it * NTHR, kc = k < N1 ? k : N1 - 1;
        const int lane = kc & 63, q = (kc >> 6) % L::L1Q, rho = (kc >> 6) / L::L1Q;
        const int in = d_row(q, lane >> 5);
        const float x = w.W1[(32 * rho + (lane & 31)) * NIN + (in < NIN ? in : 0)];
        v1[it] = (k < N1 && in < NIN) ? x : 0.0f;
    }
    // ---- W3, biases
    constexpr int N3 = NOUT * H, IT3 = (N3 + NTHR - 1) / NTHR;
    float v3[IT3];
#pragma unroll
    for (int it = 0; it < IT3; ++it) { const int k = tid + it * NTHR; v3[it] = w.W3[k < N3 ? k : N3 - 1]; }
    static_assert(H <= NTHR && NOUT <= NTHR, "one bias word per thread and layer");
    const float vb1 = w.b1[tid < H ? tid : H - 1], vb2 = w.b2[tid < H ? tid : H - 1], vb3 = w.b3[tid < NOUT ? tid : NOUT - 1];
    // ---- W2 in chunks (source order: W2[o][in], in fastest), one chunk in flight ahead of the stores
    static_assert((H * H) % NTHR == 0, "workgroup size must divide H * H");
    constexpr int IT = H * H / NTHR, CH = fill_chunk(IT);             // H = 96: 36 loads per thread in two chunks of 18
    static_assert(IT % CH == 0, "chunking");
    float cur[CH], nxt[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) cur[j] = w.W2[tid + j * NTHR];
    between();
    // ---- stores of the first batch
#pragma unroll
    for (int it = 0; it < IT1; ++it) {
        const int k = tid + it * NTHR;
        if (k < N1) lds[L::W1F + k] = v1[it];
    }
#pragma unroll
    for (int it = 0; it < IT3; ++it) { const int k = tid + it * NTHR; if (k < N3) lds[L::W3 + k] = v3[it]; }
    if (tid < H) { lds[L::B1 + tid] = vb1; lds[L::B2 + tid] = vb2; }
    if (tid < NOUT) lds[L::B3 + tid] = vb3;
#pragma unroll
    for (int base = 0; base < IT; base += CH) {
        if (base + CH < IT) {
#pragma unroll
            for (int j = 0; j < CH; ++j) nxt[j] = w.W2[tid + (base + CH + j) * NTHR];
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int k = tid + (base + j) * NTHR;
            const int o = k / H, in = k % H;
            const int rho = o >> 5, i = o & 31, tau = in >> 5, r = in & 31;
            const int q = 4 * (r >> 3) + (r & 3), h = (r >> 2) & 1;
            lds[L::W2F + (rho * L::NT + tau) * L::TILE2 + (i + 32 * h) * L::S + q] = cur[j];
        }
        if (base + CH < IT) {
#pragma unroll
            for (int j = 0; j < CH; ++j) cur[j] = nxt[j];
        }
    }
}

// The 16 A operands (q = 0..15) of lane `lane` for the layer-2 tile (rho, tau): four 16-byte LDS reads.
template <typename L>
__device__ __forceinline__ void load_w2_tile(const float* lds, int rho, int tau, int lane, float* a) {
    const float* p = lds + L::W2F + (rho * L::NT + tau) * L::TILE2 + lane * L::S;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * g);
        a[4 * g + 0] = v.x; a[4 * g + 1] = v.y; a[4 * g + 2] = v.z; a[4 * g + 3] = v.w;
    }
}

enum { MLP_ACT_NONE = 3 };      // (second hidden layer without activation: the reference's SAC actor trunk, sac_utils.py:190-200 over
                                //  neural_networks.py:50-54 — MLP applies no activation after ITS last layer)
template <>
__device__ __forceinline__ float mlp_act<MLP_ACT_NONE>(float x) { return x; }
template <>
__device__ __forceinline__ float mlp_dact<MLP_ACT_NONE>(float) { return 1.0f; }

// Forward pass of one 32-sample column tile.  x[q] (q < L1Q): input feature row(q, h) of this lane's sample.
// Leaves h1, h2 (activations, D layout: [tile][q]) and out[NOUT] (identical in both lane halves).  ACT2: activation of
// the second hidden layer when it differs from the first's.
// PF: the weight operand of layer-2 block (rho, tau) is read one block AHEAD, in front of the previous block's products (16 more
// registers) — for callers that run one wave per SIMD, where every 16-MFMA chain otherwise begins with an exposed LDS round trip.
template <int NIN, int H, int NOUT, int ACT, int SS = 20, int ACT2 = ACT, bool PF = false>
__device__ __forceinline__ void mlp_forward_tile(const float* lds, const float* x, f32x16* h1, f32x16* h2, float* out, int lane,
                                                 unsigned long long* ts = nullptr) {
    using L = MlpLds<NIN, H, NOUT, SS>;
    constexpr int NT = L::NT;
    const int h = lane >> 5;
    // ---- layer 1
#pragma unroll
    for (int rho = 0; rho < NT; ++rho) {
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {                                    // bias: rows 8g + 4h + (0..3)
            const f32x4 b = *reinterpret_cast<const f32x4*>(lds + L::B1 + 32 * rho + 8 * g + 4 * h);
            acc[4 * g + 0] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
        }
#pragma unroll
        for (int q = 0; q < L::L1Q; ++q) acc = mfma32(lds[L::W1F + (rho * L::L1Q + q) * 64 + lane], x[q], acc);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = mlp_act<ACT>(acc[q]);
        h1[rho] = acc;
    }
    if (ts) ts[0] = __builtin_readcyclecounter();
    // ---- layer 2
    float a_nx[16];
    if constexpr (PF) load_w2_tile<L>(lds, 0, 0, lane, a_nx);
#pragma unroll
    for (int rho = 0; rho < NT; ++rho) {
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(lds + L::B2 + 32 * rho + 8 * g + 4 * h);
            acc[4 * g + 0] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
        }
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            float a[16];
            if constexpr (PF) {
#pragma unroll
                for (int q = 0; q < 16; ++q) a[q] = a_nx[q];
                if (tau + 1 < NT) load_w2_tile<L>(lds, rho, tau + 1, lane, a_nx);
                else if (rho + 1 < NT) load_w2_tile<L>(lds, rho + 1, 0, lane, a_nx);
            } else {
                load_w2_tile<L>(lds, rho, tau, lane, a);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) acc = mfma32(a[q], h1[tau][q], acc);
            __builtin_amdgcn_sched_barrier(0);                  // keep the scheduler from hoisting every tile's operand loads
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = mlp_act<ACT2>(acc[q]);
        h2[rho] = acc;
    }
    if (ts) ts[1] = __builtin_readcyclecounter();
    // ---- output layer on the vector unit: each lane holds half of its sample's features
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        float s = 0.0f;
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(lds + L::W3 + o * H + 32 * tau + 8 * g + 4 * h);
                s = __builtin_fmaf(w.x, h2[tau][4 * g + 0], s); s = __builtin_fmaf(w.y, h2[tau][4 * g + 1], s);
                s = __builtin_fmaf(w.z, h2[tau][4 * g + 2], s); s = __builtin_fmaf(w.w, h2[tau][4 * g + 3], s);
            }
        }
        s += __shfl_xor(s, 32, 64);
        out[o] = s + lds[L::B3 + o];
    }
}

// D-layout tile -> "lane = feature, registers = samples" through the wave-private scratch (TR_WORDS words, 16-byte aligned):
// t[q] = tile[feature i = lane & 31][sample d_row(q, lane >> 5)] — the samples in the accumulator's own row order, so that a
// transposed tile is again a valid MFMA operand over sample pairs (contraction index <-> (q, lane half), as in the forward
// pass) and the read side is four 16-byte loads per lane (row i of the scratch = the 32 samples of feature i; the row stride
// of 36 words puts the sixteen lanes of one LDS pass on sixty-four different banks).
constexpr int TR_STRIDE = 36, TR_WORDS = 32 * TR_STRIDE;
__device__ __forceinline__ void tile_transpose(float* scr, const f32x16& tile, float* t, int lane) {
    const int c = lane & 31, h = lane >> 5;
    float* const wr = scr + 4 * h * TR_STRIDE + c;
#pragma unroll
    for (int q = 0; q < 16; ++q) wr[d_row(q, 0) * TR_STRIDE] = tile[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float* const rd = scr + c * TR_STRIDE + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(rd + 8 * g);
        t[4 * g] = v.x; t[4 * g + 1] = v.y; t[4 * g + 2] = v.z; t[4 * g + 3] = v.w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// The same, in place: the tile's registers become t[0..15].
__device__ __forceinline__ void tile_transpose_inplace(float* scr, f32x16& tile, int lane) {
    float t[16];
    tile_transpose(scr, tile, t, lane);
#pragma unroll
    for (int q = 0; q < 16; ++q) tile[q] = t[q];
}

}  // namespace scg

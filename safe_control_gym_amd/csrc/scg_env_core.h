// scg_env_core.h — one environment instance per thread: the fused control step.
//
// Device-side statement of the reference's per-env Python hot path (paths relative to
// /root/reference/safe_control_gym/envs), one launch per control step:
//   before_step / _preprocess_control   benchmark_env.py:400-420, gym_pybullet_drones/quadrotor.py:722-775,
//                                       gym_pybullet_drones/quadrotor_utils.py:16-60, gym_control/cartpole.py:479-530
//   disturbances                        disturbances.py:54-259
//   _advance_simulation (PYB_FREQ/CTRL_FREQ engine steps, semi-implicit Euler + Bullet's velocity clamp)
//                                       gym_pybullet_drones/base_aviary.py:232-286,364-384, gym_control/cartpole.py:532-583
//   _get_observation / extend_obs       quadrotor.py:777-817, cartpole.py:585-609, benchmark_env.py:422-445
//   _get_reward / _get_done / _get_info quadrotor.py:819-923, cartpole.py:611-696
//   after_step (constraints, penalty, time limit)   benchmark_env.py:447-502, constraints.py:97-131
//   auto-reset (DummyVecEnv)            env_wrappers/vectorized_env/dummy_vec_env.py:29-41,
//                                       quadrotor.py:328-392, cartpole.py:266-352, benchmark_env.py:237-268,320-359
// Written from the semantics (see oracle/ for the float64 CPU restatement they are tested against),
// not translated: state is SoA in HBM, everything between load and store lives in registers.
//
// Performance notes (gfx950).  At the headline size (65 536 envs = 1024 waves on 1024 SIMDs) every SIMD
// holds ONE wave, so a launch lasts as long as one wave's instruction stream — one instruction of any kind
// (SALU included) per 4-clock issue slot, taken branches ~25 clocks, nothing to overlap the memory phases
// with (profiles/r01_latency_budget.md): the code below is written for a short instruction stream —
//   * the engine substeps never call sin/cos: rpm are constant within a control step, so the attitude
//     advances by small angles d = h*w (|d| <= h*100 by Bullet's velocity clamp) and (sin,cos) / the
//     quaternion are rotated with short Taylor polynomials (error < 1 ulp, guarded by `small_angle`);
//   * reciprocals of mass / inertia are hoisted out of the substep loop, clamps are v_med3;
//   * disturbance code (Philox + Box-Muller per channel) is compiled out of the DIST=false kernels,
//     which every shipped RL config uses; box constraints are evaluated per variable with static
//     register indices; Philox draws at reset serve two (or, without normal draws, four) variables per block;
//   * every per-env array is addressed through a buffer resource (Slot<T>): no VALU address arithmetic.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "scg_params.h"
#include "scg_rng.h"

// box-constraint loop: fully unrolled when the row table is a compile-time constant (specialised build)
#ifndef SCG_Q3_UNROLL
#define SCG_Q3_UNROLL 5     // substeps of the packed 3-D integrator per loop iteration (specialised build)
#endif
#ifdef SCG_SPEC
#define SCG_BOX_UNROLL _Pragma("unroll")
#else
#define SCG_BOX_UNROLL _Pragma("unroll 8")
#endif

namespace scg {

// In-kernel timeline probes (tools/timeline.py): a timing-only build records the shader clock of lane 0 of every
// wave at a few marks.  Compiled out of every shipped library.
#ifdef SCG_EXP_TIMELINE
__device__ unsigned long long scg_timeline[4096 * 8];
__device__ __forceinline__ void scg_tl_mark(int k) {
    asm volatile("" ::: "memory");
    const unsigned long long t = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) scg_timeline[((blockIdx.x * blockDim.x + threadIdx.x) >> 6) % 4096 * 8 + k] = t;
    asm volatile("" ::: "memory");
}
#define SCG_TL(k) scg_tl_mark(k)
#else
#define SCG_TL(k)
#endif

// ------------------------------------------------------------------ math wrappers
__device__ __forceinline__ float m_sin(float x) { return sinf(x); }
__device__ __forceinline__ double m_sin(double x) { return sin(x); }
__device__ __forceinline__ float m_cos(float x) { return cosf(x); }
__device__ __forceinline__ double m_cos(double x) { return cos(x); }
__device__ __forceinline__ void m_sincos(float x, float* s, float* c) { sincosf(x, s, c); }
__device__ __forceinline__ void m_sincos(double x, double* s, double* c) { sincos(x, s, c); }
__device__ __forceinline__ float m_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double m_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float m_rsqrt(float x) { return rsqrtf(x); }
__device__ __forceinline__ double m_rsqrt(double x) { return 1.0 / sqrt(x); }
__device__ __forceinline__ float m_exp(float x) { return expf(x); }
__device__ __forceinline__ double m_exp(double x) { return exp(x); }
__device__ __forceinline__ float m_log(float x) { return logf(x); }
__device__ __forceinline__ double m_log(double x) { return log(x); }
__device__ __forceinline__ float m_pow(float x, float y) { return powf(x, y); }
__device__ __forceinline__ double m_pow(double x, double y) { return pow(x, y); }
__device__ __forceinline__ float m_asin(float x) { return asinf(x); }
__device__ __forceinline__ double m_asin(double x) { return asin(x); }
__device__ __forceinline__ float m_atan2(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ double m_atan2(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ float m_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double m_abs(double x) { return fabs(x); }
__device__ __forceinline__ float m_rint(float x) { return rintf(x); }
__device__ __forceinline__ double m_rint(double x) { return rint(x); }
__device__ __forceinline__ float m_floor(float x) { return floorf(x); }
__device__ __forceinline__ double m_floor(double x) { return floor(x); }
__device__ __forceinline__ float m_clamp(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
__device__ __forceinline__ double m_clamp(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }
template <typename T> __device__ __forceinline__ T m_max(T a, T b) { return a > b ? a : b; }
// x / c and sqrt(x) where the float path may use the 1-ulp hardware forms (v_rcp / v_sqrt, ~10 fewer VALU
// each than the IEEE expansions); the double path keeps the exact operations the oracle performs.
__device__ __forceinline__ float m_div_by(float x, float c) { return x * __builtin_amdgcn_rcpf(c); }
__device__ __forceinline__ double m_div_by(double x, double c) { return x / c; }
__device__ __forceinline__ float m_sqrt_fast(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double m_sqrt_fast(double x) { return sqrt(x); }

// Scheduling fences.  vmcnt retires in order and counts stores as well as loads, so a load whose first use
// sits after a run of stores makes the wave wait for the store acknowledgements too; `vreg_fence` forces the
// wait to the point where it is placed (before the stores).  `sreg_fence` does the same for kernel
// arguments: every s_load is issued in the entry block, in one scalar-memory round.
template <typename V> __device__ __forceinline__ void vreg_fence(const V& x) { asm volatile("" ::"v"(x)); }
template <typename V> __device__ __forceinline__ void sreg_fence(const V& x) { asm volatile("" ::"s"(x)); }

template <typename T> struct Const {
    static constexpr T PI = (T)3.14159265358979323846;
    static constexpr T TWO_PI = (T)6.28318530717958647692;
    static constexpr T HALF_PI = (T)1.57079632679489661923;
};

// math_and_models/normalization.py:8-10: ((x + pi) % (2 pi)) - pi with Python's floored modulo.
template <typename T>
__device__ __forceinline__ T normalize_angle(T x) {
    T y = x + Const<T>::PI;
    y = y - Const<T>::TWO_PI * m_floor(m_div_by(y, Const<T>::TWO_PI));
    return y - Const<T>::PI;
}

// ---- float-path inverse trigonometry for the Euler-angle extraction (getEulerFromQuaternion): the library
// atan2f / asinf cost ~60 instructions each with their full-range handling; these are ~25, absolute error < 3e-7 rad
// (the float observation they feed carries 6e-8 relative rounding already).  The double path keeps the library calls.
__device__ __forceinline__ float fast_atan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    // atan(mn / mx) on [0, 1]: above tan(pi/8) use atan(a) = pi/4 + atan((a - 1) / (a + 1))  (one reciprocal either way)
    const bool hi = mn > 0.41421356f * mx;
    const float num = hi ? mn - mx : mn, den = hi ? mn + mx : mx;
    const float t = (den > 0.0f) ? num * __builtin_amdgcn_rcpf(den) : 0.0f;
    const float z = t * t;
    // Cephes atanf kernel on |t| <= tan(pi/8)
    float r = ((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f;
    r = __builtin_fmaf(r * z, t, t);
    r += hi ? 0.78539816339744831f : 0.0f;
    r = (ay > ax) ? 1.57079632679489662f - r : r;
    r = (x < 0.0f) ? 3.14159265358979324f - r : r;
    return __builtin_copysignf(r, y);
}
__device__ __forceinline__ float fast_asin(float x) {         // |x| < 1
    const float a = fabsf(x);
    const bool big = a > 0.5f;
    // |x| > 1/2: asin(a) = pi/2 - 2 asin(sqrt((1 - a) / 2))
    const float z = big ? 0.5f * (1.0f - a) : a * a;
    const float s = big ? __builtin_amdgcn_sqrtf(z) : a;
    // Cephes asinf kernel: asin(s) = s + s z P(z), z = s^2 <= 1/4
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f);
    float r = __builtin_fmaf(s * z, p, s);
    r = big ? 1.57079632679489662f - 2.0f * r : r;
    return __builtin_copysignf(r, x);
}
__device__ __forceinline__ double fast_atan2(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ double fast_asin(double x) { return asin(x); }
// sincos for reset-time Euler angles: |x| <= pi/4 (every shipped config draws |angle| <= 0.2) takes the two
// Taylor/minimax kernels directly; anything larger goes through the library's argument reduction.
__device__ __forceinline__ void reset_sincos(float x, float* s, float* c) {
    if (fabsf(x) <= 0.78539816f) {
        const float z = x * x;
        *s = __builtin_fmaf(x * z, (-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f, x);
        *c = __builtin_fmaf(z * z, (2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f, 1.0f - 0.5f * z);
    } else {
        sincosf(x, s, c);
    }
}
__device__ __forceinline__ void reset_sincos(double x, double* s, double* c) { sincos(x, s, c); }

// sin(d), cos(d) for |d| <= 0.25 (Taylor; truncation < 1e-17 relative in double, < 6e-8 (1 ulp) in float).
template <typename T>
__device__ __forceinline__ void small_sincos(T d, T& sd, T& cd) {
    const T d2 = d * d;
    if constexpr (sizeof(T) == 8) {
        sd = d * ((T)1 + d2 * ((T)(-1.0 / 6) + d2 * ((T)(1.0 / 120) + d2 * ((T)(-1.0 / 5040) + d2 * ((T)(1.0 / 362880) + d2 * (T)(-1.0 / 39916800))))));
        cd = (T)1 + d2 * ((T)-0.5 + d2 * ((T)(1.0 / 24) + d2 * ((T)(-1.0 / 720) + d2 * ((T)(1.0 / 40320) + d2 * ((T)(-1.0 / 3628800) + d2 * (T)(1.0 / 479001600))))));
    } else {
        sd = d * ((T)1 + d2 * ((T)(-1.0 / 6) + d2 * (T)(1.0 / 120)));
        cd = (T)1 + d2 * ((T)-0.5 + d2 * ((T)(1.0 / 24) + d2 * (T)(-1.0 / 720)));
    }
}
// sinc(a) = sin(a)/a and cos(a) from a2 = a*a, a <= 0.125.
template <typename T>
__device__ __forceinline__ void small_sinc_cos(T a2, T& sinc, T& ca) {
    if constexpr (sizeof(T) == 8) {
        sinc = (T)1 + a2 * ((T)(-1.0 / 6) + a2 * ((T)(1.0 / 120) + a2 * ((T)(-1.0 / 5040) + a2 * (T)(1.0 / 362880))));
        ca = (T)1 + a2 * ((T)-0.5 + a2 * ((T)(1.0 / 24) + a2 * ((T)(-1.0 / 720) + a2 * ((T)(1.0 / 40320) + a2 * (T)(-1.0 / 3628800)))));
    } else {
        sinc = (T)1 + a2 * ((T)(-1.0 / 6) + a2 * (T)(1.0 / 120));
        ca = (T)1 + a2 * ((T)-0.5 + a2 * ((T)(1.0 / 24) + a2 * (T)(-1.0 / 720)));
    }
}

// One lane's element of a device array, addressed the way the CDNA buffer instructions do it: a uniform resource
// (4 SGPRs describing one allocation) + a uniform byte offset of the array / row inside it (SGPR soffset) + a 32-bit
// per-lane byte offset (one VGPR).  `buffer_load/store ... offen` then needs no address arithmetic at all per access
// (the flat/global forms cost a 64-bit VALU add or a quarter-rate v_mad_i64 each: ~110 of ~950 VALU instructions of
// the step kernel).  With one wave per SIMD every instruction, scalar or vector, costs a 4-cycle issue slot, so the
// resources are kept to TWO per kernel: the caller's workspace (all simulator arrays) and the output window (all
// output arrays, when the caller placed them within 4 GiB of each other — HipVecEnv allocates them from one arena;
// otherwise each output gets its own resource, 4 SALU more per array).
// load(k) / store(v, k): element k uniform elements further (row r of an SoA [rows][N] array for k = r*N; component k
// of an AoS [N][n] array when the lane offset was built with n).
// The host rejects configurations whose arrays exceed the 32-bit offsets (validate() in scg_kernels.hip).
typedef unsigned int u32x2 __attribute__((vector_size(8)));
typedef unsigned int u32x4 __attribute__((vector_size(16)));
// Cache policy of the simulator's buffer STORES (gfx942 / gfx950 `aux`: bit 0 = sc0, bit 1 = nt, bit 4 = sc1).
// SCG_ST_AUX = 17 (sc0 | sc1): write-through at system scope.  Each of the 8 XCDs has its own L2, so a kernel's dirty lines
// have to leave for the memory side (Infinity Cache / HBM) before the next kernel — on another XCD — may read them: with the
// default write-back policy that happens as ONE flush at the end of the kernel, behind the last store (65 536 envs,
// Quadrotor2D: 11.5 MB, ~0.9 us of a 5.2 us launch with nothing left to overlap it); written through, every store starts its
// trip when it is issued — the constraint rows at ~55 % of the wave's lifetime, the observation at ~65 % — under the remaining
// arithmetic.  Measured (tools/sessions/s81.sh, same box, alternating runs): Quadrotor2D-track 5.24 -> 4.40 us per launch,
// Quadrotor3D-track 9.85 -> 8.56 us.  Visibility is only ever EARLIER than with the default policy; values are untouched.
#ifndef SCG_ST_AUX
#define SCG_ST_AUX 17
#endif
#ifndef SCG_SEQ_ST_AUX
#define SCG_SEQ_ST_AUX 0        // the K-steps-per-launch kernels (scg_step_sequence, scg_rollout_policy, scg_rollout_random): write-back
#endif
#ifndef SCG_LD_AUX
#define SCG_LD_AUX 0
#endif
__device__ __forceinline__ float buf_ld(__amdgpu_buffer_rsrc_t r, uint32_t v, uint32_t s, float) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, v, s, SCG_LD_AUX));
}
__device__ __forceinline__ int32_t buf_ld(__amdgpu_buffer_rsrc_t r, uint32_t v, uint32_t s, int32_t) {
    return (int32_t)__builtin_amdgcn_raw_buffer_load_b32(r, v, s, SCG_LD_AUX);
}
__device__ __forceinline__ uint32_t buf_ld(__amdgpu_buffer_rsrc_t r, uint32_t v, uint32_t s, uint32_t) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, v, s, SCG_LD_AUX);
}
__device__ __forceinline__ uint8_t buf_ld(__amdgpu_buffer_rsrc_t r, uint32_t v, uint32_t s, uint8_t) {
    return __builtin_amdgcn_raw_buffer_load_b8(r, v, s, SCG_LD_AUX);
}
__device__ __forceinline__ double buf_ld(__amdgpu_buffer_rsrc_t r, uint32_t v, uint32_t s, double) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, v, s, SCG_LD_AUX));
}
template <int AUX = SCG_ST_AUX>
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, uint32_t v, uint32_t s, float x) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, x), r, v, s, AUX);
}
template <int AUX = SCG_ST_AUX>
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, uint32_t v, uint32_t s, int32_t x) {
    __builtin_amdgcn_raw_buffer_store_b32((uint32_t)x, r, v, s, AUX);
}
template <int AUX = SCG_ST_AUX>
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, uint32_t v, uint32_t s, uint32_t x) {
    __builtin_amdgcn_raw_buffer_store_b32(x, r, v, s, AUX);
}
template <int AUX = SCG_ST_AUX>
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, uint32_t v, uint32_t s, uint8_t x) {
    __builtin_amdgcn_raw_buffer_store_b8(x, r, v, s, AUX);
}
template <int AUX = SCG_ST_AUX>
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, uint32_t v, uint32_t s, double x) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, x), r, v, s, AUX);
}

// 16-byte store.  The uniform array offset goes through the VECTOR offset here (one v_add per row, the chunk constants
// fold into the instruction's immediate), NOT through soffset: a VMEM store of more than 64 bits reads its data VGPRs
// over several cycles, a VALU write to them in the next issue slots can overtake the read, and LLVM (ROCm 7.2) pads
// that hazard only when soffset is not an SGPR ("... only exists if the instruction is not using a register in the
// soffset field", GCNHazardRecognizer) — which does not hold on MI355X: `buffer_store_dwordx4 v[42:45], v27, s[40:43],
// s4 offen` + `v_mov_b32 v42, s6` tore 16-64 doubles out of 1.5 M on a cold first launch
// (tests/test_gpu_parity_scale.py; tools/hazard_lint.py proves the pattern absent from every built library).
template <int AUX = SCG_ST_AUX>
__device__ __forceinline__ void buf_st128(u32x4 d, __amdgpu_buffer_rsrc_t r, uint32_t lane_off, uint32_t array_off, uint32_t k) {
    __builtin_amdgcn_raw_buffer_store_b128(d, r, lane_off + array_off + k, 0, AUX);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    // raw buffer (stride 0), no range limit, gfx9 32-bit data format word
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0xffffffff, 0x00020000);
}

// AUX: the cache policy of this slot's STORES (SCG_ST_AUX: write-through, for the one-launch-per-control-step kernels whose
// stores would otherwise leave in one end-of-kernel flush; SCG_SEQ_ST_AUX: write-back, for the K-steps-per-launch kernels, whose
// hundreds of megabytes of stacked outputs stream out under the following control steps anyway and are faster through the L2).
template <typename V, int AUX = SCG_ST_AUX>
struct Slot {
    using U = typename std::remove_const<V>::type;
    template <int A2> __device__ __forceinline__ Slot<V, A2> with() const { return Slot<V, A2>{r, soff, off}; }
    __amdgpu_buffer_rsrc_t r;
    uint32_t soff;  // uniform byte offset of the array inside the resource; SCG_NO_OFF = array not present
    uint32_t off;   // per-lane byte offset
    __device__ __forceinline__ explicit operator bool() const { return soff != SCG_NO_OFF; }
    __device__ __forceinline__ U load(size_t k = 0) const { return buf_ld(r, off, soff + (uint32_t)(k * sizeof(V)), U()); }
    __device__ __forceinline__ void store(U x, size_t k = 0) const { buf_st<AUX>(r, off, soff + (uint32_t)(k * sizeof(V)), x); }
    // element index only known per lane / from LDS (not provably uniform): folded into the lane offset
    __device__ __forceinline__ void store_at(U x, uint32_t k) const { buf_st<AUX>(r, off + k * (uint32_t)sizeof(V), soff, x); }

    // N consecutive elements of this lane (an AoS row whose lane offset was built with elems_per_lane = N, base
    // 16-byte aligned): moved in the widest pieces the row pitch allows (16, 8 or 4 bytes).
    // N consecutive elements of this lane (an AoS row whose lane offset was built with elems_per_lane = N, base
    // 16-byte aligned): moved in the widest pieces the row pitch allows (16, 8 or 4 bytes).  Vector payloads are
    // converted with ONE whole-value bit_cast: indexing the builtin's vector result element-wise (v[1], v[2], ...)
    // miscompiles to element 0 with this clang (ROCm 7.2), for loads of both widths.
    template <int M> struct Pack { U e[M]; };
    template <int N>
    __device__ __forceinline__ void store_row(const U* x) const {
        constexpr int bytes = N * (int)sizeof(U);
        constexpr int W = bytes % 16 == 0 ? 16 : bytes % 8 == 0 ? 8 : (int)sizeof(U);
        constexpr int per = W / (int)sizeof(U);
#pragma unroll
        for (int c = 0; c < bytes / W; ++c) {
            Pack<per> p;
#pragma unroll
            for (int j = 0; j < per; ++j) p.e[j] = x[c * per + j];
            if constexpr (W == 16) buf_st128<AUX>(__builtin_bit_cast(u32x4, p), r, off, soff, (uint32_t)(c * W));
            else if constexpr (W == 8) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, p), r, off, soff + (uint32_t)(c * W), AUX);
            else buf_st<AUX>(r, off, soff + (uint32_t)(c * W), p.e[0]);
        }
    }
    // chunk c (M elements = 16 bytes) of this lane's row
    template <int M>
    __device__ __forceinline__ void store_chunk(const U* x, int c) const {
        static_assert(M * sizeof(U) == 16, "store_chunk moves 16 bytes");
        Pack<M> p;
#pragma unroll
        for (int j = 0; j < M; ++j) p.e[j] = x[j];
        buf_st128<AUX>(__builtin_bit_cast(u32x4, p), r, off, soff, (uint32_t)(c * 16));
    }
    template <int N>
    __device__ __forceinline__ void load_row(U* x) const {
        constexpr int bytes = N * (int)sizeof(U);
        constexpr int W = bytes % 16 == 0 ? 16 : bytes % 8 == 0 ? 8 : (int)sizeof(U);
        constexpr int per = W / (int)sizeof(U);
#pragma unroll
        for (int c = 0; c < bytes / W; ++c) {
            Pack<per> p;
            if constexpr (W == 16) p = __builtin_bit_cast(Pack<per>, __builtin_amdgcn_raw_buffer_load_b128(r, off, soff + (uint32_t)(c * W), SCG_LD_AUX));
            else if constexpr (W == 8) p = __builtin_bit_cast(Pack<per>, __builtin_amdgcn_raw_buffer_load_b64(r, off, soff + (uint32_t)(c * W), SCG_LD_AUX));
            else p.e[0] = buf_ld(r, off, soff + (uint32_t)(c * W), U());
#pragma unroll
            for (int j = 0; j < per; ++j) x[c * per + j] = p.e[j];
        }
    }
};
// slot inside a shared resource (array at byte offset `soff`), and slot over a stand-alone array
template <typename V>
__device__ __forceinline__ Slot<V> slot_in(__amdgpu_buffer_rsrc_t r, uint32_t soff, int lane_index, int elems_per_lane = 1) {
    return Slot<V>{r, soff, (uint32_t)lane_index * (uint32_t)(elems_per_lane * (int)sizeof(V))};
}
template <typename V>
__device__ __forceinline__ Slot<V> slot(V* base, int lane_index, int elems_per_lane = 1) {
    return Slot<V>{make_rsrc(base), base ? 0u : SCG_NO_OFF, (uint32_t)lane_index * (uint32_t)(elems_per_lane * (int)sizeof(V))};
}

// ------------------------------------------------------------------ per-system dimensions
template <int SYS> struct Dims;
template <> struct Dims<SCG_CARTPOLE> { enum { NX = 4, NU = 1, NS = 4, NP = 3, DYN = 2 }; };
template <> struct Dims<SCG_QUAD_1D> { enum { NX = 2, NU = 1, NS = 2, NP = 4, DYN = 1 }; };
template <> struct Dims<SCG_QUAD_2D> { enum { NX = 6, NU = 2, NS = 6, NP = 4, DYN = 2 }; };
template <> struct Dims<SCG_QUAD_3D> { enum { NX = 12, NU = 4, NS = 13, NP = 4, DYN = 3 }; };

// X_GOAL table access: LDS copy (generic build, table fits) or global memory.  Kept as two typed pointers
// and a flag: selecting between an LDS-derived and a global generic pointer trips a gfx950 backend bug
// ("V_CMP_NE_U32 0, src_shared_base: operand has incorrect register class", ROCm 7.2).
template <typename T>
struct GoalTab {
    const T* lds;
    const T* glob;
    bool in_lds;
    __device__ __forceinline__ T operator[](int k) const {
#ifdef SCG_SPEC
        return glob[k];
#else
        return in_lds ? lds[k] : glob[k];
#endif
    }
};

// Device view of scg_step_out with typed pointers.
// Output arrays of one reset / step call (scg_step_out).  OutTabOne: every bound array lies within 4 GiB of `base`
// (off[k] = byte offset, SCG_NO_OFF = not bound) and the kernel addresses them through ONE buffer resource;
// OutTabPtr: each array's own address (nullptr = not bound).
enum { OUT_OBS, OUT_REWARD, OUT_DONE, OUT_FLAGS, OUT_C_VALUES, OUT_MSE, OUT_TERMINAL_OBS, OUT_STATE, OUT_NOISY_ACTION,
       OUT_EP_STATS, OUT_FIN_STATS, OUT_COUNT };
struct OutTabOne { char* base; uint32_t off[OUT_COUNT]; };       // kernel argument of the one-window kernels
struct OutTabPtr { char* ptr[OUT_COUNT]; };                       // ... of the one-resource-per-array kernels
template <bool ONE> struct OutTabOf { using type = OutTabPtr; };
template <> struct OutTabOf<true> { using type = OutTabOne; };

enum : uint8_t { FLAG_TRUNCATED = 1, FLAG_VIOLATION = 2, FLAG_OOB = 4, FLAG_GOAL = 8, FLAG_GROUND = 16 };
// FLAG_GROUND (quadrotors): the body reached the ground plane of the reference's world — plane.urdf at GROUND_PLANE_Z = -0.05
// (base_aviary.py:107,219-220), collision cylinder of cf2x.urdf:31-36 half-height 0.0125 => contact at z <= -0.0375.  Bullet's
// contact response is NOT modelled (the shipped tasks end the episode out of bounds at z < 0 first); with
// `done_on_out_of_bound: False` or bounds that admit z < 0 the simulated body would fall through, so the step says so.

template <int SYS, typename T>
struct Env {
    using D = Dims<SYS>;
    T s[D::NS];        // raw simulator state
    T par[D::NP];      // inertial parameters
    int32_t step;      // ctrl_step_counter
    uint32_t episode;
    uint32_t gid;      // global env id (Philox counter word 0)
    // Initial-state draws of the NEXT episode, computed speculatively inside the integrator of the current control step
    // (PreDraw below): valid when pre_n > 0, consumed by reset().
    U4 pre[3];
    int pre_n;
};

// Speculative reset draws (specialised float builds, compact initial-state layout, no per-env parameter draws).
// The auto-reset path of a wave that holds a finished episode is its Philox4x32-10 blocks (~20 quarter-rate 32 x 32
// multiplies each); the integrator of the same control step is a DEPENDENT chain of packed fp32 instructions that issues
// every ~7-9 clocks with one wave per SIMD.  The Philox rounds depend on nothing the integrator computes, so they are
// spread over the unrolled substeps — round r between substeps — where they fill issue slots the chain leaves empty, and
// the reset path finds its words ready.  Same blocks, same counters (env, episode + 1, 0, tag), same ten rounds: results
// are bit-identical to rng_words() (tests compare the kernels with the oracle's Philox streams).
template <int NB>
struct PreDraw {
    U4 c[NB];
    uint32_t k0, k1;
    int round;                                              // rounds done on every block (0..10)
    __device__ __forceinline__ void begin(RngKey key, uint32_t gid, uint32_t next_episode) {
#pragma unroll
        for (int b = 0; b < NB; ++b) c[b] = U4{gid, next_episode, 0u, rng_tag(RNG_CH_RESET, RNG_GROUP_INIT, (uint32_t)b)};
        k0 = key.k0; k1 = key.k1; round = 0;
    }
    __device__ __forceinline__ void one_round() {           // one Philox round on every block (philox4x32_10's loop body)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            uint32_t hi0, lo0, hi1, lo1;
            mulhilo32(0xD2511F53u, c[b].x, hi0, lo0);
            mulhilo32(0xCD9E8D57u, c[b].z, hi1, lo1);
            U4 n;
            n.x = hi1 ^ c[b].y ^ k0; n.y = lo1; n.z = hi0 ^ c[b].w ^ k1; n.w = lo0;
            c[b] = n;
        }
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        ++round;
    }
    // After substep k of n (n a multiple of 10, the shipped configs: 20 and 50): one round every n / 10 substeps.  The test
    // `k % period == period - 1` is a compile-time constant in a loop unrolled by a multiple of the period, so no branch
    // is emitted; any other n leaves all ten rounds to finish().
    // `tie`: a live variable of the integrator chain.  It passes through the same (empty) asm statement as the round's words,
    // which orders substep k -> round -> substep k + 1 for every compiler pass: without it the pure integrator arithmetic is
    // free to move past the rounds (LLVM then emits all ten rounds in front of the loop, or sinks them into the reset branch).
    // U: the loop's unroll factor (the in-group index k % U is then a compile-time constant: no branch is emitted).  Within
    // every group of U substeps floor(10 U / n) rounds are issued, evenly spaced; what is left of the ten runs in finish().
    template <int U>
    __device__ __forceinline__ void tick(int k, int n, float& tie) {
        const int j = k % U, R = 10 * U / n;
        if (R > 0 && ((j + 1) * R) / U > (j * R) / U) {
            one_round();
#pragma unroll
            for (int b = 0; b < NB; ++b) { asm volatile("" : "+v"(tie) : "v"(c[b].x), "v"(c[b].y), "v"(c[b].z), "v"(c[b].w)); }
        }
    }
    __device__ __forceinline__ void finish() {
#pragma unroll
        for (int r = 0; r < 10; ++r)
            if (round < 10) one_round();
    }
};

// cos(2 pi u) for u in (0, 1): the Box-Muller angle.  Double: the library cosine of the product, as the oracle
// computes it.  Float: folded to a quarter period in revolutions (exact subtractions) and cos t = 1 - 2 sin^2(t / 2)
// with a degree-9 sine on [0, pi/4] — absolute error < 2e-7 (the noise it scales is a few 1e-3 of the state),
// a dozen VALU instead of the ~45 of cosf with its argument reduction.
__device__ __forceinline__ double cos_2pi(double u) { return cos(Const<double>::TWO_PI * u); }
__device__ __forceinline__ float cos_2pi(float u) {
    const float r = fabsf(u - 0.5f);                    // cos(2 pi u) = -cos(2 pi r), r in [0, 0.5)
    const float s = fminf(r, 0.5f - r);                 // fold: r > 1/4 -> cos(2 pi r) = -cos(2 pi (1/2 - r))
    const float x = 3.14159265358979f * s;              // half angle, [0, pi/4]
    const float x2 = x * x;
    const float sn = x * (1.0f + x2 * (-1.66666667e-1f + x2 * (8.33333333e-3f + x2 * (-1.98412698e-4f + x2 * 2.75573192e-6f))));
    const float c = 1.0f - 2.0f * sn * sn;              // cos(2 pi s) >= 0
    return (r > 0.25f) ? c : -c;
}

// ------------------------------------------------------------------ random helpers
// Reset-time draws: variable j of group g uses Philox block j/2 of item g, words 2*(j&1) and 2*(j&1)+1 — or, when no
// variable of the group is a normal draw, 21-bit field j%6 of block j/6 (scg_rng.h).
template <typename T>
__device__ __forceinline__ T rand_value(const HotRand<T>& r, const DevRand<T>& full, uint32_t w0, uint32_t w1) {
    if (r.kind == SCG_RAND_UNIFORM) return r.p0 + (r.p1 - r.p0) * u01<T>(w0);
    if (r.kind == SCG_RAND_NORMAL) {
        const T u1 = u01<T>(w0), u2 = u01<T>(w1);
        return r.p0 + r.p1 * (m_sqrt((T)-2 * m_log(u1)) * cos_2pi(u2));
    }
    if (r.kind == SCG_RAND_CHOICE) {          // choice lists live in the cold block
        const uint32_t k = int_below(w0, (uint32_t)full.n_choice);
        T v = full.choices[0];
#pragma unroll
        for (int c = 1; c < SCG_MAX_CHOICE; ++c) v = (k == (uint32_t)c) ? full.choices[c] : v;
        return v;
    }
    return (T)0;
}

// DisturbanceList.apply (disturbances.py:54-62) for channel CH; `vec` has DIM entries (registers: every loop
// over the list and over the dimensions has a compile-time bound).
// rng_step: Philox step index (pre-increment counter for action/dynamics; observation index for obs).
// Specialised build: list and dimension loops fully unrolled (kinds / masks / magnitudes are constants: each call
// site keeps only the code of the configured disturbances, `vec` stays in registers).  Generic build: the loops stay
// rolled — every kind's code once per call site; `vec` is then indexed dynamically (a few hundred bytes of scratch in
// the DIST kernels of the generic library, which is not the fast path for any config).
#ifdef SCG_SPEC
#define SCG_DIST_REF(P, ch, k) (KD.dist[ch][k])
#define SCG_DIST_UNROLL _Pragma("unroll")
#else
#define SCG_DIST_REF(P, ch, k) ((P).c.dist[ch][k])
#define SCG_DIST_UNROLL _Pragma("unroll 1")
#endif
template <typename T, int DIM, int CH>
__device__ __forceinline__ void apply_disturbances(const PV<T>& P, T* vec, RngKey key, uint32_t gid, uint32_t episode,
                                                   uint32_t rng_step, int32_t ctrl_step, int env_index) {
#ifdef SCG_SPEC
    constexpr CfgParams<T> KD = scg_make_spec_cfg<T>();
    constexpr int n = KD.n_dist[CH];
#else
    const int n = P.c.n_dist[CH];
#endif
    constexpr uint32_t rch = (uint32_t)(CH + 1);
SCG_DIST_UNROLL
    for (int k = 0; k < SCG_MAX_DISTURB; ++k) {
        if (k >= n) break;
        const HotDist<T>& d = SCG_DIST_REF(P, CH, k);
        if (d.kind == SCG_DIST_IMPULSE || d.kind == SCG_DIST_STEP) {
            int32_t off = d.step_offset;
            if (d.offset_slot >= 0)
                off = slot_in<int32_t>(make_rsrc(P.i.ws), P.i.dist_off, env_index).load((size_t)d.offset_slot * P.i.num_envs);
            T gain = (T)0;
            if (ctrl_step >= off) {
                if (d.kind == SCG_DIST_STEP) {
                    gain = (T)1;
                } else {
                    int32_t peak = (int32_t)((T)off + d.half_duration);           // int(offset + duration/2)
                    int32_t po = ctrl_step - peak; po = po < 0 ? -po : po;
                    gain = ((T)po < d.half_duration) ? m_pow(d.decay_rate, (T)po) : (T)0;
                }
            }
SCG_DIST_UNROLL
            for (int j = 0; j < DIM; ++j) vec[j] += d.a[j] * gain;
        } else if (d.kind == SCG_DIST_UNIFORM || d.kind == SCG_DIST_PERIODIC) {
            U4 w{0, 0, 0, 0};
            T tphase = (T)0;
            if (d.kind == SCG_DIST_PERIODIC) tphase = d.two_pi_freq * ((T)(ctrl_step * P.c.substeps) * P.c.pyb_dt);
SCG_DIST_UNROLL
            for (int j = 0; j < DIM; ++j) {
                if ((j & 3) == 0) w = rng_words(key, gid, episode, rng_step, rng_tag(rch, (uint32_t)k, (uint32_t)(j >> 2)));
                const T u = u01<T>(u4_get(w, j & 3));
                if (d.kind == SCG_DIST_UNIFORM) {
                    vec[j] += (d.a[j] + (d.b[j] - d.a[j]) * u) * d.mask[j];
                } else {
                    const T phase = -Const<T>::PI + Const<T>::TWO_PI * u;
                    vec[j] += d.a[j] * m_sin(tphase + phase);
                }
            }
        } else if (d.kind == SCG_DIST_WHITE) {
            U4 w{0, 0, 0, 0};
SCG_DIST_UNROLL
            for (int j = 0; j < DIM; ++j) {
                if ((j & 1) == 0) w = rng_words(key, gid, episode, rng_step, rng_tag(rch, (uint32_t)k, (uint32_t)(j >> 1)));
                const T u1 = u01<T>((j & 1) ? w.z : w.x), u2 = u01<T>((j & 1) ? w.w : w.y);
                const T z = m_sqrt((T)-2 * m_log(u1)) * cos_2pi(u2);
                vec[j] += d.a[j] * z * d.mask[j];
            }
        }
    }
}

// ------------------------------------------------------------------ quaternion helpers (PyBullet conventions)
template <typename T>
__device__ __forceinline__ void quat_to_mat(const T* q, T R[3][3]) {
    T x = q[0], y = q[1], z = q[2], w = q[3];
    T d = x * x + y * y + z * z + w * w;
    T s = (T)2 / d;
    T xs = x * s, ys = y * s, zs = z * s;
    T wx = w * xs, wy = w * ys, wz = w * zs;
    T xx = x * xs, xy = x * ys, xz = x * zs;
    T yy = y * ys, yz = y * zs, zz = z * zs;
    R[0][0] = (T)1 - (yy + zz); R[0][1] = xy - wz; R[0][2] = xz + wy;
    R[1][0] = xy + wz; R[1][1] = (T)1 - (xx + zz); R[1][2] = yz - wx;
    R[2][0] = xz - wy; R[2][1] = yz + wx; R[2][2] = (T)1 - (xx + yy);
}

// p.getEulerFromQuaternion incl. gimbal branches.
template <typename T>
__device__ __forceinline__ void quat_to_euler(const T* q, T* rpy) {
    T x = q[0], y = q[1], z = q[2], w = q[3];
    T sqx = x * x, sqy = y * y, sqz = z * z, squ = w * w;
    T sarg = (T)-2 * (x * z - w * y);
    if (sarg <= (T)-0.99999) {
        rpy[0] = (T)0; rpy[1] = -Const<T>::HALF_PI; rpy[2] = (T)2 * m_atan2(x, -y);
    } else if (sarg >= (T)0.99999) {
        rpy[0] = (T)0; rpy[1] = Const<T>::HALF_PI; rpy[2] = (T)2 * m_atan2(-x, y);
    } else {
        rpy[0] = fast_atan2((T)2 * (y * z + w * x), squ - sqx - sqy + sqz);
        rpy[1] = fast_asin(sarg);
        rpy[2] = fast_atan2((T)2 * (x * y + w * z), squ + sqx - sqy - sqz);
    }
}

// p.getQuaternionFromEuler.
template <typename T>
__device__ __forceinline__ void euler_to_quat(T r, T p, T y, T* q) {
    T sr, cr, sp, cp, sy, cy;
    reset_sincos((T)0.5 * r, &sr, &cr);
    reset_sincos((T)0.5 * p, &sp, &cp);
    reset_sincos((T)0.5 * y, &sy, &cy);
    q[0] = sr * cp * cy - cr * sp * sy;
    q[1] = cr * sp * cy + sr * cp * sy;
    q[2] = cr * cp * sy - sr * sp * cy;
    q[3] = cr * cp * cy + sr * sp * sy;
}

// The pitch PyBullet reports for a pure rotation by `th` about +y: asin(sin th) with the gimbal snap.
template <typename T>
__device__ __forceinline__ T planar_pitch(T th) {
    if (m_abs(th) < (T)1.5663) return th;       // asin(0.99999) = 1.56632...
    T sarg = m_sin(th);
    if (sarg <= (T)-0.99999) return -Const<T>::HALF_PI;
    if (sarg >= (T)0.99999) return Const<T>::HALF_PI;
    return m_asin(sarg);
}

// ------------------------------------------------------------------ the environment
// DIST: any passive disturbance or adversary configured (host-selected kernel variant).
// STAUX: cache policy of every store this code issues (its own workspace arrays and the slots it is handed), see Slot.
// WSAUX: cache policy of the stores to the handle's own WORKSPACE arrays (raw state, counters, per-env parameters, disturbance offsets,
// the stale out-of-bounds byte).  They are read back by the NEXT launch's wave of the same env group, which the dispatcher places on the
// same XCD: written back (0) they stay in that XCD's L2 instead of crossing to the memory side and back — which pays where a launch is
// several waves per SIMD and already bandwidth-bound (Quadrotor shards of 131 072 .. 524 288 envs: -4 .. -9 % per launch,
// profiles/r05_ab_s122_s123_workspace_write_back.txt) and costs where it is one latency chain (65 536 envs: +1.7 %).
template <int SYS, typename T, bool DIST, int STAUX = SCG_ST_AUX, int WSAUX = STAUX>
struct EnvOps {
    using D = Dims<SYS>;
    using E = Env<SYS, T>;
    template <typename V>
    __device__ static __forceinline__ Slot<V, WSAUX> ws_slot(__amdgpu_buffer_rsrc_t r, uint32_t soff, int lane_index) {
        return slot_in<V>(r, soff, lane_index).template with<WSAUX>();
    }
    // speculative reset draws inside the integrator (PreDraw): specialised float builds whose reset is exactly the compact
    // initial-state draw (-DSCG_NO_PREDRAW switches it off: A/B measurements).  CartPole only since round 4: its 50-substep loop is a
    // long dependent chain whose issue gaps the Philox rounds fill (5.81 us without, 5.50 with); the quadrotor loops got short
    // (recurrence integrator) and the launches issue-bound (write-through stores), and there the 40-80 quarter-rate multiplies
    // of the speculative rounds cost MORE than the divergent reset branch they save: Quadrotor2D 4.19 -> 4.02 us, Quadrotor3D
    // 8.42 -> 8.26 us without them (same box, alternating, tools/sessions/s98; round 3, write-back stores: 0 / -0.5 %).
    // Bit-identical either way: the speculative rounds produce rng_words()' words.
#if defined(SCG_SPEC) && !defined(SCG_NO_PREDRAW)
    static constexpr bool PRE = sizeof(T) == 4 && !DIST && SYS == SCG_CARTPOLE && scg_make_spec_cfg<T>().randomized_init != 0 &&
                                scg_make_spec_cfg<T>().init_compact != 0 && scg_make_spec_cfg<T>().per_env_params == 0 &&
                                scg_make_spec_cfg<T>().auto_reset != 0 && scg_make_spec_cfg<T>().substeps % 10 == 0;
#else
    static constexpr bool PRE = false;
#endif
    static constexpr int PRE_NB = (Dims<SYS>::NX + COMPACT_PER_BLOCK - 1) / COMPACT_PER_BLOCK;
#ifdef SCG_SPEC
    static constexpr int PRE_U2 = scg_make_spec_cfg<T>().substeps > 0 ? scg_make_spec_cfg<T>().substeps : 1;   // the 2-D loop is fully unrolled
#else
    static constexpr int PRE_U2 = 1;
#endif
    static constexpr bool IS_QUAD = (SYS != SCG_CARTPOLE);

    // Split load: the raw state / counters are requested before the LDS staging barrier (P = global block),
    // the inertial parameters afterwards (P = LDS copy).
    __device__ static __forceinline__ void load_state(const PV<T>& P, int i, E& e) {
        const size_t N = (size_t)P.i.num_envs;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(P.i.ws);
        const Slot<T> sp = slot_in<T>(ws, P.i.state_off, i);
#pragma unroll
        for (int k = 0; k < D::NS; ++k) e.s[k] = sp.load(k * N);
        e.step = slot_in<int32_t>(ws, P.i.step_off, i).load();
        e.episode = slot_in<uint32_t>(ws, P.i.episode_off, i).load();
        e.gid = (uint32_t)(P.i.env_id_offset + i);
        e.pre_n = 0;
    }
    __device__ static __forceinline__ void load_params(const PV<T>& P, int i, E& e) {
        const size_t N = (size_t)P.i.num_envs;
        if (P.c.per_env_params) {
#pragma unroll
            for (int k = 0; k < D::NP; ++k) e.par[k] = slot_in<T>(make_rsrc(P.i.ws), P.i.param_off, i).load(k * N);
        } else {
#pragma unroll
            for (int k = 0; k < D::NP; ++k) e.par[k] = P.c.base_param[k];
        }
    }

    __device__ static __forceinline__ void store_state(const PV<T>& P, int i, const E& e) {
        const size_t N = (size_t)P.i.num_envs;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(P.i.ws);
#pragma unroll
        for (int k = 0; k < D::NS; ++k) ws_slot<T>(ws, P.i.state_off, i).store(e.s[k], k * N);
    }
    __device__ static __forceinline__ void store(const PV<T>& P, int i, const E& e, bool params_dirty, bool with_state = true) {
        const size_t N = (size_t)P.i.num_envs;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(P.i.ws);
        if (with_state) store_state(P, i, e);
        if (P.c.per_env_params && params_dirty) {
#pragma unroll
            for (int k = 0; k < D::NP; ++k) ws_slot<T>(ws, P.i.param_off, i).store(e.par[k], k * N);
        }
        ws_slot<int32_t>(ws, P.i.step_off, i).store(e.step);
        ws_slot<uint32_t>(ws, P.i.episode_off, i).store(e.episode);
    }

    // env.state (the vector the reference exposes), from the raw simulator state.
    __device__ static __forceinline__ void state_vector(const E& e, T* st) {
        if constexpr (SYS == SCG_CARTPOLE || SYS == SCG_QUAD_1D) {
#pragma unroll
            for (int k = 0; k < D::NX; ++k) st[k] = e.s[k];
        } else if constexpr (SYS == SCG_QUAD_2D) {
            st[0] = e.s[0]; st[1] = e.s[1]; st[2] = e.s[2]; st[3] = e.s[3];
            st[4] = planar_pitch(e.s[4]); st[5] = e.s[5];
        } else {
            // quadrotor.py:794-802: positions/velocities interleaved, rpy, BODY rates R^T w.
            T R[3][3];
            quat_to_mat(&e.s[3], R);
            T rpy[3];
            quat_to_euler(&e.s[3], rpy);
            st[0] = e.s[0]; st[1] = e.s[7]; st[2] = e.s[1]; st[3] = e.s[8]; st[4] = e.s[2]; st[5] = e.s[9];
            st[6] = rpy[0]; st[7] = rpy[1]; st[8] = rpy[2];
            const T wx = e.s[10], wy = e.s[11], wz = e.s[12];
            st[9] = R[0][0] * wx + R[1][0] * wy + R[2][0] * wz;
            st[10] = R[0][1] * wx + R[1][1] * wy + R[2][1] * wz;
            st[11] = R[0][2] * wx + R[1][2] * wy + R[2][2] * wz;
        }
    }

    // Reset one env (Quadrotor.reset / CartPole.reset).  Increments the episode index, draws disturbance
    // offsets, inertial parameters and the initial state (each addressed by its own Philox counter).
    // st_out (optional): env.state of the fresh episode (what state_vector(e, .) would return).
    __device__ static __forceinline__ void reset(const PV<T>& P, int i, E& e, RngKey key, T* st_out = nullptr) {
        e.episode += 1u;
        e.step = 0;
        const int pre_n_in = e.pre_n;
        e.pre_n = 0;                                        // (the words belong to THIS reset only; callers that reset without a step see 0)

        if constexpr (DIST) {
            // disturbance offsets (ImpulseDisturbance.reset / StepDisturbance.reset), variable index 4*ch + k
#ifdef SCG_SPEC
            constexpr CfgParams<T> KD = scg_make_spec_cfg<T>();
#endif
SCG_DIST_UNROLL
            for (int ch = 0; ch < 3; ++ch) {
SCG_DIST_UNROLL
                for (int k = 0; k < SCG_MAX_DISTURB; ++k) {
#ifdef SCG_SPEC
                    if (k >= KD.n_dist[ch]) break;
#else
                    if (k >= P.c.n_dist[ch]) break;
#endif
                    const HotDist<T>& d = SCG_DIST_REF(P, ch, k);
                    if (d.offset_slot >= 0) {
                        const int j = 4 * ch + k;
                        U4 w = rng_words(key, e.gid, e.episode, 0u, rng_tag(RNG_CH_RESET, RNG_GROUP_DISTURB, (uint32_t)(j >> 1)));
                        ws_slot<int32_t>(make_rsrc(P.i.ws), P.i.dist_off, i).store(
                            (int32_t)int_below((j & 1) ? w.z : w.x, (uint32_t)d.max_step), (size_t)d.offset_slot * P.i.num_envs);
                    }
                }
            }
        }
        if (P.c.per_env_params && P.c.param_compact) {
            // one-word distributions only: six 21-bit variables per Philox block (see scg_rng.h)
            constexpr int CPB = COMPACT_PER_BLOCK;
#pragma unroll
            for (int b = 0; b < (D::NP + CPB - 1) / CPB; ++b) {
                U4 w = rng_words(key, e.gid, e.episode, 0u, rng_tag(RNG_CH_RESET, RNG_GROUP_PARAM, (uint32_t)b));
#pragma unroll
                for (int k = 0; k < CPB; ++k)
                    if (CPB * b + k < D::NP)
                        e.par[CPB * b + k] = P.c.base_param[CPB * b + k] + rand_value(P.c.param_rand[CPB * b + k], P.i.cold->param_rand[CPB * b + k], compact_word(w, k), 0u);
            }
        } else if (P.c.per_env_params) {
#pragma unroll
            for (int b = 0; b < (D::NP + 1) / 2; ++b) {
                U4 w = rng_words(key, e.gid, e.episode, 0u, rng_tag(RNG_CH_RESET, RNG_GROUP_PARAM, (uint32_t)b));
                e.par[2 * b] = P.c.base_param[2 * b] + rand_value(P.c.param_rand[2 * b], P.i.cold->param_rand[2 * b], w.x, w.y);
                if (2 * b + 1 < D::NP) e.par[2 * b + 1] = P.c.base_param[2 * b + 1] + rand_value(P.c.param_rand[2 * b + 1], P.i.cold->param_rand[2 * b + 1], w.z, w.w);
            }
        }
        T iv[D::NX];
#pragma unroll
        for (int k = 0; k < D::NX; ++k) iv[k] = P.c.init_state[k];
        if (P.c.randomized_init && P.c.init_compact) {
            constexpr int CPB = COMPACT_PER_BLOCK;
#pragma unroll
            for (int b = 0; b < (D::NX + CPB - 1) / CPB; ++b) {
                U4 w = b < pre_n_in ? e.pre[b < 3 ? b : 0]           // drawn inside the integrator of this control step (PreDraw)
                                   : rng_words(key, e.gid, e.episode, 0u, rng_tag(RNG_CH_RESET, RNG_GROUP_INIT, (uint32_t)b));
#pragma unroll
                for (int k = 0; k < CPB; ++k)
                    if (CPB * b + k < D::NX)
                        iv[CPB * b + k] += rand_value(P.c.init_rand[CPB * b + k], P.i.cold->init_rand[CPB * b + k], compact_word(w, k), 0u);
            }
        } else if (P.c.randomized_init) {
#pragma unroll
            for (int b = 0; b < D::NX / 2; ++b) {
                U4 w = rng_words(key, e.gid, e.episode, 0u, rng_tag(RNG_CH_RESET, RNG_GROUP_INIT, (uint32_t)b));
                iv[2 * b] += rand_value(P.c.init_rand[2 * b], P.i.cold->init_rand[2 * b], w.x, w.y);
                iv[2 * b + 1] += rand_value(P.c.init_rand[2 * b + 1], P.i.cold->init_rand[2 * b + 1], w.z, w.w);
            }
        }
        if constexpr (SYS == SCG_CARTPOLE || SYS == SCG_QUAD_2D) {
#pragma unroll
            for (int k = 0; k < D::NS; ++k) e.s[k] = iv[k];
        } else if constexpr (SYS == SCG_QUAD_1D) {
            // INIT_STATE_LABELS[ONE_D] = ['init_x', 'init_x_dot'] (quadrotor.py:210) while reset() reads
            // init_z / init_z_dot with default 0 (:373-374): the 1D drone always starts at z = 0, z_dot = 0
            // and the drawn init_x values only move x.  Replicated: the raw state is (z, z_dot) = 0.
            e.s[0] = (T)0; e.s[1] = (T)0;
            (void)iv;
        } else {
            e.s[0] = iv[0]; e.s[1] = iv[2]; e.s[2] = iv[4];         // pos
            euler_to_quat(iv[6], iv[7], iv[8], &e.s[3]);             // quat from (phi, theta, psi)
            e.s[7] = iv[1]; e.s[8] = iv[3]; e.s[9] = iv[5];         // vel
            e.s[10] = iv[9]; e.s[11] = iv[10]; e.s[12] = iv[11];    // p,q,r applied as WORLD rates (:379-384)
            if constexpr (sizeof(T) == 4) {
                // float: getEulerFromQuaternion(getQuaternionFromEuler(a)) == a to rounding while the angles are
                // principal values away from the gimbal snap, so the Euler extraction (two atan2 + asin on the
                // auto-reset path of every wave that holds a finished episode) is skipped; the double path keeps it.
                if (st_out && m_abs(iv[6]) < (T)3.1 && m_abs(iv[7]) < (T)1.5 && m_abs(iv[8]) < (T)3.1) {
                    T R[3][3];
                    quat_to_mat(&e.s[3], R);
#pragma unroll
                    for (int k = 0; k < 9; ++k) st_out[k] = iv[k];
                    st_out[9] = R[0][0] * iv[9] + R[1][0] * iv[10] + R[2][0] * iv[11];
                    st_out[10] = R[0][1] * iv[9] + R[1][1] * iv[10] + R[2][1] * iv[11];
                    st_out[11] = R[0][2] * iv[9] + R[1][2] * iv[10] + R[2][2] * iv[11];
                    return;
                }
            }
        }
        if (st_out) state_vector(e, st_out);
    }

    // Observation row: state (+ observation-channel noise) (+ angle wrap) (+ goal rows).
    //   next_index: first X_GOAL row appended (1 at reset, ctrl_step_counter + 2 after a step;
    //               benchmark_env.py:433-437, quadrotor.py:813-816)
    //   rng_step:   Philox step index of the observation (0 at reset, k after the k-th step)
    //   ctrl_step:  ctrl_step_counter seen by impulse/step disturbances (pre-increment value)
    // True when the whole observation is one register row of NX or 2*NX elements (every case except trajectory
    // tracking with a goal horizon > 1, whose rows are streamed from the X_GOAL table).
    __device__ static __forceinline__ bool obs_is_row(const PV<T>& P) {
        return !(P.c.cost == SCG_COST_RL_REWARD && P.c.task == SCG_TASK_TRAJ_TRACKING && P.c.obs_goal_horizon > 1);
    }
    // Builds that row in registers; returns its length (NX, or 2*NX with the goal row appended).
    __device__ static __forceinline__ int obs_row(const PV<T>& P, const GoalTab<T>& goal_tab, const T* st,
                                                  const E& e, RngKey key, int next_index, uint32_t rng_step,
                                                  int32_t ctrl_step, int env_index, const T* ext_pre, T* row) {
#pragma unroll
        for (int k = 0; k < D::NX; ++k) row[k] = st[k];
        if constexpr (DIST) {
            if (P.c.n_dist[SCG_CH_OBSERVATION] > 0)
                apply_disturbances<T, D::NX, SCG_CH_OBSERVATION>(P, row, key, e.gid, e.episode, rng_step, ctrl_step, env_index);
        }
        if constexpr (SYS == SCG_CARTPOLE) {
            if (P.c.obs_wrap_angle) row[2] = normalize_angle(row[2]);
        }
        if (!(P.c.obs_goal_horizon > 0 && P.c.cost == SCG_COST_RL_REWARD)) return D::NX;
        if (ext_pre) {
#pragma unroll
            for (int k = 0; k < D::NX; ++k) row[D::NX + k] = ext_pre[k];
        } else {
            int r = 0;
            if (P.c.task == SCG_TASK_TRAJ_TRACKING) { const int last = P.c.goal_rows - 1; r = next_index > last ? last : next_index; }
#pragma unroll
            for (int k = 0; k < D::NX; ++k) row[D::NX + k] = goal_tab[r * D::NX + k];
        }
        return 2 * D::NX;
    }
    // Row -> this env's slot of an [N][obs_dim] array (per-lane, strided by the row pitch).
    template <int SLOT_AUX>
    __device__ static __forceinline__ void store_obs_row(const PV<T>& P, const T* row, int n, Slot<T, SLOT_AUX> dst_in) {
        const Slot<T, STAUX> dst = dst_in.template with<STAUX>();
        if (n == P.c.nobs && n == 2 * D::NX) dst.template store_row<2 * D::NX>(row);
        else if (n == P.c.nobs && n == D::NX) dst.template store_row<D::NX>(row);
        else {
            for (int k = 0; k < 2 * D::NX; ++k)
                if (k < n) dst.store(row[k], k);
        }
    }

    template <int SLOT_AUX>
    __device__ static __forceinline__ void write_obs(const PV<T>& P, const GoalTab<T>& goal_tab, const T* st,
                                                     const E& e, RngKey key, int next_index, uint32_t rng_step,
                                                     int32_t ctrl_step, int env_index, Slot<T, SLOT_AUX> dst_in,
                                                     const T* ext_pre = nullptr) {
        const Slot<T, STAUX> dst = dst_in.template with<STAUX>();
        if (obs_is_row(P)) {
            T row[2 * D::NX];
            const int n = obs_row(P, goal_tab, st, e, key, next_index, rng_step, ctrl_step, env_index, ext_pre, row);
            store_obs_row(P, row, n, dst);
            return;
        }
        // trajectory tracking, horizon > 1: state row, then `horizon` rows of X_GOAL
        T o[D::NX];
#pragma unroll
        for (int k = 0; k < D::NX; ++k) o[k] = st[k];
        if constexpr (DIST) {
            if (P.c.n_dist[SCG_CH_OBSERVATION] > 0)
                apply_disturbances<T, D::NX, SCG_CH_OBSERVATION>(P, o, key, e.gid, e.episode, rng_step, ctrl_step, env_index);
        }
        if constexpr (SYS == SCG_CARTPOLE) {
            if (P.c.obs_wrap_angle) o[2] = normalize_angle(o[2]);
        }
        const int last = P.c.goal_rows - 1;
#pragma unroll
        for (int k = 0; k < D::NX; ++k) dst.store(o[k], k);
        for (int r = 0; r < P.c.obs_goal_horizon; ++r) {
            int row = next_index + r; row = row > last ? last : row;
            // gather the goal row into registers first: the table loads must not be interleaved with the stores
            T g[D::NX];
#pragma unroll
            for (int k = 0; k < D::NX; ++k) g[k] = goal_tab[row * D::NX + k];
#pragma unroll
            for (int k = 0; k < D::NX; ++k) dst.store(g[k], D::NX * (1 + r) + k);
        }
    }

    // Constraint rows (constraints.py:97-109); returns "any violated".  only_state: reset-time subset (written
    // densely at rows 0..n_state-1).  c_out: this env's column of the SoA [rows][N] output (row stride `stride`
    // elements: every store of a wave is one contiguous 256-byte segment), or absent.
    template <int SLOT_AUX>
    __device__ static __forceinline__ bool constraints(const PV<T>& P, const T* st, const T* act,
                                                       Slot<T, SLOT_AUX> c_in, size_t stride, bool only_state) {
        const Slot<T, STAUX> c_out = c_in.template with<STAUX>();
        bool viol = false;
        // (1) box rows: flat, unrolled so the row loads are all in flight together; the constrained
        //     variable is picked from registers with a select chain (no dependent memory round trips)
#ifdef SCG_SPEC
        // the table and its row counts as front-end constants: the loops below unroll whatever the pass order
        // (a trip count read back from the by-reference config object left reset_kernel with a dynamically
        // indexed stack copy of the whole block: 1.5 KB of scratch)
        constexpr CfgParams<T> K = scg_make_spec_cfg<T>();
        const int nb = only_state ? K.n_box_state_rows : K.n_box_rows;
#define SCG_BOXROW(r) K.box[r]
#else
        const int nb = only_state ? P.c.n_box_state_rows : P.c.n_box_rows;     // state slots come first
#define SCG_BOXROW(r) P.c.box[r]
#endif
#ifdef SCG_SPEC
        // every index is a compile-time constant here: the values are kept in registers and stored in one
        // straight-line run (one null check for the whole output, row offsets are scalar adds)
        T cv[SCG_MAX_CON_ROWS];
#endif
SCG_BOX_UNROLL
        for (int r = 0; r < nb; ++r) {
            const BoxRow<T> br = SCG_BOXROW(r);
            const int fl = br.packed >> 16;
            const int slot = (br.packed >> 24) & 0x1f;
            T val = (T)0;
#pragma unroll
            for (int k = 0; k < D::NX; ++k) val = (slot == k) ? st[k] : val;
#pragma unroll
            for (int k = 0; k < D::NU; ++k) val = (slot == SCG_MAX_STATE + k) ? act[k] : val;
            T c = (fl & 2) ? (m_abs(val) - br.b) : (((fl & 4) ? -val : val) - br.b);
            if (P.c.box_round > (T)0) c = m_rint(c * P.c.box_round) * P.c.box_inv_round;
            viol = viol || ((fl & 1) ? (c >= (T)0) : (c > (T)0));
#ifdef SCG_SPEC
            cv[r] = c;
#else
            // row index comes from the LDS table (not provably uniform): folded into the lane offset
            if (c_out) c_out.store_at(c, (uint32_t)((only_state ? ((br.packed >> 8) & 0xff) : (br.packed & 0xff)) * stride));
#endif
        }
#ifdef SCG_SPEC
        if (c_out) {
SCG_BOX_UNROLL
            for (int r = 0; r < nb; ++r) {
                const BoxRow<T> br = SCG_BOXROW(r);
                c_out.store(cv[r], (size_t)(only_state ? ((br.packed >> 8) & 0xff) : (br.packed & 0xff)) * stride);
            }
        }
#endif
        // (2) dense / quadratic rows
        if (P.c.n_generic_rows > 0) {
            int state_pos = 0;
            for (int r = 0; r < P.c.n_con_rows; ++r) {
                const DevRow<T>& row = P.i.cold->con[r];
                const int my_state_pos = state_pos;
                if (row.var == 0) ++state_pos;
                if (row.is_box) continue;
                if (only_state && row.var != 0) continue;
                T c = (T)0;
                if (row.kind == SCG_ROW_DENSE) {
                    if (row.var == 0) {
#pragma unroll
                        for (int k = 0; k < D::NX; ++k) c += row.coef[k] * st[k];
                    } else {
#pragma unroll
                        for (int k = 0; k < D::NU; ++k) c += row.coef[k] * act[k];
                    }
                } else {   // quadratic: v' P v over the state (or input) vector
                    const T* Pm = P.i.cold->quad_P[row.index];
                    if (row.var == 0) {
#pragma unroll
                        for (int a = 0; a < D::NX; ++a) {
                            T acc = (T)0;
#pragma unroll
                            for (int b = 0; b < D::NX; ++b) acc += Pm[a * D::NX + b] * st[b];
                            c += st[a] * acc;
                        }
                    } else {
#pragma unroll
                        for (int a = 0; a < D::NU; ++a) {
                            T acc = (T)0;
#pragma unroll
                            for (int b = 0; b < D::NU; ++b) acc += Pm[a * D::NU + b] * act[b];
                            c += act[a] * acc;
                        }
                    }
                }
                c -= row.b;
                if (row.round_scale > (T)0) c = m_rint(c * row.round_scale) * row.inv_round_scale;
                viol = viol || (row.strict ? (c >= (T)0) : (c > (T)0));
                if (c_out) c_out.store_at(c, (uint32_t)((only_state ? my_state_pos : r) * stride));
            }
        }
        return viol;
    }

    // ---- prior-model integrator (SCG_INT_RK4): the continuous-time dynamics the reference hands to CasADi
    // (cartpole.py:412-414; quadrotor.py:490, :506-509, :552-562), x in env.state order, u = clipped thrusts / force,
    // integrated with classical RK4 like rk_discrete (controllers/mpc/mpc_utils.py:42-64).
    __device__ static __forceinline__ void sym_f(const PV<T>& P, const E& e, const T* x, const T* u, T* dx) {
        const T g = P.c.gravity;
        if constexpr (SYS == SCG_CARTPOLE) {
            const T l = e.par[0], M = e.par[1], m = e.par[2];
            const T Mm = m + M, ml = m * l;
            T sn, cs;
            m_sincos(x[2], &sn, &cs);
            const T tmp = (u[0] + ml * x[3] * x[3] * sn) / Mm;
            const T thdd = (g * sn - cs * tmp) / (l * ((T)(4.0 / 3.0) - m * cs * cs / Mm));
            dx[0] = x[1]; dx[1] = tmp - ml * thdd * cs / Mm; dx[2] = x[3]; dx[3] = thdd;
        } else if constexpr (SYS == SCG_QUAD_1D) {
            dx[0] = x[1]; dx[1] = u[0] / e.par[0] - g;
        } else if constexpr (SYS == SCG_QUAD_2D) {
            const T m = e.par[0], iyy = e.par[2];
            T sn, cs;
            m_sincos(x[4], &sn, &cs);
            const T th = u[0] + u[1];
            dx[0] = x[1]; dx[1] = sn * th / m; dx[2] = x[3]; dx[3] = cs * th / m - g; dx[4] = x[5];
            dx[5] = P.c.arm * (u[1] - u[0]) / iyy;                       // arm = L / sqrt(2) in this mode
        } else {
            const T m = e.par[0], J0 = e.par[1], J1 = e.par[2], J2 = e.par[3];
            T sphi, cphi, sth, cth, spsi, cpsi;
            m_sincos(x[6], &sphi, &cphi); m_sincos(x[7], &sth, &cth); m_sincos(x[8], &spsi, &cpsi);
            const T thrust = (u[0] + u[1] + u[2] + u[3]) / m;
            // third column of Rz Ry Rx
            const T r02 = cpsi * sth * cphi + spsi * sphi, r12 = spsi * sth * cphi - cpsi * sphi, r22 = cth * cphi;
            const T pb = x[9], qb = x[10], rb = x[11];
            const T lsq = P.c.arm, gam = P.c.km / P.c.kf;
            const T mb0 = lsq * (u[0] + u[1] - u[2] - u[3]), mb1 = lsq * (-u[0] + u[1] + u[2] - u[3]);
            const T mb2 = gam * (-u[0] + u[1] - u[2] + u[3]);
            const T jw0 = J0 * pb, jw1 = J1 * qb, jw2 = J2 * rb;
            const T tth = sth / cth;
            dx[0] = x[1]; dx[1] = r02 * thrust; dx[2] = x[3]; dx[3] = r12 * thrust; dx[4] = x[5]; dx[5] = r22 * thrust - g;
            dx[6] = pb + sphi * tth * qb + cphi * tth * rb;
            dx[7] = cphi * qb - sphi * rb;
            dx[8] = (sphi * qb + cphi * rb) / cth;
            dx[9] = (mb0 - (qb * jw2 - rb * jw1)) / J0;
            dx[10] = (mb1 - (rb * jw0 - pb * jw2)) / J1;
            dx[11] = (mb2 - (pb * jw1 - qb * jw0)) / J2;
        }
    }

    // `substeps` RK4 steps of size pyb_dt (= control period / substeps) on env.state; raw state rebuilt afterwards.
    __device__ static __forceinline__ void rk4_advance(const PV<T>& P, E& e, const T* u) {
        T x[D::NX];
        state_vector_raw(e, x);
        const T h = P.c.pyb_dt;
        for (int s = 0; s < P.c.substeps; ++s) {
            T k1[D::NX], k2[D::NX], k3[D::NX], k4[D::NX], y[D::NX];
            sym_f(P, e, x, u, k1);
#pragma unroll
            for (int k = 0; k < D::NX; ++k) y[k] = x[k] + (T)0.5 * h * k1[k];
            sym_f(P, e, y, u, k2);
#pragma unroll
            for (int k = 0; k < D::NX; ++k) y[k] = x[k] + (T)0.5 * h * k2[k];
            sym_f(P, e, y, u, k3);
#pragma unroll
            for (int k = 0; k < D::NX; ++k) y[k] = x[k] + h * k3[k];
            sym_f(P, e, y, u, k4);
#pragma unroll
            for (int k = 0; k < D::NX; ++k) x[k] += h * (T)(1.0 / 6.0) * (k1[k] + (T)2 * k2[k] + (T)2 * k3[k] + k4[k]);
        }
        if constexpr (SYS != SCG_QUAD_3D) {
#pragma unroll
            for (int k = 0; k < D::NS; ++k) e.s[k] = x[k];
        } else {
            e.s[0] = x[0]; e.s[1] = x[2]; e.s[2] = x[4];
            e.s[7] = x[1]; e.s[8] = x[3]; e.s[9] = x[5];
            euler_to_quat(x[6], x[7], x[8], &e.s[3]);
            T R[3][3];
            quat_to_mat(&e.s[3], R);
            e.s[10] = R[0][0] * x[9] + R[0][1] * x[10] + R[0][2] * x[11];      // world rates = R * body rates
            e.s[11] = R[1][0] * x[9] + R[1][1] * x[10] + R[1][2] * x[11];
            e.s[12] = R[2][0] * x[9] + R[2][1] * x[10] + R[2][2] * x[11];
        }
    }
    // env.state with the UNWRAPPED planar pitch (what the prior model integrates); equals state_vector otherwise
    __device__ static __forceinline__ void state_vector_raw(const E& e, T* st) {
        if constexpr (SYS == SCG_QUAD_2D) {
#pragma unroll
            for (int k = 0; k < D::NX; ++k) st[k] = e.s[k];
        } else {
            state_vector(e, st);
        }
    }

    struct StepResult { T reward; T mse; bool done; uint8_t flags; };

    // One control step, no auto-reset.  `act_in` = raw controller action; `adv` = adversary action or null.
    // Leaves the post-step state in `e` (counter incremented) and the post-step env.state in `st`.
    // = advance() (action pre-processing, disturbances, the engine substeps: the raw state in `e` moves) followed by evaluate()
    // (env.state, reward, done, mse, constraint rows, time limit: pure functions of the advanced state and the noisy action).
    template <int SLOT_AUX>
    __device__ static __forceinline__ StepResult step(const PV<T>& P, const GoalTab<T>& goal_tab, E& e,
                                                      const T* act_in, const T* adv, RngKey key, int env_index,
                                                      T* st, T* noisy_out, Slot<T, SLOT_AUX> c_out, size_t c_stride,
                                                      const T* ref_pre = nullptr, const T* ext_pre = nullptr,
                                                      const T* ext_reset = nullptr) {
        T noisy[D::NU];
        advance(P, e, act_in, adv, key, env_index, noisy);
        if (noisy_out) {
#pragma unroll
            for (int j = 0; j < D::NU; ++j) noisy_out[j] = noisy[j];
        }
        return evaluate(P, goal_tab, e, noisy, env_index, st, c_out, c_stride, ref_pre, ext_pre, ext_reset);
    }

    // First half of step(): `noisy` = the action after normalisation, action disturbances and the adversary, BEFORE clipping.
    __device__ static __forceinline__ void advance(const PV<T>& P, E& e, const T* act_in, const T* adv, RngKey key, int env_index,
                                                   T* noisy) {
        const int32_t c0 = e.step;      // ctrl_step_counter before the increment
        // ---- _preprocess_control
        T clipped[D::NU];
#pragma unroll
        for (int j = 0; j < D::NU; ++j) {
            T a = act_in[j];
            if (P.c.normalized_action) {
                if constexpr (IS_QUAD) a = ((T)1 + P.c.act_scale * a) * P.c.hover_thrust;
                else a = P.c.act_scale * a;
            }
            noisy[j] = a;
        }
        // ---- dynamics disturbance, sampled once per control step (quadrotor.py:413-435, cartpole.py:540-551)
        T fd[D::DYN];
#pragma unroll
        for (int j = 0; j < D::DYN; ++j) fd[j] = (T)0;
        bool has_dyn = false;
        if constexpr (DIST) {
            if (P.c.n_dist[SCG_CH_ACTION] > 0)
                apply_disturbances<T, D::NU, SCG_CH_ACTION>(P, noisy, key, e.gid, e.episode, (uint32_t)c0, c0, env_index);
            if (P.c.adversary_channel == SCG_CH_ACTION && adv) {
#pragma unroll
                for (int j = 0; j < D::NU; ++j) noisy[j] += adv[j];
            }
            has_dyn = (P.c.n_dist[SCG_CH_DYNAMICS] > 0) || (P.c.adversary_channel == SCG_CH_DYNAMICS);
            if (P.c.n_dist[SCG_CH_DYNAMICS] > 0)
                apply_disturbances<T, D::DYN, SCG_CH_DYNAMICS>(P, fd, key, e.gid, e.episode, (uint32_t)c0, c0, env_index);
            if (P.c.adversary_channel == SCG_CH_DYNAMICS && adv) {
#pragma unroll
                for (int j = 0; j < D::DYN; ++j) fd[j] += adv[j];
            }
        }
#pragma unroll
        for (int j = 0; j < D::NU; ++j) clipped[j] = m_clamp(noisy[j], P.c.act_low[j], P.c.act_high[j]);
        // ---- physics
        if (P.c.integrator == SCG_INT_RK4) {
            rk4_advance(P, e, clipped);
        } else {
        const T h = P.c.pyb_dt;
        const T vmax = P.c.vmax;
        // Taylor rotations are exact to < 1 ulp below their bound.  Planar systems rotate by d = h*w with
        // |w| <= vmax (Bullet's clamp); the 3-D exponential map uses the half angle |w| h / 2, |w| <= sqrt(3) vmax.
        // (planar systems: up to 0.25 rad per substep — CartPole at 750 Hz turns by at most 100/750 = 0.133)
        const bool small_angle = (SYS == SCG_QUAD_3D) ? (h * vmax * (T)0.8660254 <= (T)0.125) : (h * vmax <= (T)0.25);
        if constexpr (SYS == SCG_CARTPOLE) {
            const T force = clipped[0];
            const T l = e.par[0], M = e.par[1], m = e.par[2];
            // Bullet recomputes the pole inertia from its collision box (see oracle/bullet.py::pole_inertia).
            const T two_l = (T)2 * l;
            const T ip = m * (P.c.pole_box_width * P.c.pole_box_width + two_l * two_l) * (T)(1.0 / 12.0);
            const T a11 = M + m, a22 = ip + m * l * l, ml = m * l;
            const T mgl = m * P.c.gravity * l;
            T x = e.s[0], xd = e.s[1], th = e.s[2], thd = e.s[3];
            T sn, cs;
            m_sincos(th, &sn, &cs);
            int k0 = 0;
            if constexpr (!DIST && sizeof(T) == 4) {
                // float, no disturbance: the substep on 2-vectors (see the Q2 integrator): (xdd, thdd), (xd, thd),
                // (x, th), (sin, cos) and the two Taylor polynomials are pairs for v_pk_fma_f32 / v_pk_mul_f32.
                if (small_angle) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 sc = {sn, cs}, vel = {xd, thd}, pos = {x, th};
                    const f2 diag = {a22, a11};
                    const f2 c1 = {(float)(1.0 / 120), (float)(-1.0 / 720)}, c0 = {(float)(-1.0 / 6), (float)(1.0 / 24)};
                    const f2 cone = {1.0f, -0.5f};
                    PreDraw<PRE_NB> pd;
                    if constexpr (PRE) pd.begin(key, e.gid, e.episode + 1u);
#ifdef SCG_SPEC
#pragma unroll 10
#endif
                    for (; k0 < P.c.substeps; ++k0) {
                        if constexpr (PRE) { float tie = sc.x; pd.template tick<10>(k0, P.c.substeps, tie); sc.x = tie; }
                        const float a12 = ml * sc.y;
                        const float b1 = force + ml * vel.y * vel.y * sc.x;
                        const float b2 = mgl * sc.x;
                        const float inv_det = m_div_by(1.0f, a11 * a22 - a12 * a12);
                        // (xdd, thdd) = ((a22 b1 - a12 b2), (a11 b2 - a12 b1)) / det
                        f2 acc = diag * (f2){b1, b2};
                        acc = __builtin_elementwise_fma((f2)(-a12), (f2){b2, b1}, acc) * (f2)inv_det;
                        vel = __builtin_elementwise_fma(acc, (f2)h, vel);
                        vel.x = m_clamp(vel.x, -vmax, vmax);
                        vel.y = m_clamp(vel.y, -vmax, vmax);
                        const f2 dpos = vel * (f2)h;
                        pos += dpos;
                        const float d = dpos.y, d2 = d * d;
                        f2 pq = __builtin_elementwise_fma((f2)d2, c1, c0);
                        pq = __builtin_elementwise_fma((f2)d2, pq, cone);
                        const float sd = d * pq.x;
                        const float cd = __builtin_fmaf(d2, pq.y, 1.0f);
                        const f2 rot = sc.yx * (f2){sd, -sd};
                        sc = __builtin_elementwise_fma(sc, (f2)cd, rot);
                    }
                    sn = sc.x; cs = sc.y; xd = vel.x; thd = vel.y; x = pos.x; th = pos.y;
                    if constexpr (PRE) {
                        pd.finish();
#pragma unroll
                        for (int b = 0; b < PRE_NB; ++b) e.pre[b] = pd.c[b];
                        e.pre_n = PRE_NB;
                    }
                }
            }
            for (int k = k0; k < P.c.substeps; ++k) {
                if (!small_angle) m_sincos(th, &sn, &cs);
                const T a12 = ml * cs;
                T b1 = force + ml * thd * thd * sn;
                T b2 = mgl * sn;
                if constexpr (DIST) {
                    if (has_dyn) { b1 += fd[0]; b2 += l * (fd[0] * cs - fd[1] * sn); }
                }
                const T inv_det = m_div_by((T)1, a11 * a22 - a12 * a12);      // float: v_rcp (1 ulp), double: exact
                const T xdd = (a22 * b1 - a12 * b2) * inv_det;
                const T thdd = (a11 * b2 - a12 * b1) * inv_det;
                xd = m_clamp(xd + h * xdd, -vmax, vmax);
                thd = m_clamp(thd + h * thdd, -vmax, vmax);
                x += h * xd;
                const T d = h * thd;
                th += d;
                if (small_angle) {
                    T sd, cd;
                    small_sincos(d, sd, cd);
                    const T ns = sn * cd + cs * sd;
                    cs = cs * cd - sn * sd;
                    sn = ns;
                }
            }
            e.s[0] = x; e.s[1] = xd; e.s[2] = th; e.s[3] = thd;
        } else {
            // cmd2pwm / pwm2rpm (quadrotor_utils.py:16-60) and per-motor forces (base_aviary.py:370-372)
            T pwm[4];
            constexpr int n_motor = 4 / D::NU;
#pragma unroll
            for (int j = 0; j < D::NU; ++j) {
                T thr = m_max(clipped[j], (T)0);
                pwm[j] = m_div_by(m_sqrt_fast(m_div_by(thr / (T)n_motor, P.c.kf)) - P.c.pwm2rpm_const, P.c.pwm2rpm_scale);
            }
            if constexpr (D::NU == 1) { pwm[1] = pwm[0]; pwm[2] = pwm[0]; pwm[3] = pwm[0]; }
            if constexpr (D::NU == 2) { pwm[2] = pwm[1]; pwm[3] = pwm[0]; }
            T f[4], tq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                T p = m_clamp(pwm[j], P.c.pwm_min, P.c.pwm_max);
                T rpm = P.c.pwm2rpm_scale * p + P.c.pwm2rpm_const;
                f[j] = rpm * rpm * P.c.kf;
                tq[j] = rpm * rpm * P.c.km;
            }
            const T thrust = f[0] + f[1] + f[2] + f[3];
            const T mass = e.par[0];
            const T inv_m = (T)1 / mass;
            const T g = P.c.gravity;
            const T arm = P.c.arm;
            if constexpr (SYS == SCG_QUAD_1D) {
                T z = e.s[0], vz = e.s[1];
                const T az = thrust / mass - g + (has_dyn ? fd[0] / mass : (T)0);
                for (int k = 0; k < P.c.substeps; ++k) {
                    vz = m_clamp(vz + h * az, -vmax, vmax);
                    z += h * vz;
                }
                e.s[0] = z; e.s[1] = vz;
            } else if constexpr (SYS == SCG_QUAD_2D) {
                // planar reduction of the free-body step: y, roll, yaw stay 0 (motors [T1,T2,T2,T1]/2)
                const T inv_iyy = (T)1 / e.par[2];
                const T tau_prop = arm * (-f[0] + f[1] + f[2] - f[3]);
                const T tm = thrust * inv_m;
                T x = e.s[0], vx = e.s[1], z = e.s[2], vz = e.s[3], th = e.s[4], w = e.s[5];
                const T x0 = x, z0 = z;       // applyExternalForce point cached at the start of the control step
                const T fx = has_dyn ? fd[0] : (T)0, fz = has_dyn ? fd[1] : (T)0;
                const T fxm = fx * inv_m, fzm = fz * inv_m - g;
                T sn, cs;
                m_sincos(th, &sn, &cs);
                int k0 = 0;
                if constexpr (!DIST && sizeof(T) == 4) {
                    // float, no disturbance: the substep written on 2-vectors — (sin, cos), (vx, vz), (x, z) and the
                    // two Taylor polynomials are pairs that CDNA's packed fp32 instructions (v_pk_fma_f32 /
                    // v_pk_mul_f32) process in one issue slot: 17 instead of 22 instructions per substep, and with
                    // one wave per SIMD the instruction count is the time.  Same operations per component.
                    //
                    // SCG_Q2_RECUR (default on): within one control step the motor forces are constant, so the pitch rate grows
                    // linearly, d_k = h w_0 + k h dw, and the rotation by d_{k+1} is the rotation by d_k composed with a CONSTANT
                    // rotation by eps = h dw.  (sin d_k, cos d_k) is carried as a second 2-vector updated by that constant rotation
                    // (2 packed instructions), (sin th, cos th) is rotated by it (2), velocity and position integrate in 3: 7 packed
                    // instructions per substep instead of 16-17 (two Taylor polynomials + three clamps per substep); th and w of the
                    // end of the control step in closed form.  Bullet's +-100 coordinate-velocity clamp cannot trigger when
                    // |w_0| + n |dw| < 100 and max(|vx|, |vz|) + n h (|T/m| + max |f/m|) < 100 — checked once per control step;
                    // lanes that fail the check take the per-substep loop below.  float32 one-step error against the oracle
                    // <= 2.6e-6 per state dimension (the per-substep loop: 6.3e-6; tolerance 2e-5), closed loop inside 1e-4
                    // (tests/test_gpu_env_parity.py, test_gpu_parity_scale.py).  With write-through stores (SCG_ST_AUX) the launch
                    // is issue-bound again: 4.39 -> 4.17 us per launch at 65 536 envs (tools/sessions/s82.sh, same box, alternating).
#ifndef SCG_Q2_RECUR
#define SCG_Q2_RECUR 1
#endif
#if SCG_Q2_RECUR
                    if (small_angle) {
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        const float nsub = (float)P.c.substeps;
                        const float dwk = h * (tau_prop * inv_iyy);
                        const float amax = fabsf(tm) + fmaxf(fabsf(fxm), fabsf(fzm));
                        const bool free_run = (fabsf(w) + nsub * fabsf(dwk) < vmax) && (fmaxf(fabsf(vx), fabsf(vz)) + nsub * h * amax < vmax);
                        if (free_run) {
                            f2 sc = {sn, cs}, v2 = {vx, vz}, p2 = {x, z};
                            const f2 fm = {fxm, fzm};
                            float se, ce, s1, c1;
                            small_sincos(h * dwk, se, ce);
                            small_sincos(h * (w + dwk), s1, c1);
                            f2 rk = {s1, c1};
                            const f2 eps_a = {se, -se};
                            PreDraw<PRE_NB> pd;
                            if constexpr (PRE) pd.begin(key, e.gid, e.episode + 1u);
#pragma unroll
                            for (; k0 < P.c.substeps; ++k0) {
                                if constexpr (PRE) { float tie = sc.x; pd.template tick<PRE_U2>(k0, P.c.substeps, tie); sc.x = tie; }
                                const f2 acc = __builtin_elementwise_fma(sc, (f2)tm, fm);
                                v2 = __builtin_elementwise_fma(acc, (f2)h, v2);
                                p2 = __builtin_elementwise_fma(v2, (f2)h, p2);
                                const f2 rot = sc.yx * (f2){rk.x, -rk.x};
                                sc = __builtin_elementwise_fma(sc, (f2)rk.y, rot);
                                const f2 rr = rk.yx * eps_a;
                                rk = __builtin_elementwise_fma(rk, (f2)ce, rr);
                            }
                            th = __builtin_fmaf(nsub * h, w, th) + (h * dwk) * (0.5f * nsub * (nsub + 1.0f));
                            w = __builtin_fmaf(nsub, dwk, w);
                            sn = sc.x; cs = sc.y; vx = v2.x; vz = v2.y; x = p2.x; z = p2.y;
                            if constexpr (PRE) {
                                pd.finish();
#pragma unroll
                                for (int b = 0; b < PRE_NB; ++b) e.pre[b] = pd.c[b];
                                e.pre_n = PRE_NB;
                            }
                        }
                    }
                    if (false) {
#else
                    if (small_angle) {
#endif
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        const float dwk = h * (tau_prop * inv_iyy);
                        f2 sc = {sn, cs}, v2 = {vx, vz}, p2 = {x, z};
                        const f2 fm = {fxm, fzm};
                        const f2 c1 = {(float)(1.0 / 120), (float)(-1.0 / 720)}, c0 = {(float)(-1.0 / 6), (float)(1.0 / 24)};
                        const f2 cone = {1.0f, -0.5f};
                        PreDraw<PRE_NB> pd;
                        if constexpr (PRE) pd.begin(key, e.gid, e.episode + 1u);
#ifdef SCG_SPEC
#pragma unroll          // constant trip count: straight-line code, no loop branches (a taken branch costs ~25 clocks)
#endif
                        for (; k0 < P.c.substeps; ++k0) {
                            if constexpr (PRE) { float tie = sc.x; pd.template tick<PRE_U2>(k0, P.c.substeps, tie); sc.x = tie; }
                            w = m_clamp(w + dwk, -vmax, vmax);
                            const float d = h * w, d2 = d * d;
                            f2 pq = __builtin_elementwise_fma((f2)d2, c1, c0);
                            pq = __builtin_elementwise_fma((f2)d2, pq, cone);
                            const float sd = d * pq.x;
                            const float cd = __builtin_fmaf(d2, pq.y, 1.0f);
                            const f2 acc = __builtin_elementwise_fma(sc, (f2)tm, fm);
                            v2 = __builtin_elementwise_fma(acc, (f2)h, v2);
                            v2.x = m_clamp(v2.x, -vmax, vmax);
                            v2.y = m_clamp(v2.y, -vmax, vmax);
                            p2 = __builtin_elementwise_fma(v2, (f2)h, p2);
                            th += d;
                            const f2 rot = sc.yx * (f2){sd, -sd};
                            sc = __builtin_elementwise_fma(sc, (f2)cd, rot);
                        }
                        sn = sc.x; cs = sc.y; vx = v2.x; vz = v2.y; x = p2.x; z = p2.y;
                        if constexpr (PRE) {
                            pd.finish();
#pragma unroll
                            for (int b = 0; b < PRE_NB; ++b) e.pre[b] = pd.c[b];
                            e.pre_n = PRE_NB;
                        }
                    }
                }
                for (int k = k0; k < P.c.substeps; ++k) {
                    if (!small_angle) m_sincos(th, &sn, &cs);
                    T tau = tau_prop;
                    if constexpr (DIST) {
                        if (has_dyn) tau += (z0 - z) * fx - (x0 - x) * fz;      // ((p0 - p) x F)_y, base_aviary.py:272
                    }
                    w = m_clamp(w + h * (tau * inv_iyy), -vmax, vmax);
                    vx = m_clamp(vx + h * (sn * tm + fxm), -vmax, vmax);
                    vz = m_clamp(vz + h * (cs * tm + fzm), -vmax, vmax);
                    x += h * vx;
                    z += h * vz;
                    const T d = h * w;
                    th += d;
                    if (small_angle) {
                        T sd, cd;
                        small_sincos(d, sd, cd);
                        const T ns = sn * cd + cs * sd;
                        cs = cs * cd - sn * sd;
                        sn = ns;
                    }
                }
                e.s[0] = x; e.s[1] = vx; e.s[2] = z; e.s[3] = vz; e.s[4] = th; e.s[5] = w;
            } else {
                const T J0 = e.par[1], J1 = e.par[2], J2 = e.par[3];
                const T iJ0 = (T)1 / J0, iJ1 = (T)1 / J1, iJ2 = (T)1 / J2;
                const T tb0 = arm * (f[0] + f[1] - f[2] - f[3]);
                const T tb1 = arm * (-f[0] + f[1] + f[2] - f[3]);
                const T tb2 = -tq[0] + tq[1] - tq[2] + tq[3];
                const T tm = thrust * inv_m;
                T p[3] = {e.s[0], e.s[1], e.s[2]};
                T q[4] = {e.s[3], e.s[4], e.s[5], e.s[6]};
                T v[3] = {e.s[7], e.s[8], e.s[9]};
                T w[3] = {e.s[10], e.s[11], e.s[12]};
                const T p0[3] = {p[0], p[1], p[2]};
                const T hh = (T)0.5 * h;
                int k0 = 0;
                if constexpr (sizeof(T) == 4) {
                    // float: the free-body substep on packed pairs, ~70 instructions instead of the ~170 the
                    // scalar form below compiles to (with one wave per SIMD the instruction count is the time).
                    //   * q = (A, B) = ((x, y), (z, w)); the rotation matrix as pairs (R00,R11) (R01,R10) (R02,R12)
                    //     (R20,R21) + R22, built from products of 2q and q (|q| = 1: re-normalised every substep);
                    //   * R^T w and R wd as three packed FMAs + three scalar ones each (op_sel swaps / broadcasts);
                    //   * Euler's equations with the inertia folded: wd = tb/J - ((J2-J1)/J0 wb1 wb2, ...), the
                    //     better-conditioned form of (tb - wb x J wb) / J when two moments are (nearly) equal;
                    //   * quaternion product dq (x) q as 2 x 4 packed FMAs.
                    if (small_angle) {
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        f2 A = {q[0], q[1]}, B = {q[2], q[3]};
                        f2 W = {w[0], w[1]}, V = {v[0], v[1]}, X = {p[0], p[1]};
                        float w2 = w[2], v2 = v[2], x2 = p[2];
                        const f2 TBI = {tb0 * iJ0, tb1 * iJ1};
                        const float tbi2 = tb2 * iJ2;
                        const f2 IJ = {iJ0, iJ1};
                        const f2 KK = {(J2 - J1) * iJ0, (J0 - J2) * iJ1};
                        const float k2 = (J1 - J0) * iJ2;
                        const float htm = h * tm;
                        float hc2 = -h * g;
                        f2 HF = {0.0f, 0.0f};
                        f2 FD2 = {0.0f, 0.0f}, FDX = {0.0f, 0.0f};
                        const f2 X0 = X;
                        const float x20 = x2;
                        bool dyn = false;
                        if constexpr (DIST) {
                            if (has_dyn) {
                                dyn = true;
                                HF = (f2){h * fd[0] * inv_m, h * fd[1] * inv_m};
                                hc2 += h * fd[2] * inv_m;
                                FD2 = (f2){fd[2], -fd[2]};
                                FDX = (f2){-fd[1], fd[0]};
                            }
                        }
                        const f2 c1 = {(float)(1.0 / 120), (float)(-1.0 / 720)}, c0 = {(float)(-1.0 / 6), (float)(1.0 / 24)};
                        const f2 cone = {1.0f, -0.5f};
                        const float hh2 = hh * hh;
                        PreDraw<PRE_NB> pd;
                        if constexpr (PRE) pd.begin(key, e.gid, e.episode + 1u);
#ifdef SCG_SPEC
#pragma unroll SCG_Q3_UNROLL
#endif
                        for (; k0 < P.c.substeps; ++k0) {
                            if constexpr (PRE) { float tie = A.x; pd.template tick<SCG_Q3_UNROLL>(k0, P.c.substeps, tie); A.x = tie; }
                            const f2 A2 = A + A, B2 = B + B;
                            const f2 P1 = A2 * A;                       // (2xx, 2yy)
                            const f2 P2 = A2 * B.xx;                    // (2xz, 2yz)
                            const f2 P3 = A2 * B.yy;                    // (2xw, 2yw)
                            const f2 XY = A2 * A.yx;                    // (2xy, 2xy)
                            const f2 ZZW = B2.xx * B;                   // (2zz, 2zw)
                            const f2 P3s = {P3.y, -P3.x};
                            const f2 Rc2 = P2 + P3s;                    // (R02, R12)
                            const f2 Rr2 = P2 - P3s;                    // (R20, R21)
                            const f2 Rod = XY + (f2){-ZZW.y, ZZW.y};    // (R01, R10)
                            const f2 Rd = (f2){1.0f, 1.0f} - (P1.yx + ZZW.xx);   // (R00, R11)
                            const float R22 = 1.0f - (P1.x + P1.y);
                            // body rates wb = R^T w
                            f2 WB = Rd * W;
                            WB = __builtin_elementwise_fma(Rod.yx, W.yx, WB);
                            WB = __builtin_elementwise_fma(Rr2, (f2)w2, WB);
                            float wb2 = Rc2.x * W.x;
                            wb2 = __builtin_fmaf(Rc2.y, W.y, wb2);
                            wb2 = __builtin_fmaf(R22, w2, wb2);
                            // torque / J in the body frame (+ the off-centre disturbance force, base_aviary.py:272)
                            f2 TQ = TBI;
                            float tq2 = tbi2;
                            if constexpr (DIST) {
                                if (dyn) {
                                    const f2 Rw = X0 - X;               // p0 - p
                                    const float r2 = x20 - x2;
                                    f2 TW = Rw.yx * FD2;                // (r1 fd2, -r0 fd2)
                                    TW = __builtin_elementwise_fma((f2)r2, FDX, TW);     // + (-r2 fd1, r2 fd0)
                                    const float tw2 = Rw.x * fd[1] - Rw.y * fd[0];
                                    f2 TB = Rd * TW;
                                    TB = __builtin_elementwise_fma(Rod.yx, TW.yx, TB);
                                    TB = __builtin_elementwise_fma(Rr2, (f2)tw2, TB);
                                    float tb2w = Rc2.x * TW.x;
                                    tb2w = __builtin_fmaf(Rc2.y, TW.y, tb2w);
                                    tb2w = __builtin_fmaf(R22, tw2, tb2w);
                                    TQ = __builtin_elementwise_fma(TB, IJ, TQ);
                                    tq2 = __builtin_fmaf(tb2w, iJ2, tq2);
                                }
                            }
                            const f2 M = WB.yx * (f2)wb2;               // (wb1 wb2, wb0 wb2)
                            const f2 WD = __builtin_elementwise_fma(-KK, M, TQ);
                            const float wd2 = __builtin_fmaf(-(k2 * WB.x), WB.y, tq2);
                            // w += h R wd, Bullet's clamp
                            f2 DW = Rd * WD;
                            DW = __builtin_elementwise_fma(Rod, WD.yx, DW);
                            DW = __builtin_elementwise_fma(Rc2, (f2)wd2, DW);
                            float dw2 = Rr2.x * WD.x;
                            dw2 = __builtin_fmaf(Rr2.y, WD.y, dw2);
                            dw2 = __builtin_fmaf(R22, wd2, dw2);
                            W = __builtin_elementwise_fma(DW, (f2)h, W);
                            w2 = __builtin_fmaf(dw2, h, w2);
                            W.x = m_clamp(W.x, -vmax, vmax);
                            W.y = m_clamp(W.y, -vmax, vmax);
                            w2 = m_clamp(w2, -vmax, vmax);
                            // v += h (R e3 thrust/m - g e3 + F/m), p += h v
                            V = __builtin_elementwise_fma(Rc2, (f2)htm, V);
                            if constexpr (DIST) {
                                if (dyn) V += HF;
                            }
                            v2 = __builtin_fmaf(R22, htm, v2 + hc2);
                            V.x = m_clamp(V.x, -vmax, vmax);
                            V.y = m_clamp(V.y, -vmax, vmax);
                            v2 = m_clamp(v2, -vmax, vmax);
                            X = __builtin_elementwise_fma(V, (f2)h, X);
                            x2 = __builtin_fmaf(v2, h, x2);
                            // q <- normalize(exp(h w / 2) (x) q), Taylor sinc / cos of the half angle
                            float n2 = W.x * W.x;
                            n2 = __builtin_fmaf(W.y, W.y, n2);
                            n2 = __builtin_fmaf(w2, w2, n2);
                            const float a2 = hh2 * n2;
                            f2 pq = __builtin_elementwise_fma((f2)a2, c1, c0);
                            pq = __builtin_elementwise_fma((f2)a2, pq, cone);
                            const float cw = __builtin_fmaf(a2, pq.y, 1.0f);
                            const float kk = hh * pq.x;
                            const f2 Dxy = W * (f2)kk;
                            const float dz = w2 * kk;
                            f2 An = A * (f2)cw;
                            An = __builtin_elementwise_fma(Dxy.xx, (f2){B.y, -B.x}, An);
                            An = __builtin_elementwise_fma(Dxy.yy, B, An);
                            An = __builtin_elementwise_fma((f2)dz, (f2){-A.y, A.x}, An);
                            f2 Bn = B * (f2)cw;
                            Bn = __builtin_elementwise_fma(Dxy.xx, (f2){A.y, -A.x}, Bn);
                            Bn = __builtin_elementwise_fma(-Dxy.yy, A, Bn);
                            Bn = __builtin_elementwise_fma((f2)dz, (f2){B.y, -B.x}, Bn);
                            f2 S = An * An;
                            S = __builtin_elementwise_fma(Bn, Bn, S);
                            const float inv = __builtin_amdgcn_rsqf(S.x + S.y);
                            A = An * (f2)inv;
                            B = Bn * (f2)inv;
                        }
                        if constexpr (PRE) {
                            pd.finish();
#pragma unroll
                            for (int b = 0; b < PRE_NB; ++b) e.pre[b] = pd.c[b];
                            e.pre_n = PRE_NB;
                        }
                        q[0] = A.x; q[1] = A.y; q[2] = B.x; q[3] = B.y;
                        w[0] = W.x; w[1] = W.y; w[2] = w2;
                        v[0] = V.x; v[1] = V.y; v[2] = v2;
                        p[0] = X.x; p[1] = X.y; p[2] = x2;
                    }
                }
                for (int k = k0; k < P.c.substeps; ++k) {
                    T R[3][3];
                    quat_to_mat(q, R);
                    T t0 = tb0, t1 = tb1, t2 = tb2;
                    T a0 = R[0][2] * tm, a1 = R[1][2] * tm, a2 = R[2][2] * tm - g;
                    if constexpr (DIST) {
                        if (has_dyn) {
                            const T r0 = p0[0] - p[0], r1 = p0[1] - p[1], r2 = p0[2] - p[2];
                            const T tw0 = r1 * fd[2] - r2 * fd[1], tw1 = r2 * fd[0] - r0 * fd[2], tw2 = r0 * fd[1] - r1 * fd[0];
                            t0 += R[0][0] * tw0 + R[1][0] * tw1 + R[2][0] * tw2;
                            t1 += R[0][1] * tw0 + R[1][1] * tw1 + R[2][1] * tw2;
                            t2 += R[0][2] * tw0 + R[1][2] * tw1 + R[2][2] * tw2;
                            a0 += fd[0] * inv_m; a1 += fd[1] * inv_m; a2 += fd[2] * inv_m;
                        }
                    }
                    const T wb0 = R[0][0] * w[0] + R[1][0] * w[1] + R[2][0] * w[2];
                    const T wb1 = R[0][1] * w[0] + R[1][1] * w[1] + R[2][1] * w[2];
                    const T wb2 = R[0][2] * w[0] + R[1][2] * w[1] + R[2][2] * w[2];
                    const T jw0 = J0 * wb0, jw1 = J1 * wb1, jw2 = J2 * wb2;
                    const T wd0 = (t0 - (wb1 * jw2 - wb2 * jw1)) * iJ0;
                    const T wd1 = (t1 - (wb2 * jw0 - wb0 * jw2)) * iJ1;
                    const T wd2 = (t2 - (wb0 * jw1 - wb1 * jw0)) * iJ2;
                    w[0] = m_clamp(w[0] + h * (R[0][0] * wd0 + R[0][1] * wd1 + R[0][2] * wd2), -vmax, vmax);
                    w[1] = m_clamp(w[1] + h * (R[1][0] * wd0 + R[1][1] * wd1 + R[1][2] * wd2), -vmax, vmax);
                    w[2] = m_clamp(w[2] + h * (R[2][0] * wd0 + R[2][1] * wd1 + R[2][2] * wd2), -vmax, vmax);
                    v[0] = m_clamp(v[0] + h * a0, -vmax, vmax);
                    v[1] = m_clamp(v[1] + h * a1, -vmax, vmax);
                    v[2] = m_clamp(v[2] + h * a2, -vmax, vmax);
                    p[0] += h * v[0]; p[1] += h * v[1]; p[2] += h * v[2];
                    // exponential-map orientation update (btMultiBody::stepPositionsMultiDof):
                    // dq = (w * sin(|w| h/2)/|w|, cos(|w| h/2)), q <- normalize(dq (x) q)
                    const T w2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
                    T kk, dw;
                    if (small_angle) {
                        T sinc;
                        small_sinc_cos(hh * hh * w2, sinc, dw);
                        kk = hh * sinc;
                    } else {
                        T ang = m_sqrt(w2);
                        if (ang * h > (T)0.78539816339744830962) ang = (T)0.78539816339744830962 / h;
                        if (ang < (T)0.001) kk = (T)0.5 * h - (h * h * h) * (T)0.020833333333 * ang * ang;
                        else kk = m_sin((T)0.5 * ang * h) / ang;
                        dw = m_cos((T)0.5 * ang * h);
                    }
                    const T dx = w[0] * kk, dy = w[1] * kk, dz = w[2] * kk;
                    const T qx = dw * q[0] + dx * q[3] + dy * q[2] - dz * q[1];
                    const T qy = dw * q[1] - dx * q[2] + dy * q[3] + dz * q[0];
                    const T qz = dw * q[2] + dx * q[1] - dy * q[0] + dz * q[3];
                    const T qw = dw * q[3] - dx * q[0] - dy * q[1] - dz * q[2];
                    const T inv = m_rsqrt(qx * qx + qy * qy + qz * qz + qw * qw);
                    q[0] = qx * inv; q[1] = qy * inv; q[2] = qz * inv; q[3] = qw * inv;
                }
                e.s[0] = p[0]; e.s[1] = p[1]; e.s[2] = p[2];
                e.s[3] = q[0]; e.s[4] = q[1]; e.s[5] = q[2]; e.s[6] = q[3];
                e.s[7] = v[0]; e.s[8] = v[1]; e.s[9] = v[2];
                e.s[10] = w[0]; e.s[11] = w[1]; e.s[12] = w[2];
            }
        }
        }   // integrator
    }

    // Second half of step(): everything the reference derives from the advanced state.  `e.step` is still the PRE-increment counter
    // on entry and is incremented here.
    template <int SLOT_AUX>
    __device__ static __forceinline__ StepResult evaluate(const PV<T>& P, const GoalTab<T>& goal_tab, E& e, const T* noisy, int env_index,
                                                          T* st, Slot<T, SLOT_AUX> c_out, size_t c_stride,
                                                          const T* ref_pre = nullptr, const T* ext_pre = nullptr,
                                                          const T* ext_reset = nullptr) {
        const int32_t c0 = e.step;      // ctrl_step_counter before the increment
        T clipped[D::NU];
#pragma unroll
        for (int j = 0; j < D::NU; ++j) clipped[j] = m_clamp(noisy[j], P.c.act_low[j], P.c.act_high[j]);
        state_vector(e, st);
        // rows of X_GOAL requested before the integrator: retire them here, ahead of the first store
        if (ref_pre) {
#pragma unroll
            for (int k = 0; k < D::NX; ++k) vreg_fence(ref_pre[k]);
        }
        if (ext_pre) {
#pragma unroll
            for (int k = 0; k < D::NX; ++k) vreg_fence(ext_pre[k]);
        }
        if (ext_reset) {
#pragma unroll
            for (int k = 0; k < D::NX; ++k) vreg_fence(ext_reset[k]);
        }
#ifdef SCG_EXP_TIMELINE
#pragma unroll
        for (int k = 0; k < D::NX; ++k) vreg_fence(st[k]);
#endif
        SCG_TL(3);

        // ---- reference row for reward / mse (tracking: X_GOAL[min(c+1, L-1)])
        T ref[D::NX];
        const bool tracking = P.c.task == SCG_TASK_TRAJ_TRACKING;
        if (ref_pre) {
#pragma unroll
            for (int k = 0; k < D::NX; ++k) ref[k] = ref_pre[k];
        } else {
            int row = 0;
            if (tracking) { row = c0 + 1; const int last = P.c.goal_rows - 1; row = row > last ? last : row; }
#pragma unroll
            for (int k = 0; k < D::NX; ++k) ref[k] = goal_tab[row * D::NX + k];
        }
        // ---- _get_reward
        T rew;
        if (P.c.cost == SCG_COST_RL_REWARD) {
            T dist = (T)0;
#pragma unroll
            for (int k = 0; k < D::NX; ++k) {
                T sv = st[k];
                if constexpr (SYS == SCG_CARTPOLE) { if (k == 2) sv = normalize_angle(sv); }   // cartpole.py:619-620
                const T err = sv - ref[k];
                dist += P.c.rew_state_weight[k] * err * err;
            }
#pragma unroll
            for (int j = 0; j < D::NU; ++j) {
                const T ae = noisy[j] - P.c.u_goal[j];      // unclipped noisy action (quadrotor.py:828); cartpole U_GOAL = 0
                dist += P.c.rew_act_weight[j] * ae * ae;
            }
            rew = P.c.rew_exponential ? m_exp(-dist) : -dist;
        } else {
            // quadratic cost with diagonal Q, R (lqr_utils.py:77-99) and the CLIPPED action.
            T xr[D::NX];
            if (tracking) {
                // quadrotor: X_GOAL[c+1] (quadrotor.py:858); cartpole: X_GOAL[c] (cartpole.py:648); no clamping upstream
                int row = (SYS == SCG_CARTPOLE) ? c0 : c0 + 1;
                const int last = P.c.goal_rows - 1; row = row > last ? last : row;
#pragma unroll
                for (int k = 0; k < D::NX; ++k) xr[k] = goal_tab[row * D::NX + k];
            } else {
#pragma unroll
                for (int k = 0; k < D::NX; ++k) xr[k] = ref[k];
            }
            T cst = (T)0;
#pragma unroll
            for (int k = 0; k < D::NX; ++k) { const T err = st[k] - xr[k]; cst += (T)0.5 * P.c.q_diag[k] * err * err; }
#pragma unroll
            for (int j = 0; j < D::NU; ++j) { const T du = clipped[j] - P.c.u_goal[j]; cst += (T)0.5 * P.c.r_diag[j] * du * du; }
            rew = -cst;
        }
        // ---- _get_done
        bool done = false;
        uint8_t flags = 0;
        bool goal = false;
        if (!tracking) {
            T n2 = (T)0;
#pragma unroll
            for (int k = 0; k < D::NX; ++k) { const T err = st[k] - ref[k]; n2 += err * err; }
            goal = m_sqrt(n2) < P.c.goal_tolerance;
            done = goal;
            if (goal && P.c.info_goal_reached) flags |= FLAG_GOAL;
        }
        if (P.c.done_on_oob) {
            bool oob = false;
            if constexpr (SYS == SCG_CARTPOLE) {
                oob = st[0] < -P.c.x_threshold || st[0] > P.c.x_threshold || st[2] < -P.c.theta_threshold || st[2] > P.c.theta_threshold;
            } else {
                // positions + angles only (quadrotor.py:878-888)
#pragma unroll
                for (int k = 0; k < D::NX; ++k) {
                    const bool masked = (SYS == SCG_QUAD_3D) ? ((k < 6 && (k & 1) == 0) || (k >= 6 && k < 9)) : ((k & 1) == 0);
                    if (masked) oob = oob || st[k] < P.c.state_low[k] || st[k] > P.c.state_high[k];
                }
            }
            if (!tracking) {
                // stale `self.out_of_bounds` on goal_reached steps (see oracle/envs.py::_stale_oob)
                const auto attr = ws_slot<uint8_t>(make_rsrc(P.i.ws), P.i.oob_off, env_index);
                const bool prev = attr.load() != 0;
                oob = goal ? prev : oob;
                attr.store(oob ? 1 : 0);
            }
            if (oob) flags |= FLAG_OOB;
            done = done || (oob && !goal);
        }
        if constexpr (SYS != SCG_CARTPOLE) {
            constexpr int ZI = SYS == SCG_QUAD_1D ? 0 : (SYS == SCG_QUAD_2D ? 2 : 4);
            if (st[ZI] <= (T)-0.0375) flags |= FLAG_GROUND;
        }
        // ---- _get_info: mse
        T mse = (T)0;
#pragma unroll
        for (int k = 0; k < D::NX; ++k) {
            T sv = st[k];
            if (tracking) {
                if constexpr (SYS == SCG_CARTPOLE) { if (k == 2) sv = normalize_angle(sv); }
                if constexpr (SYS == SCG_QUAD_2D) { if (k == 4) sv = normalize_angle(sv); }
                if constexpr (SYS == SCG_QUAD_3D) { if (k >= 6 && k < 9) sv = normalize_angle(sv); }
            }
            const T err = (sv - ref[k]) * P.c.mse_weight[k];
            mse += err * err;
        }
        // ---- after_step
#ifdef SCG_EXP_TIMELINE
        vreg_fence(rew); vreg_fence(mse);
#endif
        SCG_TL(4);
        e.step = c0 + 1;
        bool viol = false;
        if (P.c.n_con_rows > 0) {
            viol = constraints(P, st, noisy, c_out, c_stride, false);
            if (viol) {
                flags |= FLAG_VIOLATION;
                if (P.c.done_on_violation) {
                    done = true;
                    if (P.c.cost == SCG_COST_RL_REWARD && P.c.use_penalty) rew = (T)0;
                }
            }
            if (P.c.cost == SCG_COST_RL_REWARD && P.c.use_penalty && viol) {
                if (P.c.rew_exponential) rew = m_exp(m_log(rew) - P.c.constraint_penalty);
                else rew -= P.c.constraint_penalty;
            }
        }
        SCG_TL(5);
        if (e.step >= P.c.ctrl_steps) {
            if (!done) flags |= FLAG_TRUNCATED;
            done = true;
        }
        StepResult r;
        r.reward = rew; r.mse = mse; r.done = done; r.flags = flags;
        return r;
    }
};

}  // namespace scg

// scg_gae_kernels.h — GAE / discounted-return kernels (controllers/ppo/ppo_utils.py:374-400).
#pragma once
#include <hip/hip_runtime.h>

namespace scg {
constexpr int GAE_BLOCK = 256;

// ---- GAE / returns (controllers/ppo/ppo_utils.py:374-400), buffers [T][N] -------------------------
// (a) one thread per env walking T backwards: every load/store is coalesced across the wave.
template <typename T>
__global__ __launch_bounds__(GAE_BLOCK) void gae_env_kernel(T* __restrict__ rew, const T* __restrict__ v, const T* __restrict__ mask,
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

// (a') the collector's shape (many envs, T <= 16 CH): the walk above is one dependent load -> compute -> store round per time step
// with one wave per SIMD at 65 536 envs (23 us for 67 MB, 0.36 of the HBM roofline).  Here a workgroup of 64 envs x S
// segments gives every thread CH consecutive time steps: it requests all of them at once (ONE memory round trip, S times the
// waves), reduces them to the affine maps  ret_in -> ret_out,  adv_in -> adv_out  of its segment ((a, b) o (a', b') =
// (a a', b + a b'), the same algebra as gae_wave_kernel), exchanges the maps through LDS, composes the maps of the later
// segments into its carry-in and walks its own steps from registers.
template <typename T, int CH>
__global__ __launch_bounds__(1024) void gae_seg_kernel(T* __restrict__ rew, const T* __restrict__ v, const T* __restrict__ mask,
                                                       const T* __restrict__ term_v, const T* __restrict__ last_v,
                                                       T* __restrict__ ret, T* __restrict__ adv, int Tn, int N, T gamma, T lam,
                                                       int use_gae) {
    extern __shared__ __align__(16) unsigned char gae_lds[];
    T* const maps = reinterpret_cast<T*>(gae_lds);                      // [S][4][64]: a_ret, b_ret, a_adv, b_adv
    const int lane = threadIdx.x, seg = threadIdx.y, S = blockDim.y;
    const int n0 = blockIdx.x * 64 + lane;
    const bool env_live = n0 < N;
    const int n = env_live ? n0 : N - 1;
    const int t0 = seg * CH;
    T r[CH], m[CH], vt[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int t = t0 + k;
        const bool ok = t < Tn;
        const size_t idx = (size_t)(ok ? t : Tn - 1) * N + n;
        r[k] = rew[idx]; m[k] = mask[idx]; vt[k] = v[idx];
        if (term_v) r[k] += gamma * term_v[idx];
        if (!ok) { r[k] = (T)0; m[k] = (T)1; }                           // identity step
    }
    const int t_last = (t0 + CH < Tn ? t0 + CH : Tn) - 1;                // latest time of this segment (t0 <= Tn - 1 by the launch)
    const T v_after = t_last + 1 < Tn ? v[(size_t)(t_last + 1) * N + n] : last_v[n];
    // segment maps, composed in the order the recursion applies them (latest time first)
    T ar = (T)1, br = (T)0, aa = (T)1, ba = (T)0;
#pragma unroll
    for (int k = CH - 1; k >= 0; --k) {
        if (t0 + k < Tn) {
            const T gm = gamma * m[k];
            br = r[k] + gm * br; ar = gm * ar;
            const T vn = (t0 + k == t_last) ? v_after : vt[k + 1 < CH ? k + 1 : k];
            const T delta = r[k] + gm * vn - vt[k];
            const T lg = lam * gm;
            ba = delta + lg * ba; aa = lg * aa;
        }
    }
    maps[(seg * 4 + 0) * 64 + lane] = ar; maps[(seg * 4 + 1) * 64 + lane] = br;
    maps[(seg * 4 + 2) * 64 + lane] = aa; maps[(seg * 4 + 3) * 64 + lane] = ba;
    __syncthreads();
    T run_ret = last_v[n], run_adv = (T)0;
    for (int j = S - 1; j > seg; --j) {                                  // later segments first
        run_ret = maps[(j * 4 + 0) * 64 + lane] * run_ret + maps[(j * 4 + 1) * 64 + lane];
        run_adv = maps[(j * 4 + 2) * 64 + lane] * run_adv + maps[(j * 4 + 3) * 64 + lane];
    }
    if (!env_live) return;
#pragma unroll
    for (int k = CH - 1; k >= 0; --k) {
        const int t = t0 + k;
        if (t < Tn) {
            const size_t idx = (size_t)t * N + n;
            const T gm = gamma * m[k];
            run_ret = r[k] + gm * run_ret;
            if (use_gae) {
                const T vn = (t == t_last) ? v_after : vt[k + 1 < CH ? k + 1 : k];
                run_adv = run_adv * (lam * gm) + (r[k] + gm * vn - vt[k]);
            } else {
                run_adv = run_ret - vt[k];
            }
            if (term_v) rew[idx] = r[k];
            ret[idx] = run_ret;
            adv[idx] = run_adv;
        }
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


}  // namespace scg

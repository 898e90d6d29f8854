// scg_rng.h — counter-based Philox4x32-10 streams for the env kernels (device + host).
//
// Stateless addressing (identical in oracle/rng.py so the CPU oracle and the GPU draw the same numbers):
//   key     = (seed & 0xffffffff, seed >> 32)
//   counter = (global_env_id, episode_index, step_index, tag)
//   tag     = (channel << 16) | (item << 8) | block
//   channel: 0 reset, 1 action, 2 dynamics, 3 observation, 4 random-action
//   reset draws (channel 0, step 0): item = group (0 initial state, 1 inertial parameters, 2 disturbance
//                offsets).  Pair layout: variable j of a group uses block j/2 and the word pair (2*(j&1), 2*(j&1)+1):
//                uniform / choice / integer draws consume the first word, normal draws both.  Compact layout, used
//                for groups 0 and 1 when none of the group's randomised variables is a normal draw: variable j uses
//                21-BIT FIELD j%6 of block j/6 (compact_word below: fields 0-3 = the top 21 bits of the block's four
//                words, fields 4 and 5 = their 11 / 10 low bits paired up), handed to the distributions in the top
//                bits of a word.  A Philox block costs ~20 quarter-rate 32x32 multiplies and the auto-reset path of
//                the step kernel is its Philox blocks: the six initial-state draws of the planar quadrotor are ONE
//                block (round 4: two, four 32-bit words each), Quadrotor3D's twelve are two (three); 2^-21 is
//                still finer than any initial-state or inertial-parameter range needs.
//                j = INIT_STATE_LABELS index | inertial parameter index | 4 * scg_channel + list index
//   u01(word)    = ((word >> 8) + 0.5) * 2^-24     (exact in fp32)
//   normal(w0,w1)= sqrt(-2 ln u01(w0)) * cos(2 pi u01(w1))
//
// The reference draws from one NumPy PCG64 generator per env (benchmark_env.py:210,
// disturbances.py:33-35); reproducing 65 536 sequential PCG64 streams on a GPU would serialise every
// draw, so production uses Philox and parity tests inject states/noise from the host.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SCG_HD __host__ __device__ __forceinline__
#else
#define SCG_HD inline
#endif

namespace scg {

enum : uint32_t { RNG_CH_RESET = 0, RNG_CH_ACTION = 1, RNG_CH_DYNAMICS = 2, RNG_CH_OBSERVATION = 3,
                  RNG_CH_RANDOM_ACTION = 4 };
enum : uint32_t { RNG_GROUP_INIT = 0, RNG_GROUP_PARAM = 1, RNG_GROUP_DISTURB = 2 };

struct U4 { uint32_t x, y, z, w; };

SCG_HD uint32_t rng_tag(uint32_t channel, uint32_t item, uint32_t block) {
    return (channel << 16) | (item << 8) | block;
}

SCG_HD void mulhilo32(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
    uint64_t p = (uint64_t)a * (uint64_t)b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
}

SCG_HD U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo32(0xD2511F53u, c.x, hi0, lo0);
        mulhilo32(0xCD9E8D57u, c.z, hi1, lo1);
        U4 n;
        n.x = hi1 ^ c.y ^ k0;
        n.y = lo1;
        n.z = hi0 ^ c.w ^ k1;
        n.w = lo0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

struct RngKey { uint32_t k0, k1; };

SCG_HD U4 rng_words(RngKey key, uint32_t env, uint32_t episode, uint32_t step, uint32_t tag) {
    U4 c{env, episode, step, tag};
    return philox4x32_10(c, key.k0, key.k1);
}

SCG_HD uint32_t u4_get(const U4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// Compact reset-draw layout: 21-bit field k (0..5) of a block, left-aligned in a word (u01 / int_below read top bits).
constexpr int COMPACT_PER_BLOCK = 6;
SCG_HD uint32_t compact_word(const U4& v, int k) {
    if (k < 4) return u4_get(v, k) & 0xfffff800u;
    const uint32_t a = k == 4 ? v.x : v.z, b = k == 4 ? v.y : v.w;
    return (a << 21) | ((b & 0x3ffu) << 11);
}

template <typename T>
SCG_HD T u01(uint32_t w) { return ((T)(w >> 8) + (T)0.5) * (T)(1.0 / 16777216.0); }

SCG_HD uint32_t int_below(uint32_t w, uint32_t bound) { return (uint32_t)(((uint64_t)w * (uint64_t)bound) >> 32); }

}  // namespace scg

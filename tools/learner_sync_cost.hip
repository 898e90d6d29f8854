// learner_sync_cost.hip — what would a PERSISTENT multi-minibatch PPO learner launch pay for synchronisation, against the two launch
// boundaries per optimiser step it would replace?    Everything of an optimiser step EXCEPT the tile
// compute (identical in both forms) in the learner's own geometry — 254 workgroups x 256 threads, one per CU (150 KB of LDS each):
//   fill      every workgroup reads its network's 73 KB parameter image from global memory into LDS
//   publish   every workgroup writes its 74 KB partial-gradient vector (16-byte write-through stores, as scg_learn.hip does)
//   ---- seam 1: kernel boundary | grid barrier
//   reduce    2 x 18.7 K words: each word = fixed-order sum of the 127 partials of its network, then the "Adam" write of the parameter
//   ---- seam 2: kernel boundary | grid barrier
// Form A: two kernels per step (fill + publish | reduce), 48 steps captured in one HIP graph       = what scg_ppo_step is today.
// Form B: ONE launch looping 48 steps with an XCD-hierarchical grid barrier at each seam (per-XCC arrival counter, XCC leader -> top
//         counter -> generation word per XCC; MI355X_MICROARCH.md row barrier-xcd), reduction slices owned by the workgroups.
// Prints us per step for both.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/learner_sync_cost tools/learner_sync_cost.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)

constexpr int NWG = 127, NETS = 2, THREADS = 256;
constexpr int PARAM_WORDS = 18432 + 256;            // ~ one network's flat parameters (12-128-128-2): 73 KB
constexpr int PARTIAL = 18944;                      // words of one workgroup's partial vector (scg_learn.hip: PARTIAL_STRIDE for this shape ~ 74 KB)
constexpr int LDS_BYTES = 150 * 1024;
typedef unsigned int u32x4 __attribute__((vector_size(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Bar {                                        // every word on its own 128-byte line
    unsigned census[8][32], xcnt[8][32], xgen[8][32], top[32], flat[32], timeout[32];
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
__device__ __forceinline__ unsigned ld_relaxed(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_relaxed(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool spin_until(unsigned* p, unsigned want, unsigned* timeout) {
    if (ld_relaxed(timeout)) return false;                      // (sticky: after one timeout nothing spins again)
    for (unsigned spins = 0; ld_relaxed(p) < want; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if (spins > (1u << 22)) { st_relaxed(timeout, 1u); return false; }
    }
    return true;
}

// XCD-hierarchical grid barrier, epoch = 1, 2, ... within the launch (counters are monotonic: zeroed by a memset node before the launch).
__device__ __forceinline__ void grid_barrier(Bar* b, unsigned epoch, unsigned xcc, unsigned n_xcc) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every wave drains its own (write-through) stores
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned mine = ld_relaxed(&b->census[xcc][0]);
        const unsigned a = __hip_atomic_fetch_add(&b->xcnt[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a + 1 == mine * epoch) {                            // last arriver of this XCC: its leader for this epoch
            const unsigned t = __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1 == n_xcc * epoch) {                       // last XCC: release everybody
                for (unsigned x = 0; x < 8; ++x) st_relaxed(&b->xgen[x][0], epoch);
            } else {
                spin_until(&b->xgen[xcc][0], epoch, &b->timeout[0]);
            }
        } else {
            spin_until(&b->xgen[xcc][0], epoch, &b->timeout[0]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (threadIdx.x != 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__device__ __forceinline__ void fill_and_publish(const float* __restrict__ params, float* __restrict__ partials, float* lds, int wg, int net, float salt) {
    const f32x4* src = reinterpret_cast<const f32x4*>(params + (size_t)net * PARAM_WORDS);
    f32x4 v[PARAM_WORDS / 4 / THREADS + 1];
#pragma unroll
    for (int k = 0; k < PARAM_WORDS / 4 / THREADS + 1; ++k) {
        const int i = threadIdx.x + k * THREADS;
        v[k] = i < PARAM_WORDS / 4 ? __builtin_nontemporal_load(src + i) : (f32x4){0, 0, 0, 0};
    }
#pragma unroll
    for (int k = 0; k < PARAM_WORDS / 4 / THREADS + 1; ++k) {
        const int i = threadIdx.x + k * THREADS;
        if (i < PARAM_WORDS / 4) reinterpret_cast<f32x4*>(lds)[i] = v[k];
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc((void*)partials, 0, 0xffffffff, 0x00020000);
    const unsigned base = (unsigned)(((size_t)wg * NETS + net) * PARTIAL * sizeof(float));
    for (int k = 4 * threadIdx.x; k < PARTIAL; k += 4 * THREADS) {
        f32x4 x = *reinterpret_cast<const f32x4*>(lds + (k % PARAM_WORDS));
        x.x += salt;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), pr, base + 4u * (unsigned)k, 0, 17);
    }
}

// word k of network `net`: fixed-order sum over the 127 partials, four waves x 32 partials then a fixed-order sum of the four
__device__ __forceinline__ void reduce_words(const float* __restrict__ partials, float* __restrict__ params, float (*part)[64], int k0, int net, int words) {
    const int kl = threadIdx.x & 63, grp = threadIdx.x >> 6, k = k0 + kl;
    float s = 0.0f;
    if (k < words) {
#pragma unroll 8
        for (int g = grp; g < NWG; g += 4) s += __builtin_nontemporal_load(partials + ((size_t)g * NETS + net) * PARTIAL + k);
    }
    part[grp][kl] = s;
    __syncthreads();
    if (grp == 0 && k < words && k < PARAM_WORDS) {
        const float g = (part[0][kl] + part[1][kl]) + (part[2][kl] + part[3][kl]);
        const unsigned p = (unsigned)(((size_t)net * PARAM_WORDS + k) * sizeof(float));
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)params, 0, 0xffffffff, 0x00020000);
        float old = params[(size_t)net * PARAM_WORDS + k];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, old * 0.999f + 1e-6f * g), rr, p, 0, 17);
    }
    __syncthreads();
}

__global__ __launch_bounds__(THREADS, 1) void publish_kernel(const float* params, float* partials, float salt) {
    extern __shared__ __align__(16) float lds[];
    fill_and_publish(params, partials, lds, blockIdx.x, blockIdx.y, salt);
}
__global__ __launch_bounds__(THREADS) void reduce_kernel(const float* partials, float* params) {
    __shared__ float part[4][64];
    reduce_words(partials, params, part, blockIdx.x * 64, blockIdx.y, PARTIAL);
}

__global__ __launch_bounds__(THREADS, 1) void persistent_kernel(float* params, float* partials, Bar* bar, int steps, float salt) {
    extern __shared__ __align__(16) float lds[];
    __shared__ float part[4][64];
    __shared__ unsigned s_nxcc;
    const int wg = blockIdx.x % NWG, net = blockIdx.x / NWG, flat = blockIdx.x, total = NWG * NETS;
    const unsigned xcc = xcc_id();
    // census + one flat barrier: how many workgroups does each XCC hold?
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&bar->census[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&bar->flat[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        spin_until(&bar->flat[0], (unsigned)total, &bar->timeout[0]);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        unsigned n = 0;
        for (int x = 0; x < 8; ++x) n += ld_relaxed(&bar->census[x][0]) != 0u;
        s_nxcc = n;
    }
    __syncthreads();
    const unsigned n_xcc = s_nxcc;
    unsigned epoch = 0;
    // reduction slices: 64-word chunks of both networks dealt round-robin to the workgroups
    constexpr int CHUNKS = (PARTIAL + 63) / 64;
    for (int s = 0; s < steps; ++s) {
        fill_and_publish(params, partials, lds, wg, net, salt + (float)s);
        grid_barrier(bar, ++epoch, xcc, n_xcc);
        for (int c = flat; c < CHUNKS * NETS; c += total) reduce_words(partials, params, part, (c % CHUNKS) * 64, c / CHUNKS, PARTIAL);
        grid_barrier(bar, ++epoch, xcc, n_xcc);
    }
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 48, reps = argc > 2 ? atoi(argv[2]) : 50;
    float *params, *partials;
    Bar* bar;
    CHECK(hipMalloc(&params, sizeof(float) * NETS * PARAM_WORDS));
    CHECK(hipMalloc(&partials, sizeof(float) * (size_t)NWG * NETS * PARTIAL));
    CHECK(hipMalloc(&bar, sizeof(Bar)));
    CHECK(hipMemset(params, 0, sizeof(float) * NETS * PARAM_WORDS));
    CHECK(hipFuncSetAttribute((const void*)publish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CHECK(hipFuncSetAttribute((const void*)persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // ---- form A: two kernels per step, `steps` steps in one graph
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int s = 0; s < steps; ++s) {
        publish_kernel<<<dim3(NWG, NETS), dim3(THREADS), LDS_BYTES, st>>>(params, partials, (float)s);
        reduce_kernel<<<dim3((PARTIAL + 63) / 64, NETS), dim3(THREADS), 0, st>>>(partials, params);
    }
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double a_us = 1e3 * ms / reps / steps;
    // ---- form B: one persistent launch per `steps` steps (memset of the barrier words + launch, in a graph as well)
    hipGraph_t g2; hipGraphExec_t ge2;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    CHECK(hipMemsetAsync(bar, 0, sizeof(Bar), st));
    persistent_kernel<<<dim3(NWG * NETS), dim3(THREADS), LDS_BYTES, st>>>(params, partials, bar, steps, 0.5f);
    CHECK(hipStreamEndCapture(st, &g2));
    CHECK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) CHECK(hipGraphLaunch(ge2, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(ge2, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double b_us = 1e3 * ms / reps / steps;
    Bar hb;
    CHECK(hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost));
    printf("learner step minus tile compute, %d workgroups x %d threads, %d KB LDS, %d KB partial per workgroup, %d steps per graph / launch\n",
           NWG * NETS, THREADS, LDS_BYTES / 1024, (int)(PARTIAL * sizeof(float) / 1024), steps);
    printf("A two kernels per step (fill+publish | reduce+step), HIP graph : %7.2f us per step\n", a_us);
    printf("B one persistent launch, two XCD-hierarchical grid barriers   : %7.2f us per step   (B - A = %+.2f us; timeout flag %u; XCC census",
           b_us, b_us - a_us, hb.timeout[0]);
    for (int x = 0; x < 8; ++x) printf(" %u", hb.census[x][0]);
    printf(")\n");
    return 0;
}

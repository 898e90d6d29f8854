// issue_rate.hip — VALU issue interval and dependent-issue latency on gfx950, measured in shader-clock ticks inside the kernel.
//   hipcc --offload-arch=gfx950 -O3 -o tools/issue_rate tools/issue_rate.hip ;  run on the GPU box: tools/issue_rate
// For each instruction kind: CH independent chains of N instructions each, interleaved (chain c's k-th instruction depends on its
// (k-1)-th), CH = 1 (fully dependent), 2, 4, 8; W = waves per SIMD that run the same stream side by side (1 or 2: one workgroup of
// 4 or 8 waves on one CU).  ticks / instruction at CH = 8 is the ISSUE interval of a wave64 instruction, at CH = 1 the DEPENDENT-issue
// interval (what a serial chain pays per instruction).  bench.py's `chain_latency` / `valu_issue` objects use these numbers
// (profiles/r05_issue_rate.txt).  The clock itself is calibrated against HIP events (ticks per ns).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)

enum { K_FMA = 0, K_PKFMA = 1, K_MULLO = 2, K_RCP = 3, K_MAD64 = 4 };
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND, int CH>
__global__ __launch_bounds__(512) void k_issue(unsigned long long* out, float* sink, int iters) {
    float x[CH]; f2 p[CH]; unsigned u[CH]; unsigned long long q[CH];
    const float a = 0.999f + 1e-7f * threadIdx.x, b = 0.5f;
    const f2 pa = {a, a}, pb = {b, b};
#pragma unroll
    for (int c = 0; c < CH; ++c) { x[c] = threadIdx.x * 1e-3f + c; p[c] = (f2){x[c], x[c] + 1.0f}; u[c] = threadIdx.x * 7u + c; q[c] = u[c]; }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                else if constexpr (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[c]) : "v"(pa), "v"(pb));
                else if constexpr (KIND == K_MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[c]) : "v"(0x9E3779B9u));
                else if constexpr (KIND == K_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[c]));
                else asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[c]) : "v"(u[c]), "v"(0xD2511F53u) : "vcc");
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c] + p[c].x + p[c].y + (float)u[c] + (float)q[c];
    sink[threadIdx.x] = s;
}

template <int KIND, int CH>
static void run(const char* name, unsigned long long* d, float* sink, double ticks_per_ns) {
    const int iters = 200;
    for (int waves = 4; waves <= 8; waves += 4) {
        k_issue<KIND, CH><<<1, 64 * waves>>>(d, sink, iters); CK(hipDeviceSynchronize());
        k_issue<KIND, CH><<<1, 64 * waves>>>(d, sink, iters); CK(hipDeviceSynchronize());
        unsigned long long t[8]; CK(hipMemcpy(t, d, 8 * waves, hipMemcpyDeviceToHost));
        double mx = 0; for (int w = 0; w < waves; ++w) mx = t[w] > mx ? t[w] : mx;
        const double per = mx / ((double)iters * 64 * CH);
        printf("%-14s chains %d  waves/SIMD %d: %6.2f ticks per instruction per wave (%5.2f ns), SIMD issue interval %5.2f ticks\n", name, CH, waves / 4, per,
               per / ticks_per_ns, per / (waves / 4));
    }
}

__global__ __launch_bounds__(64) void k_calib(unsigned long long* out, float* sink, int iters) {
    float x = threadIdx.x * 1e-3f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 256; ++k) x = __builtin_fmaf(x, 0.999f, 0.5f);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = x;
}

int main() {
    unsigned long long* d; float* sink; CK(hipMalloc(&d, 64)); CK(hipMalloc(&sink, 4 * 512));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    k_calib<<<1, 64>>>(d, sink, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k_calib<<<1, 64>>>(d, sink, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long ticks; CK(hipMemcpy(&ticks, d, 8, hipMemcpyDeviceToHost));
    const double tpn = ticks / (ms * 1e6);
    printf("clock: %llu ticks in %.1f us -> %.4f ticks per ns (s_memtime); %.2f ticks per dependent v_fma_f32 (compiler-scheduled chain)\n", ticks, ms * 1e3, tpn,
           (double)ticks / (iters * 256.0));
    run<K_FMA, 1>("v_fma_f32", d, sink, tpn); run<K_FMA, 2>("v_fma_f32", d, sink, tpn); run<K_FMA, 4>("v_fma_f32", d, sink, tpn); run<K_FMA, 8>("v_fma_f32", d, sink, tpn);
    run<K_PKFMA, 1>("v_pk_fma_f32", d, sink, tpn); run<K_PKFMA, 2>("v_pk_fma_f32", d, sink, tpn); run<K_PKFMA, 4>("v_pk_fma_f32", d, sink, tpn); run<K_PKFMA, 8>("v_pk_fma_f32", d, sink, tpn);
    run<K_MULLO, 1>("v_mul_lo_u32", d, sink, tpn); run<K_MULLO, 8>("v_mul_lo_u32", d, sink, tpn);
    run<K_MAD64, 1>("v_mad_u64_u32", d, sink, tpn); run<K_MAD64, 8>("v_mad_u64_u32", d, sink, tpn);
    run<K_RCP, 1>("v_rcp_f32", d, sink, tpn); run<K_RCP, 8>("v_rcp_f32", d, sink, tpn);
    return 0;
}

// Launch-latency floor probes for the 65 536-env regime (1024 waves, one per SIMD): how long does a graph-replayed
// launch take when the kernel does (0) nothing, (1) one HBM round trip + stores, (2) two dependent round trips,
// (3) one round trip + a dependent chain of K FMAs.  Build: hipcc --offload-arch=gfx950 -O3 -o floor floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Args { float* s[8]; int* step; const float* table; float* out[12]; int n; };
__global__ __launch_bounds__(256) void k_empty(Args a) {}
__global__ __launch_bounds__(256) void k_round1(Args a) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= a.n) return;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = a.s[k][i];
#pragma unroll
    for (int k = 0; k < 8; ++k) a.s[k][i] = v[k] + 1.0f;
#pragma unroll
    for (int k = 0; k < 12; ++k) a.out[k][i] = v[k & 7];
}
__global__ __launch_bounds__(256) void k_round2(Args a) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= a.n) return;
    float v[8];
    int st = a.step[i];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = a.s[k][i];
    float r[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) r[k] = a.table[(st & 255) * 6 + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) a.s[k][i] = v[k] + r[k % 6];
#pragma unroll
    for (int k = 0; k < 12; ++k) a.out[k][i] = v[k & 7];
    a.step[i] = st + 1;
}
template <int K>
__global__ __launch_bounds__(256) void k_chain(Args a) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= a.n) return;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = a.s[k][i];
    float x = v[0];
#pragma unroll 8
    for (int k = 0; k < K; ++k) x = __builtin_fmaf(x, 0.999f, v[1]);
    v[0] = x;
#pragma unroll
    for (int k = 0; k < 8; ++k) a.s[k][i] = v[k];
#pragma unroll
    for (int k = 0; k < 12; ++k) a.out[k][i] = v[k & 7];
}
// store-phase probes: 8 loads, then 20 dword stores per lane (a) as launched, (b) with the block index remapped so that
// each XCD (blocks are dealt round-robin to the 8 XCDs) owns a contiguous eighth of every array, (c) as 5 float4 stores
template <int MODE>
__global__ __launch_bounds__(256) void k_stores(Args a, float4* wide) {
    int b = blockIdx.x;
    if (MODE == 1) { const int per = gridDim.x / 8; b = (b % 8) * per + b / 8; }
    int i = b * 256 + threadIdx.x; if (i >= a.n) return;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = a.s[k][i];
    if (MODE == 3) {
#pragma unroll
        for (int k = 0; k < 12; ++k) __builtin_nontemporal_store(v[k & 7], &a.out[k][i]);
#pragma unroll
        for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(v[k] + 1.0f, &a.s[k][i]);
    } else if (MODE == 4) {      // only 8 stores
#pragma unroll
        for (int k = 0; k < 8; ++k) a.s[k][i] = v[k] + 1.0f;
    } else if (MODE == 5) {      // 40 stores
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int k = 0; k < 12; ++k) a.out[k][i + r * a.n] = v[k & 7];
#pragma unroll
        for (int k = 0; k < 8; ++k) a.s[k][i + r * a.n] = v[k] + 1.0f;
        }
    } else if (MODE == 2) {
#pragma unroll
        for (int k = 0; k < 5; ++k) wide[(size_t)k * a.n + i] = make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) a.out[k][i] = v[k & 7];
#pragma unroll
        for (int k = 0; k < 8; ++k) a.s[k][i] = v[k] + 1.0f;
    }
}
__global__ __launch_bounds__(256) void k_loads_only(Args a) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= a.n) return;
    float acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += a.s[k][i];
    if (acc == 12345.678f) a.out[0][i] = acc;
}
template <int K>
__global__ __launch_bounds__(256) void k_chain_flat(Args a) {      // same chain, fully unrolled: K*8 bytes of straight-line code
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= a.n) return;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = a.s[k][i];
    float x = v[0];
#pragma unroll
    for (int k = 0; k < K; ++k) x = __builtin_fmaf(x, 0.999f + 1e-6f * (k & 15), v[1]);
    v[0] = x;
#pragma unroll
    for (int k = 0; k < 8; ++k) a.s[k][i] = v[k];
#pragma unroll
    for (int k = 0; k < 12; ++k) a.out[k][i] = v[k & 7];
}
template <int K, int ILP>
__global__ __launch_bounds__(256) void k_ilp(Args a) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= a.n) return;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = a.s[k][i];
    float x[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) x[j] = v[j & 7];
#pragma unroll 4
    for (int k = 0; k < K / ILP; ++k) {
#pragma unroll
        for (int j = 0; j < ILP; ++j) x[j] = __builtin_fmaf(x[j], 0.999f, v[7]);
    }
    float acc = 0;
#pragma unroll
    for (int j = 0; j < ILP; ++j) acc += x[j];
    v[0] = acc;
#pragma unroll
    for (int k = 0; k < 8; ++k) a.s[k][i] = v[k];
#pragma unroll
    for (int k = 0; k < 12; ++k) a.out[k][i] = v[k & 7];
}
typedef float float2v __attribute__((ext_vector_type(2)));
template <int K>
__global__ __launch_bounds__(256) void k_pk(Args a) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= a.n) return;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = a.s[k][i];
    float2v x = {v[0], v[1]}, m = {0.999f, 0.998f}, c = {v[6], v[7]};
#pragma unroll 8
    for (int k = 0; k < K; ++k) x = __builtin_elementwise_fma(x, m, c);
    v[0] = x.x + x.y;
#pragma unroll
    for (int k = 0; k < 8; ++k) a.s[k][i] = v[k];
#pragma unroll
    for (int k = 0; k < 12; ++k) a.out[k][i] = v[k & 7];
}
// clock calibration: shader-clock ticks (s_memtime) around K dependent FMAs, against the event-timed duration
__global__ __launch_bounds__(64) void k_calib(unsigned long long* out, float* sink, int iters) {
    float x = threadIdx.x * 1e-3f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 256; ++k) x = __builtin_fmaf(x, 0.999f, 0.5f);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = t1 - t0; }
    sink[threadIdx.x] = x;
}
template <typename F> float time_graph(F launch, hipStream_t st, int reps) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int k = 0; k < 1000; ++k) launch();
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1000.0f / (1000.0f * reps);
}
int main(int argc, char** argv) {
    int n = argc > 1 ? atoi(argv[1]) : 65536;
    Args a; a.n = n;
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int k = 0; k < 8; ++k) { CK(hipMalloc(&a.s[k], n * 8)); CK(hipMemset(a.s[k], 0, n * 8)); }
    for (int k = 0; k < 12; ++k) CK(hipMalloc(&a.out[k], n * 8));
    CK(hipMalloc(&a.step, n * 4)); CK(hipMemset(a.step, 0, n * 4));
    float* tab; CK(hipMalloc(&tab, 256 * 6 * 4)); CK(hipMemset(tab, 0, 256 * 6 * 4)); a.table = tab;
    dim3 grid((n + 255) / 256), block(256);
    printf("n=%d grid=%d\n", n, grid.x);
    {
        unsigned long long* d; float* sink; CK(hipMalloc(&d, 8)); CK(hipMalloc(&sink, 256));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 4000;
        k_calib<<<1, 64, 0, st>>>(d, sink, iters); hipStreamSynchronize(st);
        hipEventRecord(e0, st); k_calib<<<1, 64, 0, st>>>(d, sink, iters); hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long ticks; hipMemcpy(&ticks, d, 8, hipMemcpyDeviceToHost);
        printf("calibration: %d dependent FMAs: %.1f us, %llu shader-clock ticks -> %.3f ticks/ns, %.2f ticks per FMA, %.2f ns per FMA\n",
               iters * 256, ms * 1e3, ticks, ticks / (ms * 1e6), (double)ticks / (iters * 256.0), ms * 1e6 / (iters * 256.0));
    }
    printf("empty            %.3f us/launch\n", time_graph([&] { k_empty<<<grid, block, 0, st>>>(a); }, st, 10));
    printf("1 round + stores %.3f us/launch\n", time_graph([&] { k_round1<<<grid, block, 0, st>>>(a); }, st, 10));
    printf("2 rounds + stores %.3f us/launch\n", time_graph([&] { k_round2<<<grid, block, 0, st>>>(a); }, st, 10));
    printf("1 round + 400 fma %.3f us/launch\n", time_graph([&] { k_chain<400><<<grid, block, 0, st>>>(a); }, st, 10));
    printf("1 round + 800 fma %.3f us/launch\n", time_graph([&] { k_chain<800><<<grid, block, 0, st>>>(a); }, st, 10));
    printf("1 round + 1600 fma %.3f us/launch\n", time_graph([&] { k_chain<1600><<<grid, block, 0, st>>>(a); }, st, 10));
    printf("800 fma ILP2      %.3f us/launch\n", time_graph([&] { k_ilp<800, 2><<<grid, block, 0, st>>>(a); }, st, 10));
    printf("800 fma ILP4      %.3f us/launch\n", time_graph([&] { k_ilp<800, 4><<<grid, block, 0, st>>>(a); }, st, 10));
    printf("800 pk_fma chain  %.3f us/launch\n", time_graph([&] { k_pk<800><<<grid, block, 0, st>>>(a); }, st, 10));
    float4* wide; CK(hipMalloc(&wide, (size_t)n * 16 * 5));
    printf("8 loads only       %.3f us/launch\n", time_graph([&] { k_loads_only<<<grid, block, 0, st>>>(a); }, st, 10));
    printf("8 loads + 20 stores %.3f us/launch\n", time_graph([&] { k_stores<0><<<grid, block, 0, st>>>(a, wide); }, st, 10));
    printf("  ... XCD-contiguous blocks %.3f us/launch\n", time_graph([&] { k_stores<1><<<grid, block, 0, st>>>(a, wide); }, st, 10));
    printf("  ... as 5 float4 stores %.3f us/launch\n", time_graph([&] { k_stores<2><<<grid, block, 0, st>>>(a, wide); }, st, 10));
    printf("  ... nontemporal stores %.3f us/launch\n", time_graph([&] { k_stores<3><<<grid, block, 0, st>>>(a, wide); }, st, 10));
    printf("  ... 8 stores only %.3f us/launch\n", time_graph([&] { k_stores<4><<<grid, block, 0, st>>>(a, wide); }, st, 10));
    printf("  ... 40 stores %.3f us/launch\n", time_graph([&] { k_stores<5><<<grid, block, 0, st>>>(a, wide); }, st, 10));
    printf("400 fma flat code  %.3f us/launch\n", time_graph([&] { k_chain_flat<400><<<grid, block, 0, st>>>(a); }, st, 10));
    printf("800 fma flat code  %.3f us/launch\n", time_graph([&] { k_chain_flat<800><<<grid, block, 0, st>>>(a); }, st, 10));
    printf("1600 fma flat code %.3f us/launch\n", time_graph([&] { k_chain_flat<1600><<<grid, block, 0, st>>>(a); }, st, 10));
    dim3 g128((n + 127) / 128), b128(128);
    printf("800 fma, 128-thread WGs %.3f us/launch\n", time_graph([&] { k_chain<800><<<g128, b128, 0, st>>>(a); }, st, 10));
    dim3 g64((n + 63) / 64), b64(64);
    printf("800 fma, 64-thread WGs %.3f us/launch\n", time_graph([&] { k_chain<800><<<g64, b64, 0, st>>>(a); }, st, 10));
    printf("empty, 64-thread WGs %.3f us/launch\n", time_graph([&] { k_empty<<<g64, b64, 0, st>>>(a); }, st, 10));
    return 0;
}

// Does kernarg preloading (-mllvm -amdgpu-kernarg-preload-count=N: the command processor places the first N dword kernel
// arguments in SGPRs before the wave starts) shorten a one-wave-per-SIMD launch?  A kernel with a flat scalar signature —
// 8 array pointers + n — does one load round and 8 stores; built twice (with / without the option) and timed inside a HIP
// graph of 1000 launches.  Build: hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-kernarg-preload-count=16] -o p tools/preload_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(64) void k_flat(float* s0, float* s1, float* s2, float* s3, float* s4, float* s5, float* s6, float* s7, int n) {
    int i = blockIdx.x * 64 + threadIdx.x; if (i >= n) return;
    float v0 = s0[i], v1 = s1[i], v2 = s2[i], v3 = s3[i], v4 = s4[i], v5 = s5[i], v6 = s6[i], v7 = s7[i];
    s0[i] = v1 + 1.0f; s1[i] = v2 + 1.0f; s2[i] = v3 + 1.0f; s3[i] = v4 + 1.0f; s4[i] = v5 + 1.0f; s5[i] = v6 + 1.0f; s6[i] = v7 + 1.0f; s7[i] = v0 + 1.0f;
}
__global__ __launch_bounds__(64) void k_empty(float* s0, int n) {}
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
    return ms * 1000.0f / (1000.0f * reps);
}
int main(int argc, char** argv) {
    int n = argc > 1 ? atoi(argv[1]) : 65536;
    hipStream_t st; hipStreamCreate(&st);
    float* s[8];
    for (int k = 0; k < 8; ++k) { hipMalloc(&s[k], n * 4); hipMemset(s[k], 0, n * 4); }
    dim3 grid((n + 63) / 64), block(64);
    for (int rep = 0; rep < 3; ++rep) {
        printf("empty %.3f us   1 round + 8 stores %.3f us\n", time_graph([&] { k_empty<<<grid, block, 0, st>>>(s[0], n); }, st, 10),
               time_graph([&] { k_flat<<<grid, block, 0, st>>>(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], n); }, st, 10));
    }
    return 0;
}

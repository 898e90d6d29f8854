// hbm_mix.hip — what HBM3E gives a kernel by READ : WRITE mix and store width, beyond the 256 MiB Infinity Cache (MI355X).
//   hipcc --offload-arch=gfx950 -O3 -o tools/hbm_mix tools/hbm_mix.hip ;  run on the GPU box: tools/hbm_mix
// The step kernels write ~3/4 of their bytes (Quadrotor2D: 57 B read, 175 B written per env-step) as 4-byte-per-lane SoA stores into
// ~45 separate arrays; the guide's 6.3 TB/s is a 1 : 1 float4 copy.  This prints the ceiling for the mixes and store shapes in between,
// so that the streaming-regime fraction of the step kernel can be read against what the memory system gives THAT traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)

// R arrays read, W arrays written, 16 bytes per lane per array, n float4 elements per array
template <int R, int W>
__global__ __launch_bounds__(256) void k_mix16(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 acc = {1.0f, 2.0f, 3.0f, 4.0f};
#pragma unroll
    for (int r = 0; r < R; ++r) { const float4 v = in[r * n + i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
#pragma unroll
    for (int w = 0; w < W; ++w) out[w * n + i] = acc;
    if (W == 0 && acc.x == 12345.678f) out[i] = acc;
}
// the step kernel's shape: R dword-per-lane SoA arrays read, W written (n elements each), one thread = one element
template <int R, int W, int BLK>
__global__ __launch_bounds__(BLK) void k_mix4(const float* __restrict__ in, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (i >= n) return;
    float acc = 1.0f;
#pragma unroll
    for (int r = 0; r < R; ++r) acc += in[r * n + i];
#pragma unroll
    for (int w = 0; w < W; ++w) out[w * n + i] = acc + (float)w;
}
template <typename F> static double run(F launch, double bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int k = 0; k < 3; ++k) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int k = 0; k < reps; ++k) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return bytes * reps / (ms * 1e-3) / 1e12;
}
int main() {
    const size_t n16 = (size_t)1 << 24;            // float4 elements per array: 256 MiB per array
    const size_t n4 = (size_t)1 << 24;             // floats per array: 64 MiB per array (x 58 arrays = 3.6 GiB)
    float4 *in16, *out16; CK(hipMalloc(&in16, 4 * n16 * 16)); CK(hipMalloc(&out16, 4 * n16 * 16));
    CK(hipMemset(in16, 0, 4 * n16 * 16));
    float *in4, *out4; CK(hipMalloc(&in4, 14 * n4 * 4)); CK(hipMalloc(&out4, 44 * n4 * 4));
    CK(hipMemset(in4, 0, 14 * n4 * 4));
    const dim3 g16((unsigned)(n16 / 256)), b(256);
    printf("16 bytes per lane per array, %zu MiB per array (TB/s of bytes moved)\n", n16 * 16 >> 20);
    printf("  read 1 : write 1 (copy)      %.2f\n", run([&] { k_mix16<1, 1><<<g16, b>>>(in16, out16, n16); }, 2.0 * n16 * 16));
    printf("  read 4 : write 0             %.2f\n", run([&] { k_mix16<4, 0><<<g16, b>>>(in16, out16, n16); }, 4.0 * n16 * 16));
    printf("  read 0 : write 4             %.2f\n", run([&] { k_mix16<0, 4><<<g16, b>>>(in16, out16, n16); }, 4.0 * n16 * 16));
    printf("  read 1 : write 3             %.2f\n", run([&] { k_mix16<1, 3><<<g16, b>>>(in16, out16, n16); }, 4.0 * n16 * 16));
    printf("4 bytes per lane per array (SoA rows), %zu MiB per array\n", n4 * 4 >> 20);
    printf("  read 1 : write 1, 256-thread WGs   %.2f\n", run([&] { k_mix4<1, 1, 256><<<dim3((unsigned)(n4 / 256)), dim3(256)>>>(in4, out4, n4); }, 2.0 * n4 * 4));
    printf("  read 14 : write 44, 256-thread WGs %.2f   (the Quadrotor2D step's row counts: 57 B in, 175 B out per element)\n",
           run([&] { k_mix4<14, 44, 256><<<dim3((unsigned)(n4 / 256)), dim3(256)>>>(in4, out4, n4); }, 58.0 * n4 * 4));
    printf("  read 14 : write 44, 64-thread WGs  %.2f\n", run([&] { k_mix4<14, 44, 64><<<dim3((unsigned)(n4 / 64)), dim3(64)>>>(in4, out4, n4); }, 58.0 * n4 * 4));
    printf("  read 0 : write 44, 256-thread WGs  %.2f\n", run([&] { k_mix4<0, 44, 256><<<dim3((unsigned)(n4 / 256)), dim3(256)>>>(in4, out4, n4); }, 44.0 * n4 * 4));
    printf("  read 14 : write 0 (+1), 256-thr    %.2f\n", run([&] { k_mix4<14, 1, 256><<<dim3((unsigned)(n4 / 256)), dim3(256)>>>(in4, out4, n4); }, 15.0 * n4 * 4));
    return 0;
}

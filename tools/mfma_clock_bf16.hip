// Sustained bf16 MFMA rate (v_mfma_f32_16x16x32_bf16) of the whole chip with random-looking operands: the ceiling of the
// bf16x6 split product of split.hip.  build: hipcc -O3 --offload-arch=gfx950 tools/mfma_clock_bf16.hip -o tools/mfma_clock_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[4], b[4];
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            seed = seed * 1664525u + 1013904223u; a[i][e] = (__bf16)((float)(int)(seed >> 16) * 1e-5f - 0.3f);
            seed = seed * 1664525u + 1013904223u; b[i][e] = (__bf16)((float)(int)(seed >> 16) * 1e-5f - 0.3f);
        }
    long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * i + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[4 * i + j], 0, 0, 0);
    }
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
    float* out; long long* clk;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clk, 4096 * 16);
    const int iters = 20000;
    for (int blocks : {8, 256, 512}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, clk, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double flop = (double)blocks * 4 * iters * 16 * (2.0 * 16 * 16 * 32);
        printf("blocks %4d: %.3f ms  %.0f TF bf16 (= %.0f TF fp32-equivalent at 6 products)  clock64/wall %.3f  cycles per MFMA per wave %.1f\n",
               blocks, ms, flop / ms / 1e9, flop / ms / 1e9 / 6, (double)h[0] / h[1], (double)h[0] / (iters * 16.0));
    }
    return 0;
}

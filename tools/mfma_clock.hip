// Sustained fp32 MFMA rate and the shader clock it runs at, full chip (grid = CUs x waves) vs a few workgroups.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_clock.hip -o tools/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int iters) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
    float* out; long long* clk;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clk, 4096 * 16);
    const int iters = 20000;
    for (int blocks : {8, 64, 256, 512, 1024}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, clk, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double flop = (double)blocks * 4 * iters * 4 * 4096.0;
        printf("blocks %4d: %.3f ms  %.1f TF   clock64 %lld  wall %lld  -> clock64/wall = %.3f (x100MHz if wall is 100 MHz); cycles per MFMA per wave %.1f\n",
               blocks, ms, flop / ms / 1e9, h[0], h[1], (double)h[0] / h[1], (double)h[0] / (iters * 4.0));
    }
    return 0;
}

// fp32 MFMA rate with FRESH operand registers per instruction (as in a GEMM k-step: 16 A and 16 B values per lane,
// 32 MFMAs), against the constant-operand loop of mfma_clock.hip.  Also with LDS reads feeding the operands.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_clock2.hip -o tools/mfma_clock2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>   // 0: operands in registers (rotating), 1: operands re-read from LDS every k-step
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* out, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = src[i];
    __syncthreads();
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float av[2][8], bv[2][8];
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) { av[i][e] = lds[(i * 8 + e) * 64 + lane]; bv[i][e] = lds[2048 + (i * 8 + e) * 64 + lane]; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            const int o = (it & 3) * 1024;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) { av[i][e] = lds[o + (i * 8 + e) * 64 + lane]; bv[i][e] = lds[4096 + o + (i * 8 + e) * 64 + lane]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][e], bv[0][e], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][e], bv[1][e], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][e], bv[0][e], a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][e], bv[1][e], a3, 0, 0, 0);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* src, float* out) {
    const int iters = 4000;
    for (int blocks : {256, 512}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, src, out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s blocks %4d: %.3f ms  %.1f TF\n", name, blocks, ms, (double)blocks * 4 * iters * 32 * 4096.0 / ms / 1e9);
    }
}
int main() {
    float *src, *out; hipMalloc(&src, 8192 * 4); hipMalloc(&out, 1024 * 256 * 4); hipMemset(src, 0, 8192 * 4);
    run<0>("registers", src, out);
    run<1>("lds-fed  ", src, out);
    return 0;
}

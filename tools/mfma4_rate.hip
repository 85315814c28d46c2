#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int FORM>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f4 acc[8]; f16v big[2];
    for (int i = 0; i < 8; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) big[i][r] = 0.f;
    float a = 1.0001f + threadIdx.x * 1e-6f, b = 0.9999f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (FORM == 0) acc[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u], 0, 0, 0);
            else if (FORM == 1) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u], 0, 0, 0);
            else big[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[u & 1], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    s += big[0][0] + big[1][5];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int FORM>
void run(const char* name, float* out, int wps, double flop_per_instr) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<FORM>, dim3(256 * wps), dim3(256), 0, 0, out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<FORM>, dim3(256 * wps), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 8 * wps;
    printf("%-28s waves/SIMD %d: %.2f ns per MFMA per SIMD (%.1f cycles @2.4GHz), chip %.1f TFLOP/s\n", name, wps, ms * 1e6 / n,
           ms * 1e6 / n * 2.4, flop_per_instr * n * 1024 / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 4 * 256 * 4);
    for (int wps : {1, 2}) {
        run<0>("mfma_f32_4x4x1 (16 blocks)", out, wps, 512);
        run<1>("mfma_f32_16x16x4", out, wps, 2048);
        run<2>("mfma_f32_32x32x2", out, wps, 4096);
    }
    return 0;
}

// Empirical operand/result layout of v_mfma_f32_4x4x1_16B_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    const int l = threadIdx.x;
    // A value encodes (block, i): 100*block + i+1 ; B value encodes 1000*(block) + 10*(j+1)
    float a = 100.f * (l / 4) + (l % 4 + 1);
    float b = (l % 4 == 0) ? 1.f : (l % 4 == 1 ? 10.f : (l % 4 == 2 ? 100.f : 1000.f));
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
    float* d; (void)hipMalloc(&d, 64 * 4 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 1, 2, 3, 4, 5, 63}) printf("lane %2d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}

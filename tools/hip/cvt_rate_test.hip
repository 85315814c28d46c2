// issue rate of v_cvt_pk_bf16_f32 on gfx950 against plain VALU work and a manual round-to-nearest-even: one wave per SIMD, a long
// dependent-free stream.   hipcc -O3 --offload-arch=gfx950 tools/hip/cvt_rate_test.hip -o /tmp/t && /tmp/t
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a[8], acc[8];
    unsigned u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x + i; acc[i] = 0.f; u[i] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            if (MODE == 0) {            // v_cvt_pk_bf16_f32
                unsigned p = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a[i], a[i + 1]}, bf16x2));
                u[i] ^= p;
            } else if (MODE == 1) {     // manual RNE (finite values): two bfe + two add3 + one perm
                unsigned x = __builtin_bit_cast(unsigned, a[i]), y = __builtin_bit_cast(unsigned, a[i + 1]);
                x = x + 0x7fffu + ((x >> 16) & 1u); y = y + 0x7fffu + ((y >> 16) & 1u);
                u[i] ^= __builtin_amdgcn_perm(y, x, 0x07060302u);
            } else {                    // two plain VALU ops
                acc[i] = acc[i] * 1.0001f + a[i]; acc[i + 1] = acc[i + 1] * 1.0001f + a[i + 1];
            }
            a[i] += 1.f; a[i + 1] += 1.f;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i] + (float)u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> float run(float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 256>>>(d, 16, 1.f);
    hipEventRecord(e0);
    k<MODE><<<256, 256>>>(d, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    const int iters = 20000;
    const float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters);
    // per iteration and wave: 4 pairs.  256 workgroups of 4 waves: one wave per SIMD
    printf("per pair [ns]: cvt_pk_bf16 (+2 adds +xor) %.2f   manual RNE (+2 adds +xor) %.2f   2 fma (+2 adds) %.2f\n",
           t0 * 1e6 / iters / 4, t1 * 1e6 / iters / 4, t2 * 1e6 / iters / 4);
    return 0;
}

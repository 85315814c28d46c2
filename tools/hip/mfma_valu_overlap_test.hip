// Do vector instructions of one wave issue under the matrix instructions of ANOTHER wave of the same SIMD on gfx950?
// Workgroups of 8 waves, one per CU: waves 0-3 (one per SIMD) run a stream of independent v_mfma_f32_16x16x32_bf16, waves 4-7 a stream of
// v_fma_f32 / v_cvt_pk_bf16_f32 / v_pk_add_f32.  Timed: matrix waves alone, vector waves alone, both.
//   hipcc -O3 --offload-arch=gfx950 tools/hip/mfma_valu_overlap_test.hip -o /tmp/t && /tmp/t
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, int do_m, int do_v, int vkind) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        if (do_m) {
            bf16x8 a, b;
            for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.f + threadIdx.x); b[e] = (__bf16)(0.5f + e); }
            f32x4 acc[8];
            for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
            }
            for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][3];
        }
    } else if (do_v) {
        float x[8], y[8];
        unsigned u[4] = {0, 0, 0, 0};
        for (int j = 0; j < 8; ++j) { x[j] = 1.f + threadIdx.x + j; y[j] = 0.f; }
        for (int it = 0; it < iters; ++it) {
            if (vkind == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) y[j] = y[j] * 1.0001f + x[j];           // 32 v_fma per iteration
            } else {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {                                     // the split's mix: cvt_pk, shift, mask, packed subtract
                        const unsigned p = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x[j], x[j + 1]}, bf16x2));
                        x[j] -= __builtin_bit_cast(float, p << 16); x[j + 1] -= __builtin_bit_cast(float, p & 0xffff0000u);
                        u[j >> 1] ^= p;
                        x[j] += 3.f; x[j + 1] += 5.f;
                    }
            }
        }
        for (int j = 0; j < 8; ++j) s += y[j] + x[j];
        s += (float)(u[0] ^ u[1] ^ u[2] ^ u[3]);
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
float run(float* d, int iters, int m, int v, int vk) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<256, 512>>>(d, 8, m, v, vk);
    hipEventRecord(e0);
    k<<<256, 512>>>(d, iters, m, v, vk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 20000;
    for (int vk = 0; vk < 2; ++vk) {
        const float tm = run(d, iters, 1, 0, vk), tv = run(d, iters, 0, 1, vk), tb = run(d, iters, 1, 1, vk);
        printf("%s: matrix waves alone %.0f us (%.1f cycles per MFMA at 2.4 GHz), vector waves alone %.0f us, both %.0f us  (sum %.0f, max %.0f)\n",
               vk == 0 ? "v_fma_f32 stream" : "split mix (cvt_pk / shift / and / sub)", tm, tm * 2400.f / iters / 8, tv, tb, tm + tv, tm > tv ? tm : tv);
    }
    return 0;
}

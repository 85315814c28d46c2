// Micro-benchmark: issue rate of v_pk_fma_f32 / v_fma_f32 operand forms on gfx950 (cycles per wave-instruction per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(X) X X X X X X X X X X X X X X X X
template <int FORM>
__global__ __launch_bounds__(256) void k(const float* __restrict__ w, float* out, int iters) {
    f2 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f2){(float)threadIdx.x, 1.f};
    f2 a = {1.0001f + threadIdx.x * 1e-6f, 0.9999f};
    f2 b = {1.0002f, 0.9998f};
    // uniform (SGPR) pair
    f2 sw = {w[0], w[1]};
    float s0 = w[2];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (FORM == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b));
            else if (FORM == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "s"(sw));
            else if (FORM == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[u]) : "v"(a), "s"(sw));
            else if (FORM == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[u]) : "v"(a), "v"(b));
            else if (FORM == 4) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[u].x) : "s"(s0), "v"(a.x));
            else if (FORM == 5) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[u].x) : "v"(b.x), "v"(a.x));
            else if (FORM == 6) asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(acc[u]) : "v"(a), "v"(b));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int FORM>
void run(const char* name, const float* w, float* out, int waves_per_simd) {
    const int iters = 40000;
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, w, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, w, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 16 * waves_per_simd;
    printf("%-34s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main() {
    float *w, *out; hipMalloc(&w, 64); hipMalloc(&out, 256 * 8 * 256 * 4);
    float hw[4] = {1.0001f, 0.9999f, 1.00005f, 0.f}; hipMemcpy(w, hw, 16, hipMemcpyHostToDevice);
    for (int wps : {1, 2, 4}) {
        run<0>("v_pk_fma vgpr,vgpr", w, out, wps);
        run<1>("v_pk_fma vgpr,sgpr-pair", w, out, wps);
        run<2>("v_pk_fma bcast(vgpr),sgpr-pair", w, out, wps);
        run<3>("v_pk_fma bcast(vgpr),vgpr", w, out, wps);
        run<4>("v_fmac sgpr,vgpr", w, out, wps);
        run<5>("v_fmac vgpr,vgpr", w, out, wps);
        run<6>("v_pk_mul vgpr,vgpr", w, out, wps);
    }
    return 0;
}

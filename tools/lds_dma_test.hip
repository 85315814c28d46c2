// Semantics of global_load_lds_dwordx4 on gfx950: lane L's 16 bytes land at LDS[base + imm_offset + 16*L]; the source
// address is per lane.  build: hipcc -O3 --offload-arch=gfx950 tools/lds_dma_test.hip -o tools/lds_dma_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef const void __attribute__((address_space(1)))* gptr;
typedef void __attribute__((address_space(3)))* lptr;
__global__ void k(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ perm) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 256 * 4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // wave w fills chunk w (1 KB); lane takes source quad perm[tid]
    __builtin_amdgcn_global_load_lds((gptr)(src + perm[tid] * 4), (lptr)(lds + wave * 256), 16, 0, 0);
    // second instruction: a different LDS base (M0), sources misaligned by one float (rows of 601 floats are not 16-byte aligned)
    __builtin_amdgcn_global_load_lds((gptr)(src + 1024 + 1 + perm[tid] * 4), (lptr)(lds + 1024 + wave * 256), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);     // vmcnt(0) (and everything else)
    __syncthreads();
    for (int i = tid; i < 2048; i += 256) dst[i] = lds[i];
}
int main() {
    std::vector<float> h(2064); for (int i = 0; i < 2064; ++i) h[i] = (float)i;
    std::vector<int> p(256); for (int i = 0; i < 256; ++i) p[i] = (i * 37) % 256;
    float *s, *d; int* dp;
    hipMalloc(&s, 8256); hipMalloc(&d, 8192); hipMalloc(&dp, 1024);
    hipMemcpy(s, h.data(), 8256, hipMemcpyHostToDevice); hipMemcpy(dp, p.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, s, d, dp);
    std::vector<float> o(2048); hipMemcpy(o.data(), d, 8192, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) for (int e = 0; e < 4; ++e) {
        if (o[t * 4 + e] != (float)(p[t] * 4 + e)) ++bad;
        if (o[1024 + t * 4 + e] != (float)(1024 + 1 + p[t] * 4 + e)) ++bad;
    }
    printf("lds dma layout: %s (%d mismatches); o[0..7] = %g %g %g %g %g %g %g %g\n", bad ? "UNEXPECTED" : "as expected", bad,
           o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
    return bad != 0;
}

// raw buffer loads on gfx950: what a 16-byte load returns when its byte offset (voffset, offen) lies outside [0, num_records) -- below zero
// (as an unsigned: beyond the range), straddling the end, beyond it.  hipcc --offload-arch=gfx950 tools/hip/buffer_oob_test.hip -o /tmp/t && /tmp/t
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* __restrict__ x, float* __restrict__ y, int nbytes, const int* __restrict__ offs) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, offs[threadIdx.x], 0, 0);
    *reinterpret_cast<f32x4*>(y + threadIdx.x * 4) = __builtin_bit_cast(f32x4, v);
}
int main() {
    const int n = 64;                       // floats in the buffer (the allocation is larger: what lies behind it is real memory)
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 1000.f + i;
    float *dx, *dy; int* doff;
    hipMalloc(&dx, 256 * 4); hipMalloc(&dy, 64 * 16); hipMalloc(&doff, 64 * 4);
    hipMemcpy(dx, h.data(), 256 * 4, hipMemcpyHostToDevice);
    int offs[64];
    for (int i = 0; i < 64; ++i) offs[i] = 0;
    offs[0] = 0; offs[1] = 16; offs[2] = -16; offs[3] = -4; offs[4] = (n - 4) * 4; offs[5] = (n - 2) * 4; offs[6] = n * 4; offs[7] = n * 4 + 64;
    offs[8] = (int)0x80000000u; offs[9] = -1024 * 1024;
    hipMemcpy(doff, offs, sizeof(offs), hipMemcpyHostToDevice);
    // the buffer starts 32 floats into the allocation so that negative offsets point at real memory
    k<<<1, 64>>>(dx + 32, dy, n * 4, doff);
    std::vector<float> out(256);
    hipMemcpy(out.data(), dy, 256 * 4, hipMemcpyDeviceToHost);
    const char* what[10] = {"0", "16", "-16", "-4", "last 16 bytes", "straddles the end", "at the end", "beyond", "0x80000000", "-1 MiB"};
    for (int i = 0; i < 10; ++i) printf("offset %-18s -> %g %g %g %g\n", what[i], out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]);
    return 0;
}

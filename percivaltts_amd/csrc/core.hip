// Error reporting and identification of libpercival_hip.so.
#include "common.h"
#include <cstring>

namespace ptts {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
__global__ void zero_f32_kernel_(float* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.f;
}
__global__ void zero_f32_2d_kernel_(float* __restrict__ p, size_t ld, size_t cols, size_t rows) {
    const size_t n = cols * rows;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[(i / cols) * ld + (i % cols)] = 0.f;
}
int zero_f32(float* p, size_t n, hipStream_t st) {
    if (n == 0) return PTTS_OK;
    size_t blocks = (n + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_f32_kernel_, dim3((unsigned)blocks), dim3(256), 0, st, p, n);
    return check_launch("zero_f32");
}
int zero_f32_2d(float* p, size_t ld, size_t cols, size_t rows, hipStream_t st) {
    if (ld == cols) return zero_f32(p, cols * rows, st);
    if (cols * rows == 0) return PTTS_OK;
    size_t blocks = (cols * rows + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_f32_2d_kernel_, dim3((unsigned)blocks), dim3(256), 0, st, p, ld, cols, rows);
    return check_launch("zero_f32_2d");
}
static int g_deterministic = 0;
bool deterministic() { return g_deterministic != 0; }
static int g_bf16_products = 0;
bool bf16_products() { return g_bf16_products != 0; }
}  // namespace ptts

extern "C" const char* ptts_version(void) { return "percival_hip 0.3.0 (round 3)"; }
extern "C" int ptts_set_bf16_products(int on) { const int old = ptts::g_bf16_products; ptts::g_bf16_products = on ? 1 : 0; return old; }
extern "C" int ptts_get_bf16_products(void) { return ptts::g_bf16_products; }
extern "C" int ptts_set_deterministic(int on) { const int old = ptts::g_deterministic; ptts::g_deterministic = on ? 1 : 0; return old; }
extern "C" int ptts_get_deterministic(void) { return ptts::g_deterministic; }
extern "C" const char* ptts_device_arch(void) { return "gfx950"; }
extern "C" const char* ptts_last_error(void) { return ptts::g_err; }

// Error reporting and identification of libpercival_hip.so.
#include "common.h"
#include <cstring>
#include <mutex>

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
// ---- device status words ------------------------------------------------------------------------------------------------------
static unsigned g_status_fallback[STATUS_SLOTS] = {0, 0};     // no HIP device (the CPU-only loads of tests/test_cabi.py): host logic only
static unsigned* volatile g_status_host = nullptr;
static unsigned* g_status_dev = nullptr;
static std::mutex g_status_mu;
unsigned* status_words() {
    if (!g_status_host) {
        std::lock_guard<std::mutex> lock(g_status_mu);
        if (!g_status_host) {
            void* h = nullptr; void* d = nullptr;
            int ndev = 0;
            if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 &&
                hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) == hipSuccess &&
                hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
                memset(h, 0, 64);
                g_status_dev = (unsigned*)d;
                g_status_host = (unsigned*)h;
            } else {
                (void)hipGetLastError();
                g_status_dev = g_status_fallback;
                g_status_host = g_status_fallback;
            }
        }
    }
    return g_status_dev;
}
static unsigned status_mask() {
    (void)status_words();
    unsigned m = 0;
    for (int i = 0; i < STATUS_SLOTS; ++i) m |= __atomic_load_n(g_status_host + i, __ATOMIC_RELAXED);
    return m;
}
static int status_message(unsigned m, char* buf, size_t n) {
    if (!buf || n == 0) return PTTS_EINVAL;
    if (m == 0) { snprintf(buf, n, "ok"); return PTTS_OK; }
    snprintf(buf, n, "device status 0x%x:%s%s%s -- the results of that launch (and of everything computed from it) are invalid",
             m, (m & STATUS_C2M_HANDOFF) ? " conv2d wave-specialised forward: an LDS hand-off count never arrived (bounded poll ran out);" : "",
             (m & STATUS_LSTM_HANDOFF) ? " persistent LSTM: a step's hidden state never arrived (bounded poll ran out);" : "",
             (m & ~(STATUS_C2M_HANDOFF | STATUS_LSTM_HANDOFF)) ? " unknown bits;" : "");
    return PTTS_OK;
}
int check_status(const char* what) {
    const unsigned m = status_mask();
    if (m == 0) return PTTS_OK;
    char msg[400];
    status_message(m, msg, sizeof(msg));
    set_error("%s: %s", what, msg);
    return PTTS_EDEVICE;
}
static int g_deterministic = 0;
bool deterministic() { return g_deterministic != 0; }
static int g_bf16_products = 0;
bool bf16_products() { return g_bf16_products != 0; }
}  // namespace ptts

extern "C" const char* ptts_version(void) { return "percival_hip 0.4.0 (round 4)"; }
extern "C" int ptts_set_bf16_products(int on) { const int old = ptts::g_bf16_products; ptts::g_bf16_products = on ? 1 : 0; return old; }
extern "C" int ptts_get_bf16_products(void) { return ptts::g_bf16_products; }
extern "C" int ptts_set_deterministic(int on) { const int old = ptts::g_deterministic; ptts::g_deterministic = on ? 1 : 0; return old; }
extern "C" int ptts_get_deterministic(void) { return ptts::g_deterministic; }
extern "C" const char* ptts_device_arch(void) { return "gfx950"; }
extern "C" const char* ptts_last_error(void) { return ptts::g_err; }
extern "C" int ptts_device_status(unsigned* word_out) {
    if (word_out) *word_out = ptts::status_mask();
    return ptts::check_status("ptts_device_status");
}
extern "C" int ptts_device_status_clear(void) {
    (void)ptts::status_words();
    for (int i = 0; i < ptts::STATUS_SLOTS; ++i) __atomic_store_n(ptts::g_status_host + i, 0u, __ATOMIC_RELAXED);
    return PTTS_OK;
}
extern "C" unsigned* ptts_device_status_word(void) { (void)ptts::status_words(); return ptts::g_status_host; }
extern "C" int ptts_device_status_message(unsigned word, char* buf, size_t n) { return ptts::status_message(word, buf, n); }

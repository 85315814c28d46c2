// Shared helpers for the gfx950 kernels of libpercival_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include "../../include/percival_hip.h"

namespace ptts {

void set_error(const char* fmt, ...);
// ptts_set_deterministic(): fixed-order reductions only (no fp32 atomics between workgroups), at a price in speed
bool deterministic();
// ptts_set_bf16_products(): the split GEMM kernels (context Conv1D, Dense, their weight gradients) form ONE product of the
// operands' bf16 roundings instead of the six products of the fp32 split (BASELINE configs[2]: bf16 products, fp32 accumulation)
bool bf16_products();
// Zero fills by kernel, not hipMemsetAsync: memset nodes of a captured hipGraph were found not to be replayed reliably on
// this stack (tests/test_model_gpu.py::test_split_hipgraph_...: bias-gradient and loss scalars came back unzeroed).
int zero_f32(float* p, size_t n, hipStream_t st);
int zero_f32_2d(float* p, size_t ld, size_t cols, size_t rows, hipStream_t st);

// Device status words (core.hip).  Two kernels let waves wait for each other with BOUNDED polls (the LDS-counter hand-off of
// c2m::fwd_ws_kernel, the granule hand-off of lstm_fwd_persistent_kernel): a poll that runs out means the results of that launch are
// wrong.  The kernel then stores its code into its slot of a small block of pinned, device-mapped host memory (a plain system-scope
// store: sticky, because nothing on the device ever clears it), and every later C-ABI call of the same family -- and
// ptts_device_status(), which the Python layer calls at each step boundary -- returns PTTS_EDEVICE until ptts_device_status_clear().
// Reading the words costs the host a memory load, no synchronisation.
constexpr int STATUS_SLOT_C2M = 0, STATUS_SLOT_LSTM = 1, STATUS_SLOTS = 2;
constexpr unsigned STATUS_C2M_HANDOFF = 0x1u, STATUS_LSTM_HANDOFF = 0x2u;
unsigned* status_words();                 // device-visible pointer to STATUS_SLOTS words; never null
int check_status(const char* what);       // PTTS_OK, or PTTS_EDEVICE with the message set
__device__ __forceinline__ void raise_status(unsigned* words, int slot, unsigned code) {
    __hip_atomic_store(words + slot, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return PTTS_ELAUNCH;
    }
    return PTTS_OK;
}

#define PTTS_REQUIRE(cond, ...)                         \
    do {                                                \
        if (!(cond)) {                                  \
            ptts::set_error(__VA_ARGS__);               \
            return PTTS_EINVAL;                         \
        }                                               \
    } while (0)

constexpr int WAVE = 64;

// thin.hip: returns 1 when it handled the product, 0 to fall through to the MFMA kernels, <0 on error
int thin_gemm_dispatch(const float* A, const float* Bm, const float* bias, float* C, int M, int N, int K, int transA,
                       long long lda, long long rows_per_seg, long long seg_stride, int transB, long long ldb,
                       long long ldc, int in_mode, const float* in_scale, const float* in_shift,
                       const float* mask_src, float alpha, int accumulate, const float* out_mask, hipStream_t st);

__device__ __forceinline__ float lrelu(float p, float alpha) { return p > 0.f ? p : alpha * p; }
__device__ __forceinline__ float lrelu_d(float p, float alpha) { return p > 0.f ? 1.f : alpha; }

// wave64 butterfly sums
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace ptts

// Thin products that a 128-wide MFMA tile would waste: the critic's Dense(1) head, the generator's f0 head and the
// 4-column remainder of the 260-wide spectral part (reference networks_critic.py:89-96, modeltts_common.py:84).
//   gemv  : C[M, N<=4]  = T(A)[M,K] . B          (one wave per row, lanes along K, butterfly reduce)
//   thin-K: C[M, N]     = A[M, K<=4] . B  (* mask)   (outer-product-like, streaming write of C)
//   wcol  : C[Mo, N<=2] = T(A)[Kr, Mo]^T . B[Kr, N]   (column sums of A weighted by 1-2 columns of B; fp32 atomics)
// All are HBM-bound single passes over the big operand.
#include "common.h"
#include <mutex>
#include <vector>
#include <utility>

namespace ptts {

struct ThinArgs {
    const float* A; const float* B; const float* bias; float* C;
    int M, N, K;
    long long lda, ldb, ldc;
    int transB;
    int in_mode; const float* in_scale; const float* in_shift; const float* mask_src; float alpha;
    int accumulate; const float* out_mask;
};

__device__ __forceinline__ float thin_xform(float v, const ThinArgs& g, long long off, int ch) {
    if (g.in_mode == PTTS_IN_LRELU) {
        if (g.in_scale) v = v * g.in_scale[ch] + g.in_shift[ch];
        return lrelu(v, g.alpha);
    } else if (g.in_mode == PTTS_IN_MASKMUL) {
        return v * lrelu_d(g.mask_src[off], g.alpha);
    }
    return v;
}

__device__ __forceinline__ float4 thin_xform4(float4 v, float4 m, const ThinArgs& g, int ch) {
    float a[4] = {v.x, v.y, v.z, v.w};
    const float mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (g.in_mode == PTTS_IN_LRELU) {
            if (g.in_scale) a[e] = a[e] * g.in_scale[ch + e] + g.in_shift[ch + e];
            a[e] = lrelu(a[e], g.alpha);
        } else if (g.in_mode == PTTS_IN_MASKMUL) {
            a[e] *= lrelu_d(mm[e], g.alpha);
        }
    }
    return make_float4(a[0], a[1], a[2], a[3]);
}

// one wave per row; A row-major with leading dim lda (transA = 0).  VEC: K % 4 == 0, K <= 256 and 16-byte aligned
// rows -- one float4 of A per lane and row, B held in registers (NT x 4 values per lane), two rows in flight.
template <int NT, bool VEC>
__global__ __launch_bounds__(256) void gemv_rows_kernel(ThinArgs g) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = gridDim.x * 4;
    if (VEC) {
        const int k = lane * 4;
        const bool on = k < g.K;
        float w[NT][4];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                w[n][e] = (on && n < g.N) ? (g.transB ? g.B[(long long)n * g.ldb + k + e] : g.B[(long long)(k + e) * g.ldb + n]) : 0.f;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int m = wave; m < g.M; m += 2 * nwaves) {
            const int m2 = m + nwaves;
            const bool two = m2 < g.M;
            const long long r1 = (long long)m * g.lda + k, r2 = (long long)(two ? m2 : m) * g.lda + k;
            float4 a1 = on ? *reinterpret_cast<const float4*>(g.A + r1) : z4;
            float4 a2 = on ? *reinterpret_cast<const float4*>(g.A + r2) : z4;
            float4 k1 = z4, k2 = z4;
            if (g.in_mode == PTTS_IN_MASKMUL && on) {
                k1 = *reinterpret_cast<const float4*>(g.mask_src + r1);
                k2 = *reinterpret_cast<const float4*>(g.mask_src + r2);
            }
            if (on) { a1 = thin_xform4(a1, k1, g, k); a2 = thin_xform4(a2, k2, g, k); }
            float acc1[NT], acc2[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                acc1[n] = a1.x * w[n][0] + a1.y * w[n][1] + a1.z * w[n][2] + a1.w * w[n][3];
                acc2[n] = a2.x * w[n][0] + a2.y * w[n][1] + a2.z * w[n][2] + a2.w * w[n][3];
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) { acc1[n] = wave_sum(acc1[n]); acc2[n] = wave_sum(acc2[n]); }
            if (lane == 0) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 1 && !two) break;
                    const int mm = h ? m2 : m;
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        if (n >= g.N) continue;
                        const long long off = (long long)mm * g.ldc + n;
                        float v = (h ? acc2[n] : acc1[n]) + (g.bias ? g.bias[n] : 0.f);
                        if (g.out_mask) v *= lrelu_d(g.out_mask[off], g.alpha);
                        if (g.accumulate) g.C[off] += v; else g.C[off] = v;
                    }
                }
            }
        }
        return;
    }
    for (int m = wave; m < g.M; m += nwaves) {
        float acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = 0.f;
        const long long row = (long long)m * g.lda;
        for (int k = lane; k < g.K; k += 64) {
            const float a = thin_xform(g.A[row + k], g, row + k, k);
#pragma unroll
            for (int n = 0; n < NT; ++n)
                if (n < g.N) acc[n] = fmaf(a, g.transB ? g.B[(long long)n * g.ldb + k] : g.B[(long long)k * g.ldb + n], acc[n]);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = wave_sum(acc[n]);
        if (lane == 0) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                if (n >= g.N) continue;
                const long long off = (long long)m * g.ldc + n;
                float v = acc[n] + (g.bias ? g.bias[n] : 0.f);
                if (g.out_mask) v *= lrelu_d(g.out_mask[off], g.alpha);
                if (g.accumulate) g.C[off] += v; else g.C[off] = v;
            }
        }
    }
}

// K <= 4: C[m, :] = sum_k A[m,k] * B(k, :).  VEC: N % 4 == 0 and 16-byte aligned C / mask rows -- a float4 of C per lane.
template <int KT, bool VEC>
__global__ __launch_bounds__(256) void thin_k_kernel(ThinArgs g) {
    if (VEC) {
        const int n4 = g.N / 4;
        const long long total = (long long)g.M * n4;
        for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const int m = (int)(i / n4);
            const int n = (int)(i - (long long)m * n4) * 4;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = g.bias ? g.bias[n + e] : 0.f;
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                if (k >= g.K) continue;
                const long long offa = (long long)m * g.lda + k;
                const float a = thin_xform(g.A[offa], g, offa, k);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = fmaf(a, g.transB ? g.B[(long long)(n + e) * g.ldb + k] : g.B[(long long)k * g.ldb + n + e], v[e]);
            }
            const long long off = (long long)m * g.ldc + n;
            if (g.out_mask) {
                const float4 mk = *reinterpret_cast<const float4*>(g.out_mask + off);
                v[0] *= lrelu_d(mk.x, g.alpha); v[1] *= lrelu_d(mk.y, g.alpha);
                v[2] *= lrelu_d(mk.z, g.alpha); v[3] *= lrelu_d(mk.w, g.alpha);
            }
            float4* cp = reinterpret_cast<float4*>(g.C + off);
            if (g.accumulate) { const float4 o = *cp; v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
            *cp = make_float4(v[0], v[1], v[2], v[3]);
        }
        return;
    }
    const long long total = (long long)g.M * g.N;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int n = (int)(i % g.N);
        const long long m = i / g.N;
        float v = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            if (k >= g.K) continue;
            const long long offa = m * g.lda + k;
            const float a = thin_xform(g.A[offa], g, offa, k);
            v = fmaf(a, g.transB ? g.B[(long long)n * g.ldb + k] : g.B[(long long)k * g.ldb + n], v);
        }
        const long long off = m * g.ldc + n;
        if (g.out_mask) v *= lrelu_d(g.out_mask[off], g.alpha);
        if (g.accumulate) g.C[off] += v; else g.C[off] = v;
    }
}

// C[i, n] = sum_r T(A[r, i]) * B[r, n],  n < N <= 2,  A stored [Kr rows][Mo cols] (the transA = 1 operand).
// Lanes along the columns i (coalesced rows), 4 row groups per workgroup, LDS combine, one fp32 atomic per (i, n).
// (A float4-per-lane variant with batched loads was slower: the contended atomics on the few output addresses, not the
// row stream, set the time of this kernel.)
template <int NT>
__global__ __launch_bounds__(256) void wcol_kernel(ThinArgs g, int rows) {
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    __shared__ float sh[NT][4][64];
    const int i = blockIdx.x * 64 + lane;
    float acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = 0.f;
    if (i < g.M) {
        for (long long r = (long long)blockIdx.y * 4 + rg; r < rows; r += (long long)gridDim.y * 4) {
            const long long offa = r * g.lda + i;
            const float a = thin_xform(g.A[offa], g, offa, i);
#pragma unroll
            for (int n = 0; n < NT; ++n)
                if (n < g.N) acc[n] = fmaf(a, g.B[r * g.ldb + n], acc[n]);
        }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) sh[n][rg][lane] = acc[n];
    __syncthreads();
    if (rg == 0 && i < g.M) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if (n >= g.N) continue;
            atomicAdd(g.C + (long long)i * g.ldc + n, sh[n][0][lane] + sh[n][1][lane] + sh[n][2][lane] + sh[n][3][lane]);
        }
    }
}

// The same product in TWO stages without atomics (round 4): every workgroup of wcol_part_kernel leaves its share of the rows as one
// partial row part[row block][Mo][NT], wcol_reduce_kernel adds the row blocks in index order.  With atomics the kernel could not
// have both enough row blocks to hide the loads' latency (2048 workgroups: 512 per output address) and few enough atomics on its
// 256 output addresses -- 44 us for the [25 600, 256] activation of the critic's 1-wide head, 0.6 TB/s.  Deterministic as well.
template <int NT>
__global__ __launch_bounds__(256) void wcol_part_kernel(ThinArgs g, int rows, float* __restrict__ part) {
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    __shared__ float sh[NT][4][64];
    const int i = blockIdx.x * 64 + lane;
    float acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = 0.f;
    if (i < g.M) {
        const long long step = (long long)gridDim.y * 4;
        long long r = (long long)blockIdx.y * 4 + rg;
        for (; r + 3 * step < rows; r += 4 * step) {
            float a[4], b[4][NT];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long rr = r + u * step, offa = rr * g.lda + i;
                a[u] = thin_xform(g.A[offa], g, offa, i);
#pragma unroll
                for (int n = 0; n < NT; ++n) b[u][n] = n < g.N ? g.B[rr * g.ldb + n] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = fmaf(a[u], b[u][n], acc[n]);
        }
        for (; r < rows; r += step) {
            const long long offa = r * g.lda + i;
            const float a = thin_xform(g.A[offa], g, offa, i);
#pragma unroll
            for (int n = 0; n < NT; ++n)
                if (n < g.N) acc[n] = fmaf(a, g.B[r * g.ldb + n], acc[n]);
        }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) sh[n][rg][lane] = acc[n];
    __syncthreads();
    if (rg == 0 && i < g.M) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
            part[((size_t)blockIdx.y * g.M + i) * NT + n] = (sh[n][0][lane] + sh[n][1][lane]) + (sh[n][2][lane] + sh[n][3][lane]);
    }
}
// ... with FOUR columns per lane (16-byte loads; a wave covers 256 columns of a row, the workgroup's four waves four row phases):
// Mo % 4 == 0, lda % 4 == 0, A and the mask source 16-byte aligned
template <int NT>
__global__ __launch_bounds__(256) void wcol_part4_kernel(ThinArgs g, int rows, float* __restrict__ part) {
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    __shared__ float sh[NT][4][256];
    const int i = blockIdx.x * 256 + 4 * lane;
    float acc[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[n][e] = 0.f;
    if (i < g.M) {
        const long long step = (long long)gridDim.y * 4;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        auto one = [&](long long rr) {
            const long long offa = rr * g.lda + i;
            const float4 a = *reinterpret_cast<const float4*>(g.A + offa);
            const float4 m = g.in_mode == PTTS_IN_MASKMUL ? *reinterpret_cast<const float4*>(g.mask_src + offa) : z4;
            const float4 t = thin_xform4(a, m, g, i);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float b = n < g.N ? g.B[rr * g.ldb + n] : 0.f;
                acc[n][0] = fmaf(t.x, b, acc[n][0]); acc[n][1] = fmaf(t.y, b, acc[n][1]);
                acc[n][2] = fmaf(t.z, b, acc[n][2]); acc[n][3] = fmaf(t.w, b, acc[n][3]);
            }
        };
        long long r = (long long)blockIdx.y * 4 + rg;
        for (; r + 3 * step < rows; r += 4 * step) { one(r); one(r + step); one(r + 2 * step); one(r + 3 * step); }
        for (; r < rows; r += step) one(r);
    }
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) sh[n][rg][4 * lane + e] = acc[n][e];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < g.M) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
            part[((size_t)blockIdx.y * g.M + c) * NT + n] = (sh[n][0][threadIdx.x] + sh[n][1][threadIdx.x]) + (sh[n][2][threadIdx.x] + sh[n][3][threadIdx.x]);
    }
}
// C[i, n] (+)= sum over the row blocks of part[block][i][n]: a workgroup per 16 outputs, 256 lanes = 16 outputs x 16 row phases, four
// loads in flight per lane (64 outputs x 4 phases left every lane a chain of 128 dependent L2 loads: 17 us for 512 KB)
template <int NT>
__global__ __launch_bounds__(256) void wcol_reduce_kernel(const float* __restrict__ part, int nblocks, int Mo, int N, float* __restrict__ C,
                                                          long long ldc, int accumulate) {
    __shared__ float sh[16][16];
    const int oo = threadIdx.x & 15, ph = threadIdx.x >> 4;
    const int o = blockIdx.x * 16 + oo;
    const int total = Mo * NT;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (o < total) {
        int b = ph;
        for (; b + 48 < nblocks; b += 64) {
            const float v0 = part[(size_t)b * total + o], v1 = part[(size_t)(b + 16) * total + o];
            const float v2 = part[(size_t)(b + 32) * total + o], v3 = part[(size_t)(b + 48) * total + o];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; b < nblocks; b += 16) s0 += part[(size_t)b * total + o];
    }
    sh[ph][oo] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ph == 0 && o < total) {
        const int i = o / NT, n = o - i * NT;
        if (n < N) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) v += sh[k][oo];
            float* c = C + (long long)i * ldc + n;
            *c = accumulate ? *c + v : v;
        }
    }
}
// the partial rows' buffer, one per stream (4 MB, allocated at the first call on that stream -- outside a capture: the steps' warm-up
// runs come first; inside one the allocation is refused and the caller falls back to the atomics kernel)
constexpr size_t WCOL_WS_BYTES = 4u << 20;
static float* wcol_workspace(hipStream_t st) {
    static std::mutex mu;
    static std::vector<std::pair<hipStream_t, float*>> bufs;
    std::lock_guard<std::mutex> lk(mu);
    for (auto& e : bufs)
        if (e.first == st) return e.second;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    float* p = nullptr;
    if (hipMalloc(&p, WCOL_WS_BYTES) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    bufs.emplace_back(st, p);
    return p;
}

static inline bool thin_al16(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// returns 1 if the product was handled here, 0 if the caller should use the MFMA kernels, <0 on error
int thin_gemm_dispatch(const float* A, const float* Bm, const float* bias, float* C, int M, int N, int K, int transA,
                       long long lda, long long rows_per_seg, long long seg_stride, int transB, long long ldb,
                       long long ldc, int in_mode, const float* in_scale, const float* in_shift,
                       const float* mask_src, float alpha, int accumulate, const float* out_mask, hipStream_t st) {
    if (seg_stride != 0) return 0;
    ThinArgs g{A, Bm, bias, C, M, N, K, lda, ldb, ldc, transB, in_mode, in_scale, in_shift, mask_src, alpha, accumulate, out_mask};
    if (transA == 0 && N <= 4 && K >= 16 && M >= 256) {
        int blocks = (M + 3) / 4;
        if (blocks > 2048) blocks = 2048;
        const bool vec = K % 4 == 0 && K <= 256 && lda % 4 == 0 && thin_al16(A) && thin_al16(mask_src);
#define PTTS_GEMV(NTT) do { if (vec) hipLaunchKernelGGL((gemv_rows_kernel<NTT, true>), dim3(blocks), dim3(256), 0, st, g); \
                            else hipLaunchKernelGGL((gemv_rows_kernel<NTT, false>), dim3(blocks), dim3(256), 0, st, g); } while (0)
        if (N == 1) PTTS_GEMV(1);
        else if (N == 2) PTTS_GEMV(2);
        else PTTS_GEMV(4);
#undef PTTS_GEMV
        int rc = check_launch("gemv_rows");
        return rc ? rc : 1;
    }
    if (transA == 0 && K <= 4 && (long long)M * N >= 65536) {
        long long b = ((long long)M * N + 1023) / 1024;
        if (b > 2048) b = 2048;
        const bool vec = N % 4 == 0 && ldc % 4 == 0 && thin_al16(C) && thin_al16(out_mask);
#define PTTS_THINK(KTT) do { if (vec) hipLaunchKernelGGL((thin_k_kernel<KTT, true>), dim3((int)b), dim3(256), 0, st, g); \
                             else hipLaunchKernelGGL((thin_k_kernel<KTT, false>), dim3((int)b), dim3(256), 0, st, g); } while (0)
        if (K == 1) PTTS_THINK(1);
        else if (K == 2) PTTS_THINK(2);
        else PTTS_THINK(4);
#undef PTTS_THINK
        int rc = check_launch("thin_k");
        return rc ? rc : 1;
    }
    if (transA == 1 && transB == 0 && N <= 2 && K >= 1024 && !bias && !out_mask && rows_per_seg >= K) {
        // two stages, no atomics: enough row blocks to hide the loads, as long as the partial rows fit the stream's buffer
        const bool vec4 = M % 4 == 0 && lda % 4 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (!mask_src || (reinterpret_cast<uintptr_t>(mask_src) & 15) == 0);
        const int cbp = vec4 ? (M + 255) / 256 : (M + 63) / 64;
        int rbp = (vec4 ? 512 : 2048) / cbp;
        if (rbp > 512) rbp = 512;
        if (rbp > (K + 15) / 16) rbp = (K + 15) / 16;              // (at least four rows per lane)
        if (rbp < 1) rbp = 1;
        float* part = ((size_t)rbp * M * N * sizeof(float) <= WCOL_WS_BYTES) ? wcol_workspace(st) : nullptr;
        if (part) {
            if (vec4 && N == 1) {
                hipLaunchKernelGGL(wcol_part4_kernel<1>, dim3(cbp, rbp), dim3(256), 0, st, g, K, part);
                hipLaunchKernelGGL(wcol_reduce_kernel<1>, dim3((M + 15) / 16), dim3(256), 0, st, part, rbp, M, N, C, ldc, accumulate);
            } else if (vec4) {
                hipLaunchKernelGGL(wcol_part4_kernel<2>, dim3(cbp, rbp), dim3(256), 0, st, g, K, part);
                hipLaunchKernelGGL(wcol_reduce_kernel<2>, dim3((2 * M + 15) / 16), dim3(256), 0, st, part, rbp, M, N, C, ldc, accumulate);
            } else if (N == 1) {
                hipLaunchKernelGGL(wcol_part_kernel<1>, dim3(cbp, rbp), dim3(256), 0, st, g, K, part);
                hipLaunchKernelGGL(wcol_reduce_kernel<1>, dim3((M + 15) / 16), dim3(256), 0, st, part, rbp, M, N, C, ldc, accumulate);
            } else {
                hipLaunchKernelGGL(wcol_part_kernel<2>, dim3(cbp, rbp), dim3(256), 0, st, g, K, part);
                hipLaunchKernelGGL(wcol_reduce_kernel<2>, dim3((2 * M + 15) / 16), dim3(256), 0, st, part, rbp, M, N, C, ldc, accumulate);
            }
            int rcp = check_launch("wcol (two stages)");
            return rcp ? rcp : 1;
        }
        if (deterministic()) return 0;
        if (!accumulate) {
            if (zero_f32_2d(C, (size_t)ldc, (size_t)N, (size_t)M, st) != PTTS_OK) return PTTS_ELAUNCH;
        }
        const int cb = (M + 63) / 64;
        int rb = 1024 / cb;
        if (rb < 1) rb = 1;
        if (rb > 256) rb = 256;
        if (N == 1) hipLaunchKernelGGL(wcol_kernel<1>, dim3(cb, rb), dim3(256), 0, st, g, K);
        else hipLaunchKernelGGL(wcol_kernel<2>, dim3(cb, rb), dim3(256), 0, st, g, K);
        int rc = check_launch("wcol");
        return rc ? rc : 1;
    }
    return 0;
}

}  // namespace ptts

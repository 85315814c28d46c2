// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, one rounding per
// product, same rate as the vector ALU but on its own pipe and with 1 operand VGPR per lane).
//
// Role on the hot path: the "dense label->hidden projections" of the north star -- the context
// Conv1D(k=21, ctx->256) of both networks (reference networktts.py:116-120 via
// networks_critic.py:82-83 and modeltts_common.py:75-76) run as an IMPLICIT GEMM over a
// zero-padded frame buffer (K = 21*ctx, row stride = ctx: the im2col matrix is never built),
// plus every Dense layer (networktts.py:59-63) and the LSTM input/weight-gradient products.
//
// Tiling: 128x128x16 per 256-thread workgroup, 2x2 waves, each wave 2x2 MFMA 32x32 tiles
// (64 accumulator VGPRs).  Global->register prefetch of tile k+1 overlaps the MFMAs of tile k;
// LDS holds A as [k][m] and B as [k][n] so every fragment read is a conflict-free ds_read_b32.
// The previous layer's BatchNorm-affine/LeakyReLU (or the gradient-penalty mask) is applied
// while A is staged.  Small-MN/large-K products (weight gradients) are split along K across
// workgroups and combined with fp32 atomics into a zeroed C.
#include "common.h"

namespace ptts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDA_S = BM + 4;   // LDS leading dims (multiple of 4 floats: 16-B aligned rows)
constexpr int LDB_S = BN + 4;
constexpr int GEMM_THREADS = 256;

struct GemmArgs {
    const float* A; const float* B; const float* bias; float* C;
    int M, N, K;
    int transA; long long lda, rows_per_seg, seg_stride;
    int transB; long long ldb, ldc;
    int in_mode; const float* in_scale; const float* in_shift; const float* mask_src; float alpha;
    int accumulate; int splits; int k_per_split;   // k_per_split is a multiple of BK
};

__device__ __forceinline__ float a_transform(float v, const GemmArgs& g, long long off, int ch) {
    if (g.in_mode == PTTS_IN_LRELU) {
        if (g.in_scale) v = v * g.in_scale[ch] + g.in_shift[ch];
        return lrelu(v, g.alpha);
    } else if (g.in_mode == PTTS_IN_MASKMUL) {
        return v * lrelu_d(g.mask_src[off], g.alpha);
    }
    return v;
}

// Each thread stages 2 float4 of A and 2 float4 of B per k-step.
struct Frag { float a[2][4]; float b[2][4]; };

template <int TRANSA, int TRANSB>
__device__ __forceinline__ void load_tiles(const GemmArgs& g, int m0, int n0, int k0, int kend, Frag& fr) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = tid + j * GEMM_THREADS;   // 0..511
        if (TRANSA == 0) {
            // tile [128 m][16 k], float4 along k: q -> (m = q/4, kq = q%4)
            const int m = m0 + (q >> 2), k = k0 + (q & 3) * 4;
            if (m < g.M) {
                const long long base = (m / g.rows_per_seg) * g.seg_stride + (m % g.rows_per_seg) * g.lda;
                if (k + 3 < kend) {
                    const f32x4u v = *reinterpret_cast<const f32x4u*>(g.A + base + k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) fr.a[j][e] = a_transform(v[e], g, base + k + e, k + e);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        fr.a[j][e] = (k + e < kend) ? a_transform(g.A[base + k + e], g, base + k + e, k + e) : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) fr.a[j][e] = 0.f;
            }
        } else {
            // stored [k][m] (m contiguous): q -> (k = q/32, mq = q%32)
            const int k = k0 + (q >> 5), m = m0 + (q & 31) * 4;
            if (k < kend) {
                const long long base = (k / g.rows_per_seg) * g.seg_stride + (k % g.rows_per_seg) * g.lda;
                if (m + 3 < g.M) {
                    const f32x4u v = *reinterpret_cast<const f32x4u*>(g.A + base + m);
#pragma unroll
                    for (int e = 0; e < 4; ++e) fr.a[j][e] = a_transform(v[e], g, base + m + e, m + e);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        fr.a[j][e] = (m + e < g.M) ? a_transform(g.A[base + m + e], g, base + m + e, m + e) : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) fr.a[j][e] = 0.f;
            }
        }
        if (TRANSB == 0) {
            // stored [k][n] (n contiguous): q -> (k = q/32, nq = q%32)
            const int k = k0 + (q >> 5), n = n0 + (q & 31) * 4;
            if (k < kend) {
                const long long base = (long long)k * g.ldb;
                if (n + 3 < g.N) {
                    const f32x4u v = *reinterpret_cast<const f32x4u*>(g.B + base + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) fr.b[j][e] = v[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) fr.b[j][e] = (n + e < g.N) ? g.B[base + n + e] : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) fr.b[j][e] = 0.f;
            }
        } else {
            // stored [n][k] (k contiguous): q -> (n = q/4, kq = q%4)
            const int n = n0 + (q >> 2), k = k0 + (q & 3) * 4;
            if (n < g.N) {
                const long long base = (long long)n * g.ldb;
                if (k + 3 < kend) {
                    const f32x4u v = *reinterpret_cast<const f32x4u*>(g.B + base + k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) fr.b[j][e] = v[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) fr.b[j][e] = (k + e < kend) ? g.B[base + k + e] : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) fr.b[j][e] = 0.f;
            }
        }
    }
}

// Interior tiles: no bounds checks, transform selected at compile time, four 16-byte loads in flight per lane.
template <int TRANSA, int TRANSB, int MODE>
__device__ __forceinline__ void load_tiles_fast(const GemmArgs& g, int m0, int n0, int k0, Frag& fr) {
    const int tid = threadIdx.x;
    f32x4u va[2], vb[2], vm[2];
    long long offa[2];
    int cha[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = tid + j * GEMM_THREADS;
        if (TRANSA == 0) {
            const int m = m0 + (q >> 2), k = k0 + (q & 3) * 4;
            offa[j] = (m / g.rows_per_seg) * g.seg_stride + (m % g.rows_per_seg) * g.lda + k;
            cha[j] = k;
        } else {
            const int k = k0 + (q >> 5), m = m0 + (q & 31) * 4;
            offa[j] = (k / g.rows_per_seg) * g.seg_stride + (k % g.rows_per_seg) * g.lda + m;
            cha[j] = m;
        }
        va[j] = *reinterpret_cast<const f32x4u*>(g.A + offa[j]);
        if (MODE == PTTS_IN_MASKMUL) vm[j] = *reinterpret_cast<const f32x4u*>(g.mask_src + offa[j]);
        if (TRANSB == 0) {
            const int k = k0 + (q >> 5), n = n0 + (q & 31) * 4;
            vb[j] = *reinterpret_cast<const f32x4u*>(g.B + (long long)k * g.ldb + n);
        } else {
            const int n = n0 + (q >> 2), k = k0 + (q & 3) * 4;
            vb[j] = *reinterpret_cast<const f32x4u*>(g.B + (long long)n * g.ldb + k);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = va[j][e];
            if (MODE == PTTS_IN_LRELU) {
                if (g.in_scale) v = v * g.in_scale[cha[j] + e] + g.in_shift[cha[j] + e];
                v = lrelu(v, g.alpha);
            } else if (MODE == PTTS_IN_MASKMUL) {
                v *= lrelu_d(vm[j][e], g.alpha);
            }
            fr.a[j][e] = v;
            fr.b[j][e] = vb[j][e];
        }
    }
}

template <int TRANSA, int TRANSB>
__device__ __forceinline__ void store_tiles(float* As, float* Bs, const Frag& fr) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = tid + j * GEMM_THREADS;
        if (TRANSA == 0) {
            const int m = q >> 2, k = (q & 3) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) As[(k + e) * LDA_S + m] = fr.a[j][e];
        } else {
            const int k = q >> 5, m = (q & 31) * 4;
            *reinterpret_cast<float4*>(As + k * LDA_S + m) = make_float4(fr.a[j][0], fr.a[j][1], fr.a[j][2], fr.a[j][3]);
        }
        if (TRANSB == 0) {
            const int k = q >> 5, n = (q & 31) * 4;
            *reinterpret_cast<float4*>(Bs + k * LDB_S + n) = make_float4(fr.b[j][0], fr.b[j][1], fr.b[j][2], fr.b[j][3]);
        } else {
            const int n = q >> 2, k = (q & 3) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) Bs[(k + e) * LDB_S + n] = fr.b[j][e];
        }
    }
}

// CONV != 0 marks the implicit-convolution instantiation (same code; its own symbol so that a profile separates the
// context-Conv1D products from the small Dense ones).
// LDS is double-buffered: tile k+1 is written to the other buffer while tile k feeds the MFMAs -> one barrier per
// k-step; the fragments of k-pair kk+1 are read from LDS while the four MFMAs of k-pair kk execute.
template <int TRANSA, int TRANSB, int CONV, int MODE>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_f32_mfma_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][BK * LDA_S];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB_S];
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int l31 = lane & 31, lh = lane >> 5;
    const bool interior = (m0 + BM <= g.M) && (n0 + BN <= g.N);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Frag fr;
    auto load = [&](int k0) {
        if (interior && k0 + BK <= kend) load_tiles_fast<TRANSA, TRANSB, MODE>(g, m0, n0, k0, fr);
        else load_tiles<TRANSA, TRANSB>(g, m0, n0, k0, kend, fr);
    };
    if (kbeg < kend) {
        load(kbeg);
        store_tiles<TRANSA, TRANSB>(As[0], Bs[0], fr);
    }
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK, buf ^= 1) {
        const bool more = k0 + BK < kend;
        if (more) load(k0 + BK);                 // global -> registers, in flight during the MFMAs below
        const float* as = As[buf] + lh * LDA_S + wm + l31;
        const float* bs = Bs[buf] + lh * LDB_S + wn + l31;
        float ra[2][2], rb[2][2];
        ra[0][0] = as[0]; ra[0][1] = as[32]; rb[0][0] = bs[0]; rb[0][1] = bs[32];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int c = (kk >> 1) & 1, nx = c ^ 1;
            if (kk + 2 < BK) {
                ra[nx][0] = as[(kk + 2) * LDA_S];
                ra[nx][1] = as[(kk + 2) * LDA_S + 32];
                rb[nx][0] = bs[(kk + 2) * LDB_S];
                rb[nx][1] = bs[(kk + 2) * LDB_S + 32];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[c][0], rb[c][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[c][0], rb[c][1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[c][1], rb[c][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[c][1], rb[c][1], acc[1][1], 0, 0, 0);
        }
        if (more) store_tiles<TRANSA, TRANSB>(As[buf ^ 1], Bs[buf ^ 1], fr);
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool add_bias = g.bias != nullptr && blockIdx.z == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + j * 32 + l31;
            if (n >= g.N) continue;
            const float bv = add_bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= g.M) continue;
                float* cp = g.C + (long long)m * g.ldc + n;
                const float v = acc[i][j][r] + bv;
                if (g.splits > 1) atomicAdd(cp, v);
                else if (g.accumulate) *cp += v;
                else *cp = v;
            }
        }
}

}  // namespace ptts

using namespace ptts;

extern "C" int ptts_gemm(const float* A, const float* Bm, const float* bias, float* C, int M, int N, int K,
                         int transA, long long lda, long long rows_per_seg, long long seg_stride, int transB,
                         long long ldb, long long ldc, int in_mode, const float* in_scale,
                         const float* in_shift, const float* mask_src, float alpha, int accumulate,
                         void* stream) {
    PTTS_REQUIRE(A && Bm && C, "gemm: null matrix");
    PTTS_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: bad dims M=%d N=%d K=%d", M, N, K);
    PTTS_REQUIRE(rows_per_seg > 0 && lda > 0 && ldb > 0 && ldc >= N, "gemm: bad leading dims");
    PTTS_REQUIRE(in_mode >= 0 && in_mode <= 2, "gemm: bad in_mode %d", in_mode);
    PTTS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "gemm: scale/shift must come together");
    PTTS_REQUIRE(in_mode != PTTS_IN_MASKMUL || mask_src, "gemm: MASKMUL needs mask_src");
    hipStream_t st = (hipStream_t)stream;
    GemmArgs g;
    g.A = A; g.B = Bm; g.bias = bias; g.C = C; g.M = M; g.N = N; g.K = K;
    g.transA = transA; g.lda = lda; g.rows_per_seg = rows_per_seg; g.seg_stride = seg_stride;
    g.transB = transB; g.ldb = ldb; g.ldc = ldc;
    g.in_mode = in_mode; g.in_scale = in_scale; g.in_shift = in_shift; g.mask_src = mask_src; g.alpha = alpha;
    g.accumulate = accumulate;
    const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
    PTTS_REQUIRE(tm <= 65535, "gemm: M too large");
    const int tiles = tm * tn;
    const int ksteps = (K + BK - 1) / BK;
    int splits = 1;
    if (tiles < 256 && ksteps >= 32) {
        splits = (768 + tiles - 1) / tiles;          // aim at ~3 workgroups per CU
        const int max_by_k = ksteps / 16;            // keep >= 16 k-steps (256 k) per split
        if (splits > max_by_k) splits = max_by_k;
        if (splits < 1) splits = 1;
        if (splits > 1024) splits = 1024;
    }
    int steps_per = (ksteps + splits - 1) / splits;
    g.k_per_split = steps_per * BK;
    splits = (K + g.k_per_split - 1) / g.k_per_split;
    g.splits = splits;
    if (splits > 1 && !accumulate) {
        hipError_t e;
        if (ldc == N) e = hipMemsetAsync(C, 0, (size_t)M * N * sizeof(float), st);
        else e = hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), (size_t)M, st);
        if (e != hipSuccess) { set_error("gemm: memset failed: %s", hipGetErrorString(e)); return PTTS_ELAUNCH; }
    }
    dim3 grid(tn, tm, splits), block(GEMM_THREADS);
    const bool conv = seg_stride != 0;
#define PTTS_GEMM_LAUNCH(TA, TB, CV, MD) hipLaunchKernelGGL((gemm_f32_mfma_kernel<TA, TB, CV, MD>), grid, block, 0, st, g)
#define PTTS_GEMM_MODES(TA, TB)                                                  \
    switch (in_mode) {                                                           \
        case PTTS_IN_LRELU: PTTS_GEMM_LAUNCH(TA, TB, 0, PTTS_IN_LRELU); break;   \
        case PTTS_IN_MASKMUL: PTTS_GEMM_LAUNCH(TA, TB, 0, PTTS_IN_MASKMUL); break; \
        default: PTTS_GEMM_LAUNCH(TA, TB, 0, PTTS_IN_NONE); break;               \
    }
    if (conv && in_mode == PTTS_IN_NONE && transA == 0 && transB == 0) PTTS_GEMM_LAUNCH(0, 0, 1, PTTS_IN_NONE);
    else if (conv && in_mode == PTTS_IN_NONE && transA == 1 && transB == 0) PTTS_GEMM_LAUNCH(1, 0, 1, PTTS_IN_NONE);
    else if (transA == 0 && transB == 0) { PTTS_GEMM_MODES(0, 0) }
    else if (transA == 0 && transB == 1) { PTTS_GEMM_MODES(0, 1) }
    else if (transA == 1 && transB == 0) { PTTS_GEMM_MODES(1, 0) }
    else { PTTS_GEMM_MODES(1, 1) }
#undef PTTS_GEMM_MODES
#undef PTTS_GEMM_LAUNCH
    return check_launch("gemm");
}

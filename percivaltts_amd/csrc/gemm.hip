// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, one rounding per
// product, same rate as the vector ALU but on its own pipe and with 1 operand VGPR per lane).
//
// Role on the hot path: the "dense label->hidden projections" of the north star -- the context
// Conv1D(k=21, ctx->256) of both networks (reference networktts.py:116-120 via
// networks_critic.py:82-83 and modeltts_common.py:75-76) run as an IMPLICIT GEMM over a
// zero-padded frame buffer (K = 21*ctx, row stride = ctx: the im2col matrix is never built),
// plus every Dense layer (networktts.py:59-63) and the LSTM input/weight-gradient products.
//
// Decomposition: stream-K.  The (tile, k-step) iteration space of all 128x128 output tiles is cut
// into equal contiguous ranges, one per persistent workgroup (two per CU), so the chip is
// evenly loaded whatever the tile count (400 tiles for the context Conv1D, 4 for a 256x256
// weight gradient).  A workgroup that owns a whole tile stores it; partial tiles are combined
// with fp32 atomics into a zeroed C.
// Tile loop: 128x128xBK per 256-thread workgroup, 2x2 waves, each wave 2x2 MFMA 32x32 tiles
// (64 accumulator VGPRs); LDS double-buffered as A[k][m], B[k][n] (conflict-free ds_read_b32
// fragment reads), global->register prefetch of tile k+1 in flight during the MFMAs of tile k,
// one barrier per k-step.  The previous layer's BatchNorm-affine/LeakyReLU (or the
// gradient-penalty mask) is applied while A is staged.
#include "common.h"

namespace ptts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDA_S = BM + 4;   // LDS leading dims (multiple of 4 floats: 16-B aligned rows)
constexpr int LDB_S = BN + 4;
constexpr int GEMM_THREADS = 256;
constexpr int NLD = BM * BK / 4 / GEMM_THREADS;   // float4 loads per lane per operand per k-step
constexpr int KQ = BK / 4;                        // float4 per row when k is the contiguous index

struct GemmArgs {
    const float* A; const float* B; const float* bias; float* C;
    int M, N, K;
    int transA; long long lda, rows_per_seg, seg_stride;
    int transB; long long ldb, ldc;
    int in_mode; const float* in_scale; const float* in_shift; const float* mask_src; float alpha;
    int accumulate;
    const float* out_mask; float out_alpha;   // C *= (out_mask > 0 ? 1 : out_alpha), out_mask laid out like C
    int tiles_n, ksteps;          // tiles along N; k-steps per tile
    long long iters_total;        // tiles * ksteps
    int workers;
    int prio;                     // raise the issue priority of the second half of the grid (see the kernel)
    float* colsum_b;              // optional [N]: += column sums of B over k (the bias gradient beside a weight gradient)
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

// element offset of stored row r of A (32-bit divide: r and rows_per_seg fit an int)
__device__ __forceinline__ long long rowbase(const GemmArgs& g, int r) {
    const int rps = (int)g.rows_per_seg;
    const int seg = r / rps;
    return (long long)seg * g.seg_stride + (long long)(r - seg * rps) * g.lda;
}

// One k-step of operands in flight: raw A and B quads, the GP mask quads (MASKMUL) and the BN affine of this lane's
// four channels (LRELU).  The transform is applied when the registers are written to LDS -- after the MFMAs of the
// current k-step -- so that nothing waits on the loads before the matrix pipe has its work.
struct Frag { f32x4u a[NLD], b[NLD], m[NLD], sc, sh; };

// 4 consecutive floats along the contiguous index c of a row; CHECK guards row validity and the end of the row.
template <bool CHECK>
__device__ __forceinline__ f32x4u load4(const float* __restrict__ p, bool row_ok, int c, int limit) {
    f32x4u v = {0.f, 0.f, 0.f, 0.f};
    if (!CHECK || (row_ok && c + 3 < limit)) {
        v = *reinterpret_cast<const f32x4u*>(p);
    } else if (row_ok && c < limit) {          // the vector straddles the end of the row: rare, scalar
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < limit) v[e] = p[e];
    }
    return v;
}

// Row addressing of this lane's A quads, kept out of the k-loop: the 32-bit division of rowbase() costs ~40 vector
// instructions, and four of them per k-step were a sixth of the MFMA time of a step.
//  TRANSA == 0: the stored rows (m) are fixed for a tile: one division per quad per tile.
//  TRANSA == 1: the stored row is k; (segment, row in segment) advance by BK per k-step without dividing.
struct ARows {
    long long base[NLD];     // TRANSA == 0: element offset of the row;  TRANSA == 1: unused
    int seg[NLD], rem[NLD];  // TRANSA == 1: segment and row-in-segment of this lane's k row at the current k-step
};

template <int TRANSA>
__device__ __forceinline__ void arows_begin(const GemmArgs& g, int m0, int k0, ARows& ar) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int q = tid + j * GEMM_THREADS;
        if (TRANSA == 0) {
            const int m = m0 + q / KQ;
            ar.base[j] = rowbase(g, m < g.M ? m : 0);
        } else {
            const int k = k0 + q / (BM / 4);
            const int rps = (int)g.rows_per_seg;
            ar.seg[j] = k / rps;
            ar.rem[j] = k - ar.seg[j] * rps;
        }
    }
}

template <int TRANSA>
__device__ __forceinline__ void arows_advance(const GemmArgs& g, ARows& ar) {
    if (TRANSA == 1) {
        const int rps = (int)g.rows_per_seg;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            ar.rem[j] += BK;
            while (ar.rem[j] >= rps) { ar.rem[j] -= rps; ++ar.seg[j]; }
        }
    }
}

// Issue the loads of one 128xBK tile of A and one BKx128 tile of B.  CHECK=false for interior tiles and full k-steps.
// Out-of-range elements load as 0 (also scale/shift), and an out-of-range row of one operand always meets zeros of
// the other or an output row that is never stored, so no masking is needed after the affine.
// `ar` must describe k-step k0 (arows_begin at the first step of a tile segment, arows_advance after every load).
template <int TRANSA, int TRANSB, int MODE, bool CHECK>
__device__ __forceinline__ void load_tiles(const GemmArgs& g, int m0, int n0, int k0, const ARows& ar, Frag& fr) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int q = tid + j * GEMM_THREADS;
        if (TRANSA == 0) {
            const int m = m0 + q / KQ, k = k0 + (q % KQ) * 4;
            const bool ok = !CHECK || m < g.M;
            const long long offa = ar.base[j] + k;
            fr.a[j] = load4<CHECK>(g.A + offa, ok, k, g.K);
            if (MODE == PTTS_IN_MASKMUL) fr.m[j] = load4<CHECK>(g.mask_src + offa, ok, k, g.K);
        } else {
            const int k = k0 + q / (BM / 4), m = m0 + (q % (BM / 4)) * 4;
            const bool ok = !CHECK || k < g.K;
            const long long offa = (ok ? (long long)ar.seg[j] * g.seg_stride + (long long)ar.rem[j] * g.lda : 0) + m;
            fr.a[j] = load4<CHECK>(g.A + offa, ok, m, g.M);
            if (MODE == PTTS_IN_MASKMUL) fr.m[j] = load4<CHECK>(g.mask_src + offa, ok, m, g.M);
        }
        if (TRANSB == 0) {
            const int k = k0 + q / (BN / 4), n = n0 + (q % (BN / 4)) * 4;
            const bool ok = !CHECK || k < g.K;
            fr.b[j] = load4<CHECK>(g.B + (long long)(ok ? k : 0) * g.ldb + n, ok, n, g.N);
        } else {
            const int n = n0 + q / KQ, k = k0 + (q % KQ) * 4;
            const bool ok = !CHECK || n < g.N;
            fr.b[j] = load4<CHECK>(g.B + (long long)(ok ? n : 0) * g.ldb + k, ok, k, g.K);
        }
    }
    if (MODE == PTTS_IN_LRELU && g.in_scale) {
        // the channel is the stored column of A: the same four for every quad of this lane
        const int ch = TRANSA == 0 ? k0 + (tid % KQ) * 4 : m0 + (tid % (BM / 4)) * 4;
        const int lim = TRANSA == 0 ? g.K : g.M;
        fr.sc = load4<true>(g.in_scale + ch, true, ch, lim);
        fr.sh = load4<true>(g.in_shift + ch, true, ch, lim);
    }
}

template <int TRANSA, int TRANSB, int MODE>
__device__ __forceinline__ void store_tiles(const GemmArgs& g, float* As, float* Bs, const Frag& fr, float* bsum) {
    const int tid = threadIdx.x;
    if (TRANSB == 0 && bsum) {           // this lane's four columns of B, every k row it stages
#pragma unroll
        for (int j = 0; j < NLD; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) bsum[e] += fr.b[j][e];
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int q = tid + j * GEMM_THREADS;
        float a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = fr.a[j][e];
            if (MODE == PTTS_IN_LRELU) {
                if (g.in_scale) v = v * fr.sc[e] + fr.sh[e];
                v = lrelu(v, g.alpha);
            } else if (MODE == PTTS_IN_MASKMUL) {
                v *= lrelu_d(fr.m[j][e], g.alpha);
            }
            a[e] = v;
        }
        if (TRANSA == 0) {
            const int m = q / KQ, k = (q % KQ) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) As[(k + e) * LDA_S + m] = a[e];
        } else {
            const int k = q / (BM / 4), m = (q % (BM / 4)) * 4;
            *reinterpret_cast<float4*>(As + k * LDA_S + m) = make_float4(a[0], a[1], a[2], a[3]);
        }
        if (TRANSB == 0) {
            const int k = q / (BN / 4), n = (q % (BN / 4)) * 4;
            *reinterpret_cast<float4*>(Bs + k * LDB_S + n) = make_float4(fr.b[j][0], fr.b[j][1], fr.b[j][2], fr.b[j][3]);
        } else {
            const int n = q / KQ, k = (q % KQ) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) Bs[(k + e) * LDB_S + n] = fr.b[j][e];
        }
    }
}

// CONV != 0 marks the implicit-convolution instantiation (same code; its own symbol so that a profile separates the
// context-Conv1D products from the small Dense ones).
// One contiguous run of k-steps [ks0, ks1) of one 128x128 tile: the body of both the single-product kernel and the
// grouped weight-gradient kernel.  ATOMIC forces the fp32-atomic epilogue (accumulation into a buffer other
// workgroups / streams also add to).
template <int TRANSA, int TRANSB, int MODE, bool ATOMIC>
__device__ __forceinline__ void gemm_segment(const GemmArgs& g, float (*As)[BK * LDA_S], float (*Bs)[BK * LDB_S],
                                             int tile, int ks0, int ks1) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int l31 = lane & 31, lh = lane >> 5;
    const int m0 = (tile / g.tiles_n) * BM, n0 = (tile % g.tiles_n) * BN;
    const bool interior = (m0 + BM <= g.M) && (n0 + BN <= g.N);
    const int kbeg = ks0 * BK;
    const int kend = min(g.K, ks1 * BK);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Frag fr;
    ARows ar;
    arows_begin<TRANSA>(g, m0, kbeg, ar);
    auto load = [&](int k0) {           // called for kbeg, kbeg + BK, ... in order
        if (interior && k0 + BK <= g.K) load_tiles<TRANSA, TRANSB, MODE, false>(g, m0, n0, k0, ar, fr);
        else load_tiles<TRANSA, TRANSB, MODE, true>(g, m0, n0, k0, ar, fr);
        arows_advance<TRANSA>(g, ar);
    };
    // bias gradient: the workgroups of the first row of tiles also sum the columns of B over their k range
    float bsum_r[4] = {0.f, 0.f, 0.f, 0.f};
    float* bsum = (TRANSB == 0 && g.colsum_b != nullptr && m0 == 0) ? bsum_r : nullptr;
    load(kbeg);
    store_tiles<TRANSA, TRANSB, MODE>(g, As[0], Bs[0], fr, bsum);
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK, buf ^= 1) {
        const bool more = k0 + BK < kend;
        if (more) load(k0 + BK);                 // global -> registers, in flight during the MFMAs below
        const float* as = As[buf] + lh * LDA_S + wm + l31;
        const float* bs = Bs[buf] + lh * LDB_S + wn + l31;
        float ra[BK / 2][2], rb[BK / 2][2];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            ra[kk][0] = as[2 * kk * LDA_S];
            ra[kk][1] = as[2 * kk * LDA_S + 32];
            rb[kk][0] = bs[2 * kk * LDB_S];
            rb[kk][1] = bs[2 * kk * LDB_S + 32];
        }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[kk][0], rb[kk][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[kk][0], rb[kk][1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[kk][1], rb[kk][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[kk][1], rb[kk][1], acc[1][1], 0, 0, 0);
        }
        if (more) store_tiles<TRANSA, TRANSB, MODE>(g, As[buf ^ 1], Bs[buf ^ 1], fr, bsum);
        __syncthreads();
    }
    if (bsum) {
        // lanes tid and tid+32k hold the same four columns (rows k differ): combine the 8 through LDS (As is free)
        float* red = As[0];
#pragma unroll
        for (int e = 0; e < 4; ++e) red[(tid >> 5) * BN + (tid & 31) * 4 + e] = bsum_r[e];
        __syncthreads();
        if (tid < BN) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < GEMM_THREADS / 32; ++r) t += red[r * BN + tid];
            if (n0 + tid < g.N) atomicAdd(g.colsum_b + n0 + tid, t);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool whole = !ATOMIC && (ks0 == 0) && (ks1 == g.ksteps);
    const bool add_bias = g.bias != nullptr && ks0 == 0;
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
                float v = acc[i][j][r] + bv;
                if (g.out_mask) v *= lrelu_d(g.out_mask[(long long)m * g.ldc + n], g.out_alpha);
                if (!whole) atomicAdd(cp, v);
                else if (g.accumulate) *cp += v;
                else *cp = v;
            }
        }
}

template <int TRANSA, int TRANSB, int CONV, int MODE>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_f32_mfma_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][BK * LDA_S];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB_S];
    // Two workgroups share a CU and run the same program: left alone they fall into lockstep (MFMA phases together,
    // then load/store/barrier phases together with the matrix pipe idle).  Giving one of the pair -- the second half
    // of the grid, dispatched onto the CUs the first half already occupies -- a higher issue priority breaks the tie.
    if (g.prio && (int)blockIdx.x >= (int)(gridDim.x / 2)) __builtin_amdgcn_s_setprio(3);
    // this workgroup's contiguous share of the (tile, k-step) space
    long long it = g.iters_total * blockIdx.x / g.workers;
    const long long it_end = g.iters_total * (blockIdx.x + 1) / g.workers;
    while (it < it_end) {
        const int tile = (int)(it / g.ksteps);
        const int ks0 = (int)(it - (long long)tile * g.ksteps);
        const int ks1 = (int)min((long long)g.ksteps, ks0 + (it_end - it));
        it += ks1 - ks0;
        gemm_segment<TRANSA, TRANSB, MODE, false>(g, As, Bs, tile, ks0, ks1);
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant of the 128x128 kernel for products without an A transform and B stored [k][n] (the context Conv1D
// forward and weight gradient): the tiles go global -> LDS with global_load_lds_dwordx4 (each lane names its own 16
// source bytes; a wave-instruction fills 1 KB of LDS: lane L at base + 16 L), three stages deep, so the k-loop holds no
// staging registers, no ds_write and no wait on loads in front of the MFMAs.  The legacy body serves edge tiles.
//   A, TRANSA == 0: stage image [128 rows][16 k], the four quads of a row XOR-swizzled by (row>>2)&3 so that the
//     fragment reads are conflict-free ds_read_b128: lane (l31, lh) takes k = 8 lh + 4 j + e of row l31 (j = 0,1) and
//     MFMA (j, e) multiplies the k-pair {4j+e, 8+4j+e};  B (and A^T for TRANSA == 1): image [16 k][128], read at row
//     8 lh + 4 j + e.
// ------------------------------------------------------------------------------------------------
typedef const void __attribute__((address_space(1)))* dma_gptr;
typedef void __attribute__((address_space(3)))* dma_lptr;
constexpr int DMA_STAGES = 3;
constexpr int DMA_STAGE_FLOATS = BM * BK + BK * BN;      // 4096 floats = 16 KB

// Issued through inline assembly: the compiler treats its builtin for this instruction as an LDS write that every later
// LDS read and every following DMA must wait for (s_waitcnt vmcnt(0) after each one), which serialises exactly the loads
// this path exists to keep in flight.  The kernel orders them itself (s_waitcnt vmcnt(N) + barrier before a stage is read).
__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
    // wave-uniform LDS byte address -> M0
    const unsigned lds_off = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long)(dma_lptr)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_off) : "memory", "m0");
}

template <int TRANSA, bool ATOMIC>
__device__ __forceinline__ void gemm_segment_dma(const GemmArgs& g, float* lds, int tile, int ks0, int ks1) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int l31 = lane & 31, lh = lane >> 5;
    const int m0 = (tile / g.tiles_n) * BM, n0 = (tile % g.tiles_n) * BN;
    const int kfull = g.K / BK;                      // k-steps that lie completely inside K

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per-lane source addressing of the two A and two B wave-instructions of a k-step
    long long abase[2];            // TRANSA == 0: element offset of this lane's row (+ swizzled quad);  == 1: column offset
    int aseg[2], arem[2];          // TRANSA == 1: (segment, row in segment) of this lane's k row at the current step
    if (TRANSA == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = 32 * wave + 16 * q + (lane >> 2);          // row of the tile
            const int cg = (lane & 3) ^ ((r >> 2) & 3);              // global quad that goes to this lane's LDS slot
            abase[q] = rowbase(g, m0 + r) + 4 * cg;
        }
    } else {
        const int rps = (int)g.rows_per_seg;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k = ks0 * BK + 4 * wave + 2 * q + (lane >> 5);
            aseg[q] = k / rps; arem[q] = k - aseg[q] * rps;
            abase[q] = m0 + 4 * (lane & 31);
        }
    }
    const float* bsrc = g.B + (long long)(4 * wave + (lane >> 5)) * g.ldb + n0 + 4 * (lane & 31);   // + (k0 + 2q) * ldb

    auto issue = [&](int ks, int stage) {        // the four DMA wave-instructions of k-step ks into `stage`
        float* st = lds + stage * DMA_STAGE_FLOATS;
        const int k0 = ks * BK;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (TRANSA == 0) {
                dma16(g.A + abase[q] + k0, st + (32 * wave + 16 * q) * BK);
            } else {
                dma16(g.A + (long long)aseg[q] * g.seg_stride + (long long)arem[q] * g.lda + abase[q], st + (4 * wave + 2 * q) * BM);
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) dma16(bsrc + (long long)(k0 + 2 * q) * g.ldb, st + BM * BK + (4 * wave + 2 * q) * BN);
        if (TRANSA == 1) {
            const int rps = (int)g.rows_per_seg;
#pragma unroll
            for (int q = 0; q < 2; ++q) { arem[q] += BK; while (arem[q] >= rps) { arem[q] -= rps; ++aseg[q]; } }
        }
    };
    // a k-step that crosses K (the tail): registers, zeros beyond K, same LDS image
    auto stage_tail = [&](int ks, int stage) {
        float* st = lds + stage * DMA_STAGE_FLOATS;
        const int k0 = ks * BK;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (TRANSA == 0) {
                const int r = 32 * wave + 16 * q + (lane >> 2);
                const int cg = (lane & 3) ^ ((r >> 2) & 3);
                const float* p = g.A + abase[q] + k0;
                const int k = k0 + 4 * cg;
                if (k < g.K) v.x = p[0];
                if (k + 1 < g.K) v.y = p[1];
                if (k + 2 < g.K) v.z = p[2];
                if (k + 3 < g.K) v.w = p[3];
                *reinterpret_cast<float4*>(st + r * BK + 4 * (lane & 3)) = v;
            } else {
                const int kr = 4 * wave + 2 * q + (lane >> 5);
                if (k0 + kr < g.K) {
                    const f32x4u t = *reinterpret_cast<const f32x4u*>(g.A + (long long)aseg[q] * g.seg_stride + (long long)arem[q] * g.lda + abase[q]);
                    v = make_float4(t[0], t[1], t[2], t[3]);
                }
                *reinterpret_cast<float4*>(st + kr * BM + 4 * (lane & 31)) = v;
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int kr = 4 * wave + 2 * q + (lane >> 5);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 + kr < g.K) {
                const f32x4u t = *reinterpret_cast<const f32x4u*>(bsrc + (long long)(k0 + 2 * q) * g.ldb);
                v = make_float4(t[0], t[1], t[2], t[3]);
            }
            *reinterpret_cast<float4*>(st + BM * BK + kr * BN + 4 * (lane & 31)) = v;
        }
    };
    auto feed = [&](int ks, int stage) { if (ks < kfull) issue(ks, stage); else stage_tail(ks, stage); };

    const bool want_bsum = g.colsum_b != nullptr && m0 == 0;
    float bsum = 0.f;
    const int nsteps = ks1 - ks0;
    constexpr int AHEAD = DMA_STAGES - 1;          // k-steps in flight beyond the one being multiplied
#pragma unroll
    for (int p = 0; p < AHEAD; ++p)
        if (p < nsteps) feed(ks0 + p, p);
    for (int s = 0; s < nsteps; ++s) {
        // this wave's DMAs of step s have landed when at most those of the later steps already issued (4 each) are in
        // flight; a tail step (staged through registers) drains everything itself
        int later = nsteps - 1 - s;
        if (later > AHEAD - 1) later = AHEAD - 1;
        if (ks0 + s + later >= kfull) later = 0;
        if (later >= 2) __builtin_amdgcn_s_waitcnt(0xF78);           // vmcnt(8)
        else if (later == 1) __builtin_amdgcn_s_waitcnt(0xF74);      // vmcnt(4)
        else __builtin_amdgcn_s_waitcnt(0xF70);                      // vmcnt(0)
        __syncthreads();           // every wave's part of stage s is in LDS; the stage of step s-1 is no longer read
        if (s + AHEAD < nsteps) feed(ks0 + s + AHEAD, (s + AHEAD) % DMA_STAGES);
        const float* st = lds + (s % DMA_STAGES) * DMA_STAGE_FLOATS;
        const float* bt = st + BM * BK;
        if (want_bsum && tid < BN) {          // bias gradient: column sums of the B tile (first row of tiles only)
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) bsum += bt[kk * BN + tid];
        }
        float av[2][2][4], bv[2][2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (TRANSA == 0) {
                const int r = wm + 32 * i + l31;
                const int sw = (r >> 2) & 3;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 t = *reinterpret_cast<const float4*>(st + r * BK + 4 * ((2 * lh + j) ^ sw));
                    av[i][j][0] = t.x; av[i][j][1] = t.y; av[i][j][2] = t.z; av[i][j][3] = t.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[i][j][e] = st[(8 * lh + 4 * j + e) * BM + wm + 32 * i + l31];
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[t][j][e] = bt[(8 * lh + 4 * j + e) * BN + wn + 32 * t + l31];
        // (forcing all fragment reads ahead of the MFMAs with a sched_barrier measured 10 % slower than the scheduler's
        // interleaving; so did a fourth stage: the loop is not bound by load latency any more)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][j][e], bv[0][j][e], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][j][e], bv[1][j][e], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][j][e], bv[0][j][e], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][j][e], bv[1][j][e], acc[1][1], 0, 0, 0);
            }
    }
    __syncthreads();               // the stages may be refilled by the next segment
    if (want_bsum && tid < BN) atomicAdd(g.colsum_b + n0 + tid, bsum);

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool whole = !ATOMIC && (ks0 == 0) && (ks1 == g.ksteps);
    const bool add_bias = g.bias != nullptr && ks0 == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + j * 32 + l31;
            const float bv2 = add_bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float* cp = g.C + (long long)m * g.ldc + n;
                const float v = acc[i][j][r] + bv2;
                if (!whole) atomicAdd(cp, v);
                else if (g.accumulate) *cp += v;
                else *cp = v;
            }
        }
}

// CONV != 0 marks the implicit-convolution instantiation (same code; its own symbol so that a profile separates the
// context-Conv1D products from the other ones that take this path).
template <int TRANSA, int CONV>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_dma_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[DMA_STAGES * DMA_STAGE_FLOATS];
    if (g.prio && (int)blockIdx.x >= (int)(gridDim.x / 2)) __builtin_amdgcn_s_setprio(3);
    long long it = g.iters_total * blockIdx.x / g.workers;
    const long long it_end = g.iters_total * (blockIdx.x + 1) / g.workers;
    while (it < it_end) {
        const int tile = (int)(it / g.ksteps);
        const int ks0 = (int)(it - (long long)tile * g.ksteps);
        const int ks1 = (int)min((long long)g.ksteps, ks0 + (it_end - it));
        it += ks1 - ks0;
        const int m0 = (tile / g.tiles_n) * BM, n0 = (tile % g.tiles_n) * BN;
        if ((m0 + BM <= g.M) && (n0 + BN <= g.N)) {
            gemm_segment_dma<TRANSA, false>(g, lds, tile, ks0, ks1);
        } else {
            // edge tile: the register-staged body (its two double-buffered images fit in the same LDS)
            float (*As)[BK * LDA_S] = reinterpret_cast<float (*)[BK * LDA_S]>(lds);
            float (*Bs)[BK * LDB_S] = reinterpret_cast<float (*)[BK * LDB_S]>(lds + 2 * BK * LDA_S);
            gemm_segment<TRANSA, 0, PTTS_IN_NONE, false>(g, As, Bs, tile, ks0, ks1);
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Grouped weight gradients: up to WG_MAX products C_i += a_i^T . dy_i (+ column sums of dy_i) in ONE launch.  A Dense
// layer's weight gradient alone (256 x 256 x 25 600) has 4 tiles: split over 512 workgroups each gets 12 k-steps and
// pays the launch, the zero-fill and a 64 KB atomic epilogue for them (47 TF).  All layers of a backward pass together
// give every workgroup a long contiguous run of k-steps and few partial tiles.  Every product accumulates with fp32
// atomics into its (zero-initialised or already partly filled) gradient buffer.
// ------------------------------------------------------------------------------------------------
constexpr int WG_MAX = 12;
struct WGroup {
    const float* A; const float* B; float* C; float* colsum_b;
    const float* in_scale; const float* in_shift; const float* mask_src;
    int M, N, K, lda, ldb, ldc, in_mode; float alpha;
};
struct WGroupArgs {
    int n, workers;
    long long it_begin[WG_MAX + 1];       // first (tile, k-step) iteration of each product; [n] = total
    WGroup gr[WG_MAX];
};

__global__ __launch_bounds__(GEMM_THREADS) void gemm_wgrad_grouped_kernel(WGroupArgs ga) {
    __shared__ __attribute__((aligned(16))) float As[2][BK * LDA_S];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB_S];
    if ((int)blockIdx.x >= (int)(gridDim.x / 2)) __builtin_amdgcn_s_setprio(3);
    const long long total = ga.it_begin[ga.n];
    long long it = total * blockIdx.x / ga.workers;
    const long long it_end = total * (blockIdx.x + 1) / ga.workers;
    int gi = 0;
    while (it < it_end) {
        while (it >= ga.it_begin[gi + 1]) ++gi;
        const WGroup& w = ga.gr[gi];
        GemmArgs g;
        g.A = w.A; g.B = w.B; g.bias = nullptr; g.C = w.C; g.M = w.M; g.N = w.N; g.K = w.K;
        g.transA = 1; g.lda = w.lda; g.rows_per_seg = w.K; g.seg_stride = 0;
        g.transB = 0; g.ldb = w.ldb; g.ldc = w.ldc;
        g.in_mode = w.in_mode; g.in_scale = w.in_scale; g.in_shift = w.in_shift; g.mask_src = w.mask_src; g.alpha = w.alpha;
        g.accumulate = 1; g.out_mask = nullptr; g.out_alpha = w.alpha;
        g.tiles_n = (w.N + BN - 1) / BN; g.ksteps = (w.K + BK - 1) / BK;
        g.colsum_b = w.colsum_b;
        const long long rel = it - ga.it_begin[gi];
        const int tile = (int)(rel / g.ksteps);
        const int ks0 = (int)(rel - (long long)tile * g.ksteps);
        long long room = ga.it_begin[gi + 1] - it;            // stay inside this product
        if (room > it_end - it) room = it_end - it;
        const int ks1 = (int)min((long long)g.ksteps, ks0 + room);
        it += ks1 - ks0;
        if (w.in_mode == PTTS_IN_LRELU) gemm_segment<1, 0, PTTS_IN_LRELU, true>(g, As, Bs, tile, ks0, ks1);
        else if (w.in_mode == PTTS_IN_MASKMUL) gemm_segment<1, 0, PTTS_IN_MASKMUL, true>(g, As, Bs, tile, ks0, ks1);
        else gemm_segment<1, 0, PTTS_IN_NONE, true>(g, As, Bs, tile, ks0, ks1);
    }
}

// ------------------------------------------------------------------------------------------------
// "Tall" GEMM: M >> N, 128 < N <= 256, K <= 2048 (the Dense layers forward and backward-data).  One workgroup owns
// TBM x 256 outputs -- ALL columns -- so A is streamed from HBM/L2 exactly once and TBM is chosen so that ceil(M/TBM)
// fills the 256 CUs in one round (M = 25 600 -> TBM = 112 -> 229 workgroups).  No split, no atomics: bitwise
// reproducible.  v_mfma_f32_16x16x4_f32; 8 waves, wave w owns columns [32w, 32w+32) x all TBM/16 row tiles
// (14 accumulator tiles at TBM = 112).
// ------------------------------------------------------------------------------------------------
typedef float f32x4c __attribute__((ext_vector_type(4)));
constexpr int TBN = 256;
constexpr int TALL_THREADS = 512;            // 8 waves: two per SIMD hide each other's LDS latency and barriers
constexpr int NTW = TBN / 16 / (TALL_THREADS / 64);   // column tiles of 16 per wave (2)
constexpr int TLDB = TBN + 4;

// ------------------------------------------------------------------------------------------------
// The kernel: k-contiguous A tile, BKT = 16 (or 32) per k-step:
//  * the A tile keeps the global layout [row][k] in LDS (row stride BKT+8 floats: conflict-free ds_read_b128), staged
//    with one ds_write_b128 per quad; a lane fetches four k-values of a row with ONE ds_read_b128.  MFMA number e of a
//    16-wide k-group multiplies the k-set {e, 4+e, 8+e, 12+e} (the sum over k does not care about the order) and B is
//    read at rows 4*q + e;
//  * every fragment of a 16-wide k-group is requested before its 8*MT MFMAs, and with BKT = 32 the second group's
//    fragments are requested before the first group's MFMAs issue, so one barrier is amortised over 16*MT MFMAs.
// ------------------------------------------------------------------------------------------------
template <int TRANSB, int MODE, int MT, int BKT>
__global__ __launch_bounds__(TALL_THREADS) void gemm_tall_kc_kernel(GemmArgs g) {
    constexpr int TBM = 16 * MT;
    constexpr int SA = BKT + 8;                 // 24 / 40: both conflict-free for the b128 fragment reads
    constexpr int QR = BKT / 4;                 // quads per A row (and per B row when B is stored [n][k])
    constexpr int KG = BKT / 16;                // 16-wide k-groups per k-step
    __shared__ __attribute__((aligned(16))) float As[2][TBM * SA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BKT * TLDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * TBM;
    const bool interior = (m0 + TBM <= g.M) && (g.N == TBN);
    constexpr int NA = (TBM * QR + TALL_THREADS - 1) / TALL_THREADS;
    constexpr int NB = TBN * QR / TALL_THREADS;

    f32x4c acc[MT][NTW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4c){0.f, 0.f, 0.f, 0.f};

    // Loop-invariant addresses of this lane's quads: the row base of A (one 32-bit division each, done once) and the
    // B column / row; a k-step adds k0 (or k0*ldb).  Rows beyond M are clamped to row 0 and flagged.
    const float* pa[NA]; const float* pm[NA]; bool oka[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int qq = tid + j * TALL_THREADS;
        const int row = qq / QR;
        oka[j] = row < TBM && m0 + row < g.M;
        const long long base = rowbase(g, oka[j] ? m0 + row : 0) + (qq % QR) * 4;
        pa[j] = g.A + base;
        pm[j] = MODE == PTTS_IN_MASKMUL ? g.mask_src + base : nullptr;
    }
    const float* pb[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int qq = tid + j * TALL_THREADS;
        if (TRANSB == 0) pb[j] = g.B + (long long)(qq >> 6) * g.ldb + (qq & 63) * 4;
        else pb[j] = g.B + (long long)((qq / QR) < g.N ? (qq / QR) : 0) * g.ldb + (qq % QR) * 4;
    }
    const long long bstep = TRANSB == 0 ? g.ldb : 1;     // B pointer advance per unit of k

    struct Stage { f32x4u va[NA], vm[NA], vb[NB], vsc, vsh; };
    auto load = [&](int k0, bool check, Stage& sg) {
        const f32x4u z4 = {0.f, 0.f, 0.f, 0.f};
        if (!check) {
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                sg.va[j] = oka[j] ? *reinterpret_cast<const f32x4u*>(pa[j] + k0) : z4;
                if (MODE == PTTS_IN_MASKMUL) sg.vm[j] = oka[j] ? *reinterpret_cast<const f32x4u*>(pm[j] + k0) : z4;
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) sg.vb[j] = *reinterpret_cast<const f32x4u*>(pb[j] + k0 * bstep);
        } else {
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const int k = k0 + ((tid + j * TALL_THREADS) % QR) * 4;
                sg.va[j] = load4<true>(pa[j] + k0, oka[j], k, g.K);
                if (MODE == PTTS_IN_MASKMUL) sg.vm[j] = load4<true>(pm[j] + k0, oka[j], k, g.K);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int qq = tid + j * TALL_THREADS;
                if (TRANSB == 0) {                          // stored [k][n]
                    const int k = k0 + (qq >> 6), n = (qq & 63) * 4;
                    const bool ok = k < g.K;
                    sg.vb[j] = load4<true>(ok ? pb[j] + k0 * bstep : g.B, ok, n, g.N);
                } else {                                    // stored [n][k]
                    const int n = qq / QR, k = k0 + (qq % QR) * 4;
                    sg.vb[j] = load4<true>(pb[j] + k0, n < g.N, k, g.K);
                }
            }
        }
        if (MODE == PTTS_IN_LRELU && g.in_scale) {      // this lane's four channels (k-quad) of the k-step
            const int ch = k0 + (tid % QR) * 4;
            sg.vsc = load4<true>(g.in_scale + ch, true, ch, g.K);
            sg.vsh = load4<true>(g.in_shift + ch, true, ch, g.K);
        }
    };
    auto store = [&](float* as, float* bs, const Stage& sg) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int qq = tid + j * TALL_THREADS;
            const int row = qq / QR, k = (qq % QR) * 4;
            if (row < TBM) {
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = sg.va[j][e];
                    if (MODE == PTTS_IN_LRELU) {
                        if (g.in_scale) v = v * sg.vsc[e] + sg.vsh[e];
                        v = lrelu(v, g.alpha);
                    } else if (MODE == PTTS_IN_MASKMUL) {
                        v *= lrelu_d(sg.vm[j][e], g.alpha);
                    }
                    t[e] = v;
                }
                *reinterpret_cast<float4*>(as + row * SA + k) = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int qq = tid + j * TALL_THREADS;
            if (TRANSB == 0) {
                const int k = qq >> 6, n = (qq & 63) * 4;
                *reinterpret_cast<float4*>(bs + k * TLDB + n) = make_float4(sg.vb[j][0], sg.vb[j][1], sg.vb[j][2], sg.vb[j][3]);
            } else {
                const int n = qq / QR, k = (qq % QR) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) bs[(k + e) * TLDB + n] = sg.vb[j][e];
            }
        }
    };
    auto compute = [&](int buf) {
        const float* as = As[buf] + r16 * SA + 4 * q;
        const float* bs = Bs[buf] + (4 * q) * TLDB + wave * (16 * NTW) + r16;
        float4 fa[KG][MT];
        float fb[KG][4][NTW];
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[kg][i] = *reinterpret_cast<const float4*>(as + i * 16 * SA + kg * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < NTW; ++j) fb[kg][e][j] = bs[(kg * 16 + e) * TLDB + j * 16];
        }
        __builtin_amdgcn_sched_barrier(0);      // every fragment of the k-step is requested before the first MFMA
#pragma unroll
        for (int kg = 0; kg < KG; ++kg)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const float av = e == 0 ? fa[kg][i].x : e == 1 ? fa[kg][i].y : e == 2 ? fa[kg][i].z : fa[kg][i].w;
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, fb[kg][e][j], acc[i][j], 0, 0, 0);
                }
    };

    // Two register stages: the loads of k-step s+2 are issued before the MFMAs of step s and the loads of step s+1
    // (issued one step earlier) are written to LDS after them, so every global load has two k-steps to arrive.
    const int K = g.K;
    auto chk = [&](int k0) { return !(interior && k0 + BKT <= K); };
    Stage s0, s1;
    load(0, chk(0), s0);
    store(As[0], Bs[0], s0);
    if (BKT < K) load(BKT, chk(BKT), s1);
    __syncthreads();
    auto step = [&](int k0, int buf, const Stage& pend, Stage& next) {
        if (k0 + 2 * BKT < K) load(k0 + 2 * BKT, chk(k0 + 2 * BKT), next);
        compute(buf);
        if (k0 + BKT < K) store(As[buf ^ 1], Bs[buf ^ 1], pend);
        __syncthreads();
    };
    for (int k0 = 0; k0 < K; k0 += 2 * BKT) {
        step(k0, 0, s1, s0);
        if (k0 + BKT < K) step(k0 + BKT, 1, s0, s1);
    }
    // epilogue: each 16-row band goes through the (now free) B buffers so that C and the output mask move as whole rows
    const int col4 = (tid & 63) * 4, erow = tid >> 6;
    float bv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = (g.bias && col4 + e < g.N) ? g.bias[col4 + e] : 0.f;
    const bool vec = (g.N == TBN) && (g.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0) &&
                     (!g.out_mask || (reinterpret_cast<uintptr_t>(g.out_mask) & 15) == 0);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float* sb = Bs[i & 1];
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) sb[(q * 4 + r) * TLDB + wave * (16 * NTW) + j * 16 + r16] = acc[i][j][r];
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = erow + 8 * h;
            const int m = m0 + i * 16 + row;
            if (m >= g.M) continue;
            const float4 t = *reinterpret_cast<const float4*>(sb + row * TLDB + col4);
            float v[4] = {t.x + bv[0], t.y + bv[1], t.z + bv[2], t.w + bv[3]};
            const long long off = (long long)m * g.ldc + col4;
            if (vec) {
                if (g.out_mask) {
                    const float4 mk = *reinterpret_cast<const float4*>(g.out_mask + off);
                    v[0] *= lrelu_d(mk.x, g.out_alpha); v[1] *= lrelu_d(mk.y, g.out_alpha);
                    v[2] *= lrelu_d(mk.z, g.out_alpha); v[3] *= lrelu_d(mk.w, g.out_alpha);
                }
                float4* cp = reinterpret_cast<float4*>(g.C + off);
                if (g.accumulate) { const float4 o = *cp; v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
                *cp = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (col4 + e >= g.N) continue;
                    float w = v[e];
                    if (g.out_mask) w *= lrelu_d(g.out_mask[off + e], g.out_alpha);
                    if (g.accumulate) g.C[off + e] += w; else g.C[off + e] = w;
                }
            }
        }
    }
}

// rows per workgroup (multiple of 16, <= 128) that best fills 256 CUs in whole rounds
static int pick_tall_mt(int M) {
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("PTTS_TALL_MT"); forced = e ? atoi(e) : 0; }
    if (forced >= 4 && forced <= 8) return forced;
    int best = 8; double best_eff = -1.0;
    for (int mt = 4; mt <= 8; ++mt) {
        const long long blocks = (M + 16 * mt - 1) / (16 * mt);
        const long long rounds = (blocks + 255) / 256;
        const double eff = (double)M / (double)(rounds * 256 * 16 * mt);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = mt; }
    }
    return best;
}

}  // namespace ptts

using namespace ptts;

static bool thin_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PTTS_GEMM_THIN"); v = e ? atoi(e) : 1; }
    return v != 0;
}

static long long gemm_slots() {
    static long long slots_env = -1;
    if (slots_env < 0) { const char* e = getenv("PTTS_GEMM_SLOTS"); slots_env = e ? atoll(e) : 0; }
    return slots_env > 0 ? slots_env : 512;
}

extern "C" int ptts_gemm_wgrad_grouped(const ptts_wgrad_desc* descs, int n, void* stream) {
    PTTS_REQUIRE(descs && n > 0, "gemm_wgrad_grouped: no products");
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += WG_MAX) {
        WGroupArgs ga;
        ga.n = n - base < WG_MAX ? n - base : WG_MAX;
        long long total = 0;
        for (int i = 0; i < ga.n; ++i) {
            const ptts_wgrad_desc& d = descs[base + i];
            PTTS_REQUIRE(d.A && d.B && d.C && d.M > 0 && d.N > 0 && d.K > 0, "gemm_wgrad_grouped: bad product %d", base + i);
            PTTS_REQUIRE(d.lda >= d.M && d.ldb >= d.N && d.ldc >= d.N && d.lda < (1LL << 31) && d.ldb < (1LL << 31) &&
                         d.ldc < (1LL << 31), "gemm_wgrad_grouped: bad leading dims in product %d", base + i);
            PTTS_REQUIRE(d.in_mode >= 0 && d.in_mode <= 2 && (d.in_mode != PTTS_IN_MASKMUL || d.mask_src) &&
                         ((d.in_scale == nullptr) == (d.in_shift == nullptr)), "gemm_wgrad_grouped: bad transform in product %d", base + i);
            WGroup& w = ga.gr[i];
            w.A = d.A; w.B = d.B; w.C = d.C; w.colsum_b = d.colsum_b;
            w.in_scale = d.in_scale; w.in_shift = d.in_shift; w.mask_src = d.mask_src;
            w.M = d.M; w.N = d.N; w.K = d.K; w.lda = (int)d.lda; w.ldb = (int)d.ldb; w.ldc = (int)d.ldc;
            w.in_mode = d.in_mode; w.alpha = d.alpha;
            ga.it_begin[i] = total;
            total += (long long)((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN) * ((d.K + BK - 1) / BK);
        }
        ga.it_begin[ga.n] = total;
        long long workers = gemm_slots();
        if (workers > total / 8) workers = total / 8;
        if (workers < 1) workers = 1;
        ga.workers = (int)workers;
        hipLaunchKernelGGL(gemm_wgrad_grouped_kernel, dim3(ga.workers), dim3(GEMM_THREADS), 0, st, ga);
        int rc = check_launch("gemm_wgrad_grouped");
        if (rc) return rc;
    }
    return PTTS_OK;
}

extern "C" int ptts_gemm(const float* A, const float* Bm, const float* bias, float* C, int M, int N, int K,
                         int transA, long long lda, long long rows_per_seg, long long seg_stride, int transB,
                         long long ldb, long long ldc, int in_mode, const float* in_scale,
                         const float* in_shift, const float* mask_src, float alpha, int accumulate,
                         const float* out_mask, float* colsum_b, void* stream) {
    PTTS_REQUIRE(A && Bm && C, "gemm: null matrix");
    PTTS_REQUIRE(!colsum_b || (transA == 1 && transB == 0), "gemm: colsum_b needs transA=1, transB=0 (a weight-gradient product)");
    PTTS_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: bad dims M=%d N=%d K=%d", M, N, K);
    PTTS_REQUIRE(rows_per_seg > 0 && rows_per_seg < (1LL << 31) && lda > 0 && ldb > 0 && ldc >= N, "gemm: bad leading dims");
    PTTS_REQUIRE(in_mode >= 0 && in_mode <= 2, "gemm: bad in_mode %d", in_mode);
    PTTS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "gemm: scale/shift must come together");
    PTTS_REQUIRE(in_mode != PTTS_IN_MASKMUL || mask_src, "gemm: MASKMUL needs mask_src");
    PTTS_REQUIRE(!(out_mask && bias), "gemm: out_mask with bias is not defined");
    hipStream_t st = (hipStream_t)stream;
    GemmArgs g;
    g.A = A; g.B = Bm; g.bias = bias; g.C = C; g.M = M; g.N = N; g.K = K;
    g.transA = transA; g.lda = lda; g.rows_per_seg = rows_per_seg; g.seg_stride = seg_stride;
    g.transB = transB; g.ldb = ldb; g.ldc = ldc;
    g.in_mode = in_mode; g.in_scale = in_scale; g.in_shift = in_shift; g.mask_src = mask_src; g.alpha = alpha;
    g.accumulate = accumulate;
    g.out_mask = out_mask; g.out_alpha = alpha;
    g.colsum_b = colsum_b;
    static int prio = -1;
    if (prio < 0) { const char* e = getenv("PTTS_GEMM_PRIO"); prio = e ? atoi(e) : 1; }
    g.prio = prio;

    // thin products (N <= 4 heads, K <= 4 outer products, 1-2 weighted column sums) never reach the MFMA tiles
    if (thin_enabled() && !colsum_b) {
        const int thin = thin_gemm_dispatch(A, Bm, bias, C, M, N, K, transA, lda, rows_per_seg, seg_stride, transB, ldb, ldc,
                                            in_mode, in_scale, in_shift, mask_src, alpha, accumulate, out_mask, st);
        if (thin < 0) return thin;
        if (thin == 1) return PTTS_OK;
    }
    // N slightly above one full-width tile (the critic's 260-wide spectral part): full-width tile + remainder
    if (transA == 0 && N > TBN && N <= 2 * TBN && M >= 2048 && K <= 2048) {
        const int n1 = TBN, n2 = N - TBN;
        const float* B2 = transB == 0 ? Bm + n1 : Bm + (long long)n1 * ldb;
        int rc = ptts_gemm(A, Bm, bias, C, M, n1, K, transA, lda, rows_per_seg, seg_stride, transB, ldb, ldc, in_mode,
                           in_scale, in_shift, mask_src, alpha, accumulate, out_mask, nullptr, stream);
        if (rc) return rc;
        return ptts_gemm(A, B2, bias ? bias + n1 : nullptr, C + n1, M, n2, K, transA, lda, rows_per_seg, seg_stride,
                         transB, ldb, ldc, in_mode, in_scale, in_shift, mask_src, alpha, accumulate,
                         out_mask ? out_mask + n1 : nullptr, nullptr, stream);
    }
    // tall products (M >> N, N in (128, 256]) take the full-width tile kernel: A read once, one round, no atomics
    // (for deep K the stream-K 128x128 kernel below is faster: its two co-resident workgroups per CU overlap better)
    static int tall_kmax = -1;
    if (tall_kmax < 0) { const char* e = getenv("PTTS_TALL_KMAX"); tall_kmax = e ? atoi(e) : 2048; }
    if (transA == 0 && N > 128 && N <= TBN && M >= 2048 && K <= tall_kmax) {
        const int mt = pick_tall_mt(M);
        dim3 tgrid((M + 16 * mt - 1) / (16 * mt)), tblock(TALL_THREADS);
#define PTTS_TALL(TB, MD, MTT) hipLaunchKernelGGL((gemm_tall_kc_kernel<TB, MD, MTT, 16>), tgrid, tblock, 0, st, g)
#define PTTS_TALL_MT(TB, MD)                                          \
        switch (mt) {                                                 \
            case 4: PTTS_TALL(TB, MD, 4); break;                      \
            case 5: PTTS_TALL(TB, MD, 5); break;                      \
            case 6: PTTS_TALL(TB, MD, 6); break;                      \
            case 7: PTTS_TALL(TB, MD, 7); break;                      \
            default: PTTS_TALL(TB, MD, 8); break;                     \
        }
        if (transB == 0) {
            if (in_mode == PTTS_IN_LRELU) { PTTS_TALL_MT(0, PTTS_IN_LRELU) }
            else if (in_mode == PTTS_IN_MASKMUL) { PTTS_TALL_MT(0, PTTS_IN_MASKMUL) }
            else { PTTS_TALL_MT(0, PTTS_IN_NONE) }
        } else {
            if (in_mode == PTTS_IN_LRELU) { PTTS_TALL_MT(1, PTTS_IN_LRELU) }
            else if (in_mode == PTTS_IN_MASKMUL) { PTTS_TALL_MT(1, PTTS_IN_MASKMUL) }
            else { PTTS_TALL_MT(1, PTTS_IN_NONE) }
        }
#undef PTTS_TALL_MT
#undef PTTS_TALL
        return check_launch("gemm_tall");
    }
    const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
    const long long tiles = (long long)tm * tn;
    g.tiles_n = tn;
    g.ksteps = (K + BK - 1) / BK;
    g.iters_total = tiles * g.ksteps;
    // persistent workgroups: two per CU (the second hides the first one's barriers), never less than
    // 8 k-steps of work each
    // stream-K pays when the tile count loads the 512 workgroup slots unevenly AND K is deep enough to amortise
    // the atomic epilogue; otherwise one workgroup per tile (plain stores).
    const long long slots = gemm_slots();
    const double eff_dp = (double)tiles / (double)(((tiles + slots - 1) / slots) * slots);
    long long workers;
    // (atomic combination makes the last bits order-dependent: it is kept for the deep products only, K >= 2048 -- the
    // reference's own batch, 10 x 400 frames, is a reduction over K = 4000 and ran at 3.5 TF on 14 tiles without it)
    if (g.ksteps >= 128 && eff_dp < 0.9 && !deterministic()) {
        workers = slots;
        if (workers > g.iters_total / 8) workers = g.iters_total / 8;
        if (workers < 1) workers = 1;
    } else {
        workers = tiles;
    }
    g.workers = (int)workers;
    const bool split = (g.iters_total % workers != 0) || ((g.iters_total / workers) % g.ksteps != 0);
    if (split && !accumulate) {
        if (zero_f32_2d(C, (size_t)ldc, (size_t)N, (size_t)M, st) != PTTS_OK) return PTTS_ELAUNCH;
    }
    if (colsum_b && zero_f32(colsum_b, (size_t)N, st) != PTTS_OK) return PTTS_ELAUNCH;
    dim3 grid(g.workers), block(GEMM_THREADS);
    const bool conv = seg_stride != 0;
#define PTTS_GEMM_LAUNCH(TA, TB, CV, MD) hipLaunchKernelGGL((gemm_f32_mfma_kernel<TA, TB, CV, MD>), grid, block, 0, st, g)
#define PTTS_GEMM_MODES(TA, TB)                                                  \
    switch (in_mode) {                                                           \
        case PTTS_IN_LRELU: PTTS_GEMM_LAUNCH(TA, TB, 0, PTTS_IN_LRELU); break;   \
        case PTTS_IN_MASKMUL: PTTS_GEMM_LAUNCH(TA, TB, 0, PTTS_IN_MASKMUL); break; \
        default: PTTS_GEMM_LAUNCH(TA, TB, 0, PTTS_IN_NONE); break;               \
    }
    static int use_dma = -1;
    if (use_dma < 0) { const char* e = getenv("PTTS_GEMM_DMA"); use_dma = e ? atoi(e) : 1; }
    // A rows are fetched as 16-byte pieces by the LDS-DMA path: the product must not need masking inside a row
    const bool dma_ok = use_dma && in_mode == PTTS_IN_NONE && transB == 0 && !out_mask && K >= 4 * BK;
    if (dma_ok && transA == 0 && conv) hipLaunchKernelGGL((gemm_dma_kernel<0, 1>), grid, block, 0, st, g);
    else if (dma_ok && transA == 1 && conv) hipLaunchKernelGGL((gemm_dma_kernel<1, 1>), grid, block, 0, st, g);
    else if (dma_ok && transA == 0) hipLaunchKernelGGL((gemm_dma_kernel<0, 0>), grid, block, 0, st, g);
    else if (dma_ok && transA == 1) hipLaunchKernelGGL((gemm_dma_kernel<1, 0>), grid, block, 0, st, g);
    else if (conv && in_mode == PTTS_IN_NONE && transA == 0 && transB == 0) PTTS_GEMM_LAUNCH(0, 0, 1, PTTS_IN_NONE);
    else if (conv && in_mode == PTTS_IN_NONE && transA == 1 && transB == 0) PTTS_GEMM_LAUNCH(1, 0, 1, PTTS_IN_NONE);
    else if (transA == 0 && transB == 0) { PTTS_GEMM_MODES(0, 0) }
    else if (transA == 0 && transB == 1) { PTTS_GEMM_MODES(0, 1) }
    else if (transA == 1 && transB == 0) { PTTS_GEMM_MODES(1, 0) }
    else { PTTS_GEMM_MODES(1, 1) }
#undef PTTS_GEMM_MODES
#undef PTTS_GEMM_LAUNCH
    return check_launch("gemm");
}

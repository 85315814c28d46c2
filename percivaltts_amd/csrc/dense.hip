// Dense products C[M,N] = T(A)[M,K] . B[K,N] (+ bias) with M >> N on the bf16 matrix cores of gfx950, in fp32 arithmetic.
//
// Role on the hot path: every Dense layer of critic and generator (reference networktts.py:59-63, networks_critic.py:86-93,
// modeltts_common.py:77-79) -- forward, backward-data (B = W^T, with the LeakyReLU mask of the layer input fused into the
// store) and the masked forward of the gradient penalty's second-order sweep -- and the LSTM input projections
// (networktts.py:85-96).  The fp32-MFMA "tall" kernel of gemm.hip runs these at 71-77 TF, half of a pipe whose peak is 157;
// here the same sums run on v_mfma_f32_16x16x32_bf16 through the three-way bf16 split of BOTH operands (x = x1 + x2 + x3,
// xi = bf16(remainder), exact; the six products of order >= 2^-16 kept, fp32 accumulation: the arithmetic of split.hip and
// conv2d_mfma.hip, admitted as fp32 by the round-1 verdict).
//
// Operands.  The weights are split ONCE per update into three planes in MFMA-fragment order
//     planes[p][nt][ks][lane][8]:  n = 16 nt + (lane & 15),  k = 32 ks + 8 (lane >> 4) + e,   zero beyond N / K,
// so that a wave fetches a fragment with one coalesced 1-KB load straight from L2 into registers (no LDS, no transform).
// The activations stay fp32 in HBM; a workgroup stages TBM rows x 32 k per step, applies the pending transform of the
// producing layer (LeakyReLU, BatchNorm-affine + LeakyReLU, or the gradient-penalty mask), splits, and writes three bf16
// planes to the LDS (rows of 64 B, the four 16-byte quads XOR-swizzled by bit 2 of the row: conflict-free ds_read_b128).
//
// Tiling.  One workgroup (8 waves) owns TBM x 256 outputs -- all columns of a 256-wide block, so A is read once -- with
// TBM = 16 MT chosen to fill the CUs in whole rounds (M = 25 600 -> 112 -> 229 workgroups); wave w owns columns
// [32 w, 32 w + 32) x all MT row tiles.  The MFMA's first operand is the WEIGHT fragment and its second the activations,
// so a lane's accumulator holds four consecutive output columns of one row: 16-byte stores, no epilogue staging.
// Per k-step and wave: 3 MT ds_read_b128, 6 global 16-byte loads, 12 MT MFMAs.
#include "common.h"
#include <cstdlib>

namespace ptts {
namespace dns {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

constexpr int THREADS = 512;
constexpr int NBLK = 256;          // columns per workgroup
constexpr int BK = 32;

#define DNS_PRODUCTS(X) X(2, 0) X(1, 1) X(0, 2) X(1, 0) X(0, 1) X(0, 0)
// NPL = 3: the six products of the fp32 split; NPL = 1 (ptts_set_bf16_products: BASELINE configs[2]): ONE product of the
// operands' bf16 roundings, fp32 accumulation
#define DNS_PRODUCTS_NPL(NPL, X) do { if (NPL == 3) { X(2, 0) X(1, 1) X(0, 2) X(1, 0) X(0, 1) } X(0, 0) } while (0)

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
// x = h1 + h2 + h3, hi = bf16(remainder), round to nearest even (oracle.np_split3_bf16)
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = pk_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, p1 << 16), rb = b - __builtin_bit_cast(float, p1 & 0xffff0000u);
    p2 = pk_bf16(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, p2 << 16), sb = rb - __builtin_bit_cast(float, p2 & 0xffff0000u);
    p3 = pk_bf16(sa, sb);
}
__device__ __forceinline__ float max_fast(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ------------------------------------------------------------------------------------------------------------
// weight planes.  transposed == 0: B[k][n] = w[k * ldw + n] (w stored [K][N]: the forward product);
//                 transposed == 1: B[k][n] = w[n * ldw + k] (w stored [N][K]: dX = dY . W^T reads W as it lies)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split3_dense_weight_kernel(const float* __restrict__ w, long long ldw, int K, int N,
                                                                  int transposed, u16* __restrict__ planes, int NT, int KS) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;        // (nt, ks, lane)
    if (idx >= (long long)NT * KS * 64) return;
    const int lane = (int)(idx & 63);
    const long long t = idx >> 6;
    const int ks = (int)(t % KS), nt = (int)(t / KS);
    const int n = nt * 16 + (lane & 15), k0 = ks * 32 + (lane >> 4) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        v[e] = (n < N && k < K) ? (transposed ? w[(long long)n * ldw + k] : w[(long long)k * ldw + n]) : 0.f;
    }
    unsigned q[3][4];
#pragma unroll
    for (int h = 0; h < 4; ++h) split3_pair(v[2 * h], v[2 * h + 1], q[0][h], q[1][h], q[2][h]);
    const size_t ps = (size_t)NT * KS * 512;
#pragma unroll
    for (int p = 0; p < 3; ++p)
        *reinterpret_cast<uint4*>(planes + p * ps + (size_t)idx * 8) = make_uint4(q[p][0], q[p][1], q[p][2], q[p][3]);
}

// the same for up to SPLIT_GROUP weights in one launch (blockIdx.y = weight): after an update every Dense kernel of a network needs its
// planes rebuilt (forward and transposed: 18 launches of 4 us per critic step, each with a launch boundary of its own)
constexpr int SPLIT_GROUP = 32;
struct SplitGroupArgs {
    const float* w[SPLIT_GROUP]; u16* planes[SPLIT_GROUP];
    long long ldw[SPLIT_GROUP];
    int K[SPLIT_GROUP], N[SPLIT_GROUP], transposed[SPLIT_GROUP], NT[SPLIT_GROUP], KS[SPLIT_GROUP];
    int rlo[SPLIT_GROUP], rhi[SPLIT_GROUP];       // rows k outside [rlo, rhi) are zero and never read (windows of a frame sequence)
};
__global__ __launch_bounds__(256) void split3_dense_weight_grouped_kernel(SplitGroupArgs a) {
    const int i = blockIdx.y;
    const int NT = a.NT[i], KS = a.KS[i], K = a.K[i], N = a.N[i], transposed = a.transposed[i];
    const int rlo = a.rlo[i], rhi = a.rhi[i];
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;        // (nt, ks, lane)
    if (idx >= (long long)NT * KS * 64) return;
    const float* __restrict__ w = a.w[i];
    u16* __restrict__ planes = a.planes[i];
    const long long ldw = a.ldw[i];
    const int lane = (int)(idx & 63);
    const long long t = idx >> 6;
    const int ks = (int)(t % KS), nt = (int)(t / KS);
    const int n = nt * 16 + (lane & 15), k0 = ks * 32 + (lane >> 4) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        v[e] = (n < N && k < K && k >= rlo && k < rhi) ? (transposed ? w[(long long)n * ldw + k] : w[(long long)k * ldw + n]) : 0.f;
    }
    unsigned q[3][4];
#pragma unroll
    for (int h = 0; h < 4; ++h) split3_pair(v[2 * h], v[2 * h + 1], q[0][h], q[1][h], q[2][h]);
    const size_t ps = (size_t)NT * KS * 512;
#pragma unroll
    for (int p = 0; p < 3; ++p)
        *reinterpret_cast<uint4*>(planes + p * ps + (size_t)idx * 8) = make_uint4(q[p][0], q[p][1], q[p][2], q[p][3]);
}

// ---- frequency-domain context Conv1D: the kernel's transform, straight into the per-frequency planes ----------------------------
// B_f [2 Kh][N] = [Wr_f ; Wi_f],  W^_f[c][n] = sum_k w[k][c][n] e^{-2 pi i f (pl - k) / P}  (rows c < Cin of each half, the rest zero),
// written as the planes ptts_dense_bf16x6_batched reads for frequency f.  A lane owns one 16-byte fragment piece (8 consecutive rows
// of one column) for FCH frequencies: its 8 x KW taps stay in registers, a frequency costs 8 KW FMAs and one split.  (A GEMM with
// the twiddle matrix + a grouped split pass did the same in 0.35 ms per update: 263 MB of fp32 written, read again and split.)
constexpr int WDFT_FCH = 32;
template <int KW>
__global__ __launch_bounds__(256) void conv1d_wdft_planes_kernel(const float* __restrict__ w, const float* __restrict__ tw,
                                                                u16* __restrict__ planes, long long fstride, int NB, int Cin,
                                                                int N, int Kh, int NT, int KS) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;        // (nt, ks, lane)
    if (idx >= (long long)NT * KS * 64) return;
    const int lane = (int)(idx & 63);
    const long long t = idx >> 6;
    const int ks = (int)(t % KS), nt = (int)(t / KS);
    const int n = nt * 16 + (lane & 15), k0 = ks * 32 + (lane >> 4) * 8;
    const int part = k0 >= Kh ? 1 : 0, c0 = k0 - part * Kh;                 // Kh % 8 == 0: the 8 rows lie in one half
    float wr[KW][8];
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e)
            wr[k][e] = (n < N && c0 + e < Cin && k0 < 2 * Kh) ? w[((long long)k * Cin + c0 + e) * N + n] : 0.f;
    const size_t ps = (size_t)NT * KS * 512;
    const int f0 = blockIdx.y * WDFT_FCH;
    for (int f = f0; f < min(NB, f0 + WDFT_FCH); ++f) {
        const float* __restrict__ tr = tw + ((long long)f * 2 + part) * KW;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const float c = tr[k];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(c, wr[k][e], v[e]);
        }
        unsigned q[3][4];
#pragma unroll
        for (int h = 0; h < 4; ++h) split3_pair(v[2 * h], v[2 * h + 1], q[0][h], q[1][h], q[2][h]);
        u16* pf = planes + (long long)f * fstride;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            *reinterpret_cast<uint4*>(pf + p * ps + (size_t)idx * 8) = make_uint4(q[p][0], q[p][1], q[p][2], q[p][3]);
    }
}

// ... and for matrices at REGULAR strides / windows of frame sequences (blockIdx.y = matrix, any number of them in one launch: the
// descriptor form above carries 32 per launch, and the frequency-domain Conv1D splits 256 at a time)
struct SplitStridedArgs {
    const float* w; u16* planes;
    long long stride_w, stride_p, ldw;            // floats, u16 elements, floats
    int K, N, transposed, NT, KS;
    int windows;                                  // 1: matrix z = (b, s) is the window x[b][row_off + s S + k][:] of a frame sequence
    int T, NS, S, row_off, kvalid;
};
__global__ __launch_bounds__(256) void split3_dense_weight_strided_kernel(SplitStridedArgs a) {
    const int z = blockIdx.y;
    const int NT = a.NT, KS = a.KS, K = a.K, N = a.N, transposed = a.transposed;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;        // (nt, ks, lane)
    if (idx >= (long long)NT * KS * 64) return;
    const float* __restrict__ w;
    int rlo = 0, rhi = K;
    if (a.windows) {
        const int b = z / a.NS, sg = z - b * a.NS;
        const int r0 = a.row_off + sg * a.S;                                // source row of window row 0 (may be negative)
        w = a.w + ((long long)b * a.T + r0) * a.ldw;                        // (only rows inside [rlo, rhi) are dereferenced)
        rlo = r0 < 0 ? -r0 : 0;
        rhi = min(a.kvalid, a.T - r0);
    } else {
        w = a.w + (long long)z * a.stride_w;
    }
    u16* __restrict__ planes = a.planes + (long long)z * a.stride_p;
    const long long ldw = a.ldw;
    const int lane = (int)(idx & 63);
    const long long t = idx >> 6;
    const int ks = (int)(t % KS), nt = (int)(t / KS);
    const int n = nt * 16 + (lane & 15), k0 = ks * 32 + (lane >> 4) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        v[e] = (n < N && k < K && k >= rlo && k < rhi) ? (transposed ? w[(long long)n * ldw + k] : w[(long long)k * ldw + n]) : 0.f;
    }
    unsigned q[3][4];
#pragma unroll
    for (int h = 0; h < 4; ++h) split3_pair(v[2 * h], v[2 * h + 1], q[0][h], q[1][h], q[2][h]);
    const size_t ps = (size_t)NT * KS * 512;
#pragma unroll
    for (int p = 0; p < 3; ++p)
        *reinterpret_cast<uint4*>(planes + p * ps + (size_t)idx * 8) = make_uint4(q[p][0], q[p][1], q[p][2], q[p][3]);
}

// The same planes for row-major matrices B[K][N] (not transposed, no windows) through the LDS: a workgroup takes the 32 k-rows of one
// k-step of one matrix, reads them as whole rows with 16-byte loads (a wave per row: 1 KB runs), and builds the fragments -- a lane's
// 8 consecutive k of ONE column -- out of the LDS.  The direct kernel above reads every value with a 4-byte load of its own (8 per lane,
// 256 B per wave and instruction): 38 us for the 32 MB of per-frequency products that the frequency-domain Conv1D hands to its inverse
// transform as the "weight" operand, 3 % of the loop's kernel time.  N % 4 == 0, ldw % 4 == 0, 16-byte aligned source.
constexpr int SPL_ROW = 256 + 2;         // floats per LDS row: rows 8 apart land 16 banks apart
__global__ __launch_bounds__(256) void split3_dense_weight_strided_lds_kernel(SplitStridedArgs a) {
    __shared__ float tile[32 * SPL_ROW];
    const int z = blockIdx.y, ks = blockIdx.x % a.KS, cb = blockIdx.x / a.KS;          // matrix, k-step, block of 256 columns
    const float* __restrict__ w = a.w + (long long)z * a.stride_w;
    u16* __restrict__ planes = a.planes + (long long)z * a.stride_p;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = cb * 256;
    // rows k = 32 ks + r: wave w loads rows w, w + 4, ...; lane: four columns
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = wave + 4 * j, k = ks * 32 + r, n = n0 + 4 * lane;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < a.K && n < a.N) v = *reinterpret_cast<const f32x4*>(w + (long long)k * a.ldw + n);
        float* d = tile + r * SPL_ROW + 4 * lane;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
    const size_t ps = (size_t)a.NT * a.KS * 512;
    const int li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ntl = wave + 4 * j;                                  // column tile inside the block: 0 .. 15
        const int nt = cb * 16 + ntl;
        if (nt >= a.NT) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tile[(8 * lg + e) * SPL_ROW + 16 * ntl + li];
        unsigned q[3][4];
#pragma unroll
        for (int h = 0; h < 4; ++h) split3_pair(v[2 * h], v[2 * h + 1], q[0][h], q[1][h], q[2][h]);
        const size_t idx = ((size_t)nt * a.KS + ks) * 64 + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            *reinterpret_cast<uint4*>(planes + p * ps + idx * 8) = make_uint4(q[p][0], q[p][1], q[p][2], q[p][3]);
    }
}

#ifndef DNS_DEPTH
#define DNS_DEPTH 2        // register stages of the A operand in dense_bf16x6_kernel (1 = the single stage of rounds 2-3; measured 1 / 2 / 3 / 4 / 6:
                           // 25.6 / 25.0 / 25.6 / 26.5 / 29.8 us at 25 600 rows, 82.4 / 75.3 / 80.3 / 83.0 / 86.0 at 76 800: the loads are not what bounds it)
#endif
struct DenseArgs {
    const float* A; const float* mask_src; const float* in_scale; const float* in_shift;
    const u16* planes; const float* bias; const float* out_mask; float* C;
    int M, N, K, NT, KS;
    long long lda, ldc;
    float alpha, out_alpha;
    int accumulate, has_affine, vec_out;
    const float* res; int res_rows; long long ldr;      // residual added in the store: C[m][n] += res[m % res_rows][n] (NULL: none).  accumulate = res == C
    long long bsA, bsP, bsC;          // batched launch (gridDim.z products of one shape): strides of A and C in floats, of the planes in u16
    double* stats;                    // (ptts_dense_bf16x6_stats) per row tile: the column sums and the column sums of squares of what is stored, [gridDim.x][2 N]
    const float* om_scale; const float* om_shift;     // (ptts_dense_bf16x6_bwd_affine) out_mask holds z of a BatchNorm-affine + LeakyReLU input: see the store
};

template <int MODE, bool AFFINE, int MT, int NPL>
__global__ __launch_bounds__(THREADS) void dense_bf16x6_kernel(DenseArgs g) {
    if (gridDim.z > 1) {
        g.A += (long long)blockIdx.z * g.bsA;
        g.planes += (long long)blockIdx.z * g.bsP;
        g.C += (long long)blockIdx.z * g.bsC;
    }
    constexpr int TBM = 16 * MT;
    constexpr int NA = (TBM * 8 + THREADS - 1) / THREADS;     // 16-byte quads (4 k) per lane and k-step
    constexpr int ROWS = NA * THREADS / 8;                    // staged rows incl. the pad rows the idle lanes of the last slot write
    constexpr int PL = ROWS * BK;                             // elements of one plane of a stage
    constexpr bool MASK = MODE == PTTS_IN_MASKMUL;
    __shared__ __attribute__((aligned(16))) u16 As[2][NPL * PL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int m0 = blockIdx.x * TBM, n0 = blockIdx.y * NBLK;
    const int KS = g.KS;
    const bool wave_live = n0 + 32 * wave < g.N;              // a wave whose columns lie beyond N only stages

    // ---- this lane's staging slots: quad q = tid + 512 j -> row q >> 3, k-quad q & 7 (the same for every j).  No branches
    // in the k-loop: rows beyond the tile or M read row 0 and are zeroed by a select, and so are the quads beyond K.
    const int kq = tid & 7;
    const float* pa[NA]; const float* pm[NA]; bool oka[NA]; int dst[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int row = (tid + j * THREADS) >> 3;
        oka[j] = row < TBM && m0 + row < g.M;
        const long long base = (long long)(oka[j] ? m0 + row : 0) * g.lda + 4 * kq;
        pa[j] = g.A + base;
        pm[j] = MASK ? g.mask_src + base : nullptr;
        dst[j] = row * BK + ((((kq >> 1) ^ ((row >> 1) & 2))) << 3) + (kq & 1) * 4;
    }
    struct Stage { f32x4 va[NA], vm[MASK ? NA : 1], sc, sh; bool kok; };
    auto load_a = [&](int s, Stage& sg) {
        const int k = s * BK + 4 * kq;
        sg.kok = k < g.K;                                     // K % 4 == 0: a quad is all inside or all outside
        const int ko = sg.kok ? s * BK : -4 * kq;             // (outside: the row's first quad, discarded)
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            sg.va[j] = *reinterpret_cast<const f32x4*>(pa[j] + ko);
            if (MASK) sg.vm[j] = *reinterpret_cast<const f32x4*>(pm[j] + ko);
        }
        if (AFFINE) {
            sg.sc = *reinterpret_cast<const f32x4*>(g.in_scale + (sg.kok ? k : 0));
            sg.sh = *reinterpret_cast<const f32x4*>(g.in_shift + (sg.kok ? k : 0));
        }
    };
    auto commit = [&](const Stage& sg, u16* as) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            f32x4 a = sg.va[j];
            if (MODE == PTTS_IN_LRELU) {
                if (AFFINE) a = a * sg.sc + sg.sh;
                if (!(oka[j] && sg.kok)) a = z4;
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = max_fast(a[e], g.alpha * a[e]);
            } else if (MASK) {
                if (!(oka[j] && sg.kok)) a = z4;
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = a[e] * (sg.vm[j][e] > 0.f ? 1.f : g.alpha);
            } else {
                if (!(oka[j] && sg.kok)) a = z4;
            }
            u16* d = as + dst[j];
            if (NPL == 3) {
                unsigned a1, a2, a3, b1, b2, b3;
                split3_pair(a[0], a[1], a1, a2, a3);
                split3_pair(a[2], a[3], b1, b2, b3);
                *reinterpret_cast<u32x2*>(d) = (u32x2){a1, b1};
                *reinterpret_cast<u32x2*>(d + PL) = (u32x2){a2, b2};
                *reinterpret_cast<u32x2*>(d + 2 * PL) = (u32x2){a3, b3};
            } else {
                *reinterpret_cast<u32x2*>(d) = (u32x2){pk_bf16(a[0], a[1]), pk_bf16(a[2], a[3])};      // the one rounding of bf16 products
            }
        }
    };

    // ---- weight fragments of this wave's two column tiles, straight from global memory
    const size_t ps = (size_t)g.NT * KS * 512;
    const u16* wp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) wp[j] = g.planes + (((size_t)((n0 >> 4) + 2 * (wave_live ? wave : 0) + j) * KS) * 64 + lane) * 8;
    auto load_w = [&](int s, bf16x8 (&wf)[2][NPL]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < NPL; ++p) wf[j][p] = *reinterpret_cast<const bf16x8*>(wp[j] + p * ps + (size_t)s * 512);
    };

    f32x4 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // activation fragment of row tile i: row 16 i + li, quad lg (swizzle = bit 2 of li), plane p: + p PL
    const int boff = li * BK + ((lg ^ ((li >> 1) & 2)) << 3);
    auto mfma_rows = [&](const u16* as, const bf16x8 (&wf)[2][NPL], int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            bf16x8 bf[NPL];
#pragma unroll
            for (int p = 0; p < NPL; ++p) bf[p] = *reinterpret_cast<const bf16x8*>(as + p * PL + boff + i * 16 * BK);
#define DNS_MM(PA, PW)                                                                                    \
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][PW < NPL ? PW : 0], bf[PA < NPL ? PA : 0], acc[i][0], 0, 0, 0);    \
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][PW < NPL ? PW : 0], bf[PA < NPL ? PA : 0], acc[i][1], 0, 0, 0);
            DNS_PRODUCTS_NPL(NPL, DNS_MM);
#undef DNS_MM
        }
    };

    // One basic block per k-step: the staging of step s+1 (transform, split, LDS stores into the other buffer) and the loads
    // of step s+2 sit between the MFMAs of step s.  (A wave whose columns lie beyond N runs the same code on the first
    // tile's planes: the MFMAs are cheap next to a divergent barrier structure, and nothing of it is stored.)
    // DEPTH register stages of A: the loads of k-step s + DEPTH are issued while step s is multiplied.  The kernel's time is linear in M
    // from 25 600 rows up (tools/dense_ramp_probe.py: 1.0 us per 1000 rows, 130 TF) and hardly moves with DEPTH: what bounds a k-step is
    // the matrix pipe -- TWO waves per SIMD x 84 six-product MFMAs x 16 cycles = 2 700 cycles at the ~1.8 GHz it holds under that load.
    constexpr int DEPTH = DNS_DEPTH;
    Stage sg[DEPTH];
    bf16x8 wc[2][NPL], wn[2][NPL];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) load_a(j, sg[j]);
    load_w(0, wc);
    commit(sg[0], As[0]);
    load_a(DEPTH, sg[0]);
    __syncthreads();
    // The two waves of a SIMD leave every barrier together; with the staging at the same place of their instruction streams
    // both would do vector work at the same time and leave the matrix pipe idle.  Waves 0-3 (one per SIMD) stage after two
    // row tiles, waves 4-7 after MT - 2.
    auto step = [&](int s, int cut, Stage& nx) {             // nx: the stage that holds step s + 1 (slot (s + 1) % DEPTH), refilled with step s + 1 + DEPTH
        const u16* as = As[s & 1];
        load_w(s + 1 < KS ? s + 1 : s, wn);
        mfma_rows(as, wc, 0, cut);
        commit(nx, As[(s + 1) & 1]);                          // the other buffer: every wave left it at the last barrier
        load_a(s + 1 + DEPTH, nx);
        mfma_rows(as, wc, cut, MT);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < NPL; ++p) wc[j][p] = wn[j][p];
        __syncthreads();
    };
    auto run = [&](int cut) {
        int s = 0;
        for (; s + DEPTH <= KS; s += DEPTH) {                 // (s % DEPTH == 0: the stage indices are compile-time)
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) step(s + j, cut, sg[(j + 1) % DEPTH]);
        }
#pragma unroll
        for (int j = 0; j < DEPTH; ++j)
            if (s + j < KS) step(s + j, cut, sg[(j + 1) % DEPTH]);
    };
    if (wave < 4) run(2); else run(MT - 2);
    if (!wave_live) return;
    // ---- store: lane (li, lg) of acc[i][j] holds row m0 + 16 i + li, columns n0 + 32 wave + 16 j + 4 lg .. + 3
    const bool interior = m0 + TBM <= g.M;                    // no row guards: the mask / old-value loads go out together
    if (!g.vec_out) {
        // N or ldc no multiple of 4 (the 65-bin spectral head): element-wise stores
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = m0 + 16 * i + li;
                if (m >= g.M) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = n0 + 32 * wave + 16 * j + 4 * lg + e;
                    if (n >= g.N) continue;
                    const long long off = (long long)m * g.ldc + n;
                    float v = acc[i][j][e] + (g.bias ? g.bias[n] : 0.f);
                    if (g.out_mask) v *= g.out_mask[off] > 0.f ? 1.f : g.out_alpha;
                    if (g.res) { int mr = m; while (mr >= g.res_rows) mr -= g.res_rows; v += g.res[(long long)mr * g.ldr + n]; }
                    g.C[off] = v;
                }
            }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + 32 * wave + 16 * j + 4 * lg;
        if (n >= g.N) continue;                               // N % 4 == 0
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) bv = *reinterpret_cast<const f32x4*>(g.bias + n);
        const long long off0 = (long long)(m0 + li) * g.ldc + n;
        // row of the residual: m modulo res_rows (a product shared by k stacked evaluations is added to each of its k row blocks)
        auto res_at = [&](int m) { int mr = m; while (mr >= g.res_rows) mr -= g.res_rows; return g.res + (long long)mr * g.ldr + n; };
        f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
        // ptts_dense_bf16x6_bwd_affine: the product is da = dy . W^T of a layer whose input was lrelu(sc z + sh), z = out_mask.  What is
        // stored is dz = da lrelu'(sc z + sh) sc; the tile's column sums are those of da lrelu'(.) z (ssum: the affine's dscale) and of
        // da lrelu'(.) (ssq: its dshift) -- ptts_affine_act_bwd's arithmetic, without its pass over da and z
        f32x4 osc = {1.f, 1.f, 1.f, 1.f}, osh = {0.f, 0.f, 0.f, 0.f};
        if (g.om_scale) { osc = *reinterpret_cast<const f32x4*>(g.om_scale + n); osh = *reinterpret_cast<const f32x4*>(g.om_shift + n); }
        auto aff_store = [&](f32x4& v, const f32x4& z) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gd = v[e] * ((z[e] * osc[e] + osh[e]) > 0.f ? 1.f : g.out_alpha);
                ssum[e] += gd * z[e]; ssq[e] += gd;
                v[e] = gd * osc[e];
            }
        };
        if (interior) {
            f32x4 mk[MT], old[MT];
            if (g.out_mask) {
#pragma unroll
                for (int i = 0; i < MT; ++i) mk[i] = *reinterpret_cast<const f32x4*>(g.out_mask + off0 + (long long)16 * i * g.ldc);
            }
            if (g.res) {
#pragma unroll
                for (int i = 0; i < MT; ++i) old[i] = *reinterpret_cast<const f32x4*>(res_at(m0 + 16 * i + li));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                f32x4 v = acc[i][j] + bv;
                if (g.om_scale) {
                    aff_store(v, mk[i]);
                } else if (g.out_mask) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * (mk[i][e] > 0.f ? 1.f : g.out_alpha);
                }
                if (g.res) v += old[i];
                if (g.stats && !g.om_scale) { ssum += v; ssq += v * v; }
                *reinterpret_cast<f32x4*>(g.C + off0 + (long long)16 * i * g.ldc) = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = m0 + 16 * i + li;
                if (m >= g.M) continue;
                const long long off = off0 + (long long)16 * i * g.ldc;
                f32x4 v = acc[i][j] + bv;
                if (g.out_mask) {
                    const f32x4 mk = *reinterpret_cast<const f32x4*>(g.out_mask + off);
                    if (g.om_scale) {
                        aff_store(v, mk);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] * (mk[e] > 0.f ? 1.f : g.out_alpha);
                    }
                }
                if (g.res) v += *reinterpret_cast<const f32x4*>(res_at(m));
                if (g.stats && !g.om_scale) { ssum += v; ssq += v * v; }
                *reinterpret_cast<f32x4*>(g.C + off) = v;
            }
        }
        if (g.stats) {
            // the tile's column sums: a lane holds MT of the TBM rows of its four columns; the 16 lanes li of a group hold the others
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { ssum[e] += __shfl_xor(ssum[e], m, 64); ssq[e] += __shfl_xor(ssq[e], m, 64); }
            if (li == 0) {
                double* row = g.stats + (size_t)blockIdx.x * 2 * g.N;
#pragma unroll
                for (int e = 0; e < 4; ++e) { row[n + e] = (double)ssum[e]; row[g.N + n + e] = (double)ssq[e]; }
            }
        }
    }
}

// rows per workgroup (multiple of 16, 64..128) that best fills 256 CUs in whole rounds
static int pick_mt(int M, int col_blocks) {
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("PTTS_DENSE_MT"); forced = e ? atoi(e) : 0; }
    if (forced >= 4 && forced <= 8) return forced;
    int best = 8; double best_eff = -1.0;
    for (int mt = 4; mt <= 8; ++mt) {
        const long long blocks = (long long)((M + 16 * mt - 1) / (16 * mt)) * col_blocks;
        const long long rounds = (blocks + 255) / 256;
        const double eff = (double)M * col_blocks / (double)(rounds * 256 * 16 * mt);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = mt; }
    }
    return best;
}


// ------------------------------------------------------------------------------------------------------------
// Weight gradient  dW[Kin][N] += T(A)[M][Kin]^T . dY[M][N],  db[N] += sum_m dY[m][:]   (reduction over the M frames).
// Both operands are activations: each 32-row slab of A (with the pending transform of the layer input) and of dY is split
// on its way into the LDS as row-major planes [32 m][128 columns] and read TRANSPOSED by ds_read_b64_tr_b16, so that a
// lane gets 8 consecutive m of its column (the MFMA's k index) -- the construction of conv2d_mfma.hip's weight gradient.
// A row is eight 32-byte segments (one 16-column MFMA tile each); segment t of row r lies at position
// (t + r + 4 (r >> 3)) & 7: the four rows a 16-lane group reads, and the two groups of a half-wave, hit disjoint banks.
// Output tile 128 x 128 per workgroup (8 waves: 2 x 4 MFMA tiles each), the frames of a tile shared out over
// 256 / tiles workgroups; every workgroup stores its partial tile (and 128 bias sums, from the staged dY values of the first
// tile row) as a row of the caller's workspace, and dense_wgrad_reduce_kernel -- one grouped launch for a batch of products -- adds
// the rows in a fixed order into the gradient buffers (a first version flushed with fp32 atomics from every workgroup).
// ------------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef s16x4 __attribute__((address_space(3)))* lds_s16x4_ptr;
__device__ __forceinline__ bf16x4 tr_read(const u16* p) {
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p)));
}
__device__ __forceinline__ bf16x8 cat8(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

struct WgradArgs {
    const float* A; const float* dY; const float* mask_src; const float* in_scale; const float* in_shift;
    float* partials;                         // [tile][split][WPART]: the workgroups' partial tiles (+ 128 bias sums)
    int Kin, N, M;
    long long lda, ldb;
    float alpha;
    int tiles_n, steps_total, split;         // 128-column tiles along N; 32-row steps of M; workgroups per tile
};

constexpr int WT = 128;                       // tile edge (columns of A x columns of dY)
constexpr int WROW = WT;                      // elements per staged row of a plane
constexpr int WPL = 32 * WROW;                // elements of one plane of one operand
constexpr int WPART = WT * WT + WT;           // floats of a workgroup's partial row: the tile and the bias sums

template <int MODE, bool AFFINE, int NPL>
__global__ __launch_bounds__(THREADS) void dense_wgrad_bf16x6_kernel(WgradArgs g) {
    constexpr bool MASK = MODE == PTTS_IN_MASKMUL;
    extern __shared__ __attribute__((aligned(16))) u16 lds_w[];          // [buffer][A planes 0..2 | dY planes 0..2]: 96 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int tile = blockIdx.y, tk = tile / g.tiles_n, tn = tile - tk * g.tiles_n;
    const int k0 = tk * WT, n0 = tn * WT;
    // this workgroup's share of the 32-row steps
    const int per = (g.steps_total + g.split - 1) / g.split;
    const int s_begin = blockIdx.x * per, s_end = min(g.steps_total, s_begin + per);
    const int nsteps = max(0, s_end - s_begin);           // (a share without steps still writes its row of zeros)

    // ---- staging slots: float4 q = tid + 512 j (j = 0, 1) -> row q >> 5, columns 4 (q & 31) ..; the same columns for both j
    const int c4 = tid & 31;
    const bool a_ok = k0 + 4 * c4 < g.Kin, d_ok = n0 + 4 * c4 < g.N;     // Kin, N multiples of 4
    const float* pa = g.A + (a_ok ? k0 + 4 * c4 : 0);
    const float* pm = MASK ? g.mask_src + (a_ok ? k0 + 4 * c4 : 0) : nullptr;
    const float* pd = g.dY + (d_ok ? n0 + 4 * c4 : 0);
    int dst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (tid + j * THREADS) >> 5;
        dst[j] = row * WROW + ((((c4 >> 2) + row + 4 * (row >> 3)) & 7) << 4) + (c4 & 3) * 4;
    }
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (AFFINE && a_ok) { sc = *reinterpret_cast<const f32x4*>(g.in_scale + k0 + 4 * c4); sh = *reinterpret_cast<const f32x4*>(g.in_shift + k0 + 4 * c4); }
    struct Stage { f32x4 va[2], vm[MASK ? 2 : 1], vd[2]; bool ok[2]; };
    auto load = [&](int s, Stage& sg) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = (s_begin + s) * 32 + ((tid + j * THREADS) >> 5);
            sg.ok[j] = s < nsteps && m < g.M;
            const long long r = sg.ok[j] ? m : 0;
            sg.va[j] = *reinterpret_cast<const f32x4*>(pa + r * g.lda);
            if (MASK) sg.vm[j] = *reinterpret_cast<const f32x4*>(pm + r * g.lda);
            sg.vd[j] = *reinterpret_cast<const f32x4*>(pd + r * g.ldb);
        }
    };
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    auto commit = [&](const Stage& sg, u16* ls) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 a = sg.va[j], d = sg.vd[j];
            if (MODE == PTTS_IN_LRELU) {
                if (AFFINE) a = a * sc + sh;
                if (!(sg.ok[j] && a_ok)) a = z4;
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = max_fast(a[e], g.alpha * a[e]);
            } else if (MASK) {
                if (!(sg.ok[j] && a_ok)) a = z4;
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = a[e] * (sg.vm[j][e] > 0.f ? 1.f : g.alpha);
            } else {
                if (!(sg.ok[j] && a_ok)) a = z4;
            }
            if (!(sg.ok[j] && d_ok)) d = z4;
            bsum += d;
            u16* q = ls + dst[j];
            if (NPL == 3) {
                unsigned a1, a2, a3, b1, b2, b3;
                split3_pair(a[0], a[1], a1, a2, a3);
                split3_pair(a[2], a[3], b1, b2, b3);
                *reinterpret_cast<u32x2*>(q) = (u32x2){a1, b1};
                *reinterpret_cast<u32x2*>(q + WPL) = (u32x2){a2, b2};
                *reinterpret_cast<u32x2*>(q + 2 * WPL) = (u32x2){a3, b3};
                split3_pair(d[0], d[1], a1, a2, a3);
                split3_pair(d[2], d[3], b1, b2, b3);
                q += 3 * WPL;
                *reinterpret_cast<u32x2*>(q) = (u32x2){a1, b1};
                *reinterpret_cast<u32x2*>(q + WPL) = (u32x2){a2, b2};
                *reinterpret_cast<u32x2*>(q + 2 * WPL) = (u32x2){a3, b3};
            } else {
                *reinterpret_cast<u32x2*>(q) = (u32x2){pk_bf16(a[0], a[1]), pk_bf16(a[2], a[3])};
                *reinterpret_cast<u32x2*>(q + 3 * WPL) = (u32x2){pk_bf16(d[0], d[1]), pk_bf16(d[2], d[3])};
            }
        }
    };

    // ---- MFMA tiles of this wave: A-column tiles 2 (wave & 3) + {0, 1}, dY-column tiles 4 (wave >> 2) + {0..3}
    // transposed-read address of lane (li, lg) in tile t: row 8 lg + (li >> 2) (+ 4 for the second half), columns 4 (li & 3)
    auto tr_off = [&](int t, int half) {
        const int row = 8 * lg + (li >> 2) + 4 * half;
        return row * WROW + (((t + row + 4 * (row >> 3)) & 7) << 4) + (li & 3) * 4;
    };
    int ao[2][2], bo[4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 2; ++i) ao[i][h] = tr_off(2 * (wave & 3) + i, h);
#pragma unroll
        for (int i = 0; i < 4; ++i) bo[i][h] = 3 * WPL + tr_off(4 * (wave >> 2) + i, h);
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // one k-step of MFMAs on buffer BUF (a compile-time index: every LDS address is a lane base + an immediate); the dY
    // fragments are read once and serve both A-column tiles; `stage` (the next step's transform / split / LDS stores and the
    // loads of the step after) runs between the two tile rows in waves 0-3 and in front of them in waves 4-7
    auto mfma_tile_row = [&](const u16* ls, int i, const bf16x8 (&bf)[4][NPL]) {
        bf16x8 af[NPL];
#pragma unroll
        for (int p = 0; p < NPL; ++p) af[p] = cat8(tr_read(ls + p * WPL + ao[i][0]), tr_read(ls + p * WPL + ao[i][1]));
#define DNS_MM(PA, PW)                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[PA < NPL ? PA : 0], bf[j][PW < NPL ? PW : 0], acc[i][j], 0, 0, 0);
        DNS_PRODUCTS_NPL(NPL, DNS_MM);
#undef DNS_MM
    };
    auto read_bf = [&](const u16* ls, bf16x8 (&bf)[4][NPL]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int p = 0; p < NPL; ++p) bf[j][p] = cat8(tr_read(ls + p * WPL + bo[j][0]), tr_read(ls + p * WPL + bo[j][1]));
    };

    Stage sg;
    load(0, sg);
    commit(sg, lds_w);
    load(1, sg);
    __syncthreads();
    u16* const buf0 = lds_w;
    u16* const buf1 = lds_w + 6 * WPL;
    auto step = [&](int s, const u16* cur, u16* nxt, bool stage_first) {
        bf16x8 bf[4][NPL];
        if (stage_first) { commit(sg, nxt); load(s + 2, sg); }       // (rows of a step beyond the share are staged as zeros)
        read_bf(cur, bf);
        mfma_tile_row(cur, 0, bf);
        if (!stage_first) { commit(sg, nxt); load(s + 2, sg); }
        mfma_tile_row(cur, 1, bf);
        __syncthreads();
    };
    // The two waves of a SIMD leave every barrier together: waves 0-3 run half of their MFMAs before the staging of the next
    // step, waves 4-7 stage first -- one wave's vector work under the other's matrix work.  Two steps per trip: the
    // buffers are compile-time.
    if (wave < 4) {
        int s = 0;
        for (; s + 1 < nsteps; s += 2) { step(s, buf0, buf1, false); step(s + 1, buf1, buf0, false); }
        if (s < nsteps) step(s, buf0, buf1, false);
    } else {
        int s = 0;
        for (; s + 1 < nsteps; s += 2) { step(s, buf0, buf1, true); step(s + 1, buf1, buf0, true); }
        if (s < nsteps) step(s, buf0, buf1, true);
    }

    // ---- the workgroup's partial tile, row (tile, split index) of the partials: [128 k][128 n] | 128 column sums of dY.
    // lane (li, lg) of acc[i][j] holds rows k = 32 (wave & 3) + 16 i + 4 lg + r, column 64 (wave >> 2) + 16 j + li
    float* out = g.partials + ((size_t)tile * g.split + blockIdx.x) * WPART;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = 64 * (wave >> 2) + 16 * j + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 32 * (wave & 3) + 16 * i + 4 * lg + r;
                out[k * WT + n] = acc[i][j][r];
            }
        }
    {
        // 16 lanes (tid >> 5) share the column group c4: sum through the LDS (the planes are dead)
        float* red = reinterpret_cast<float*>(lds_w);
        __syncthreads();
        *reinterpret_cast<f32x4*>(red + tid * 4) = bsum;
        __syncthreads();
        if (tid < WT) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) v += red[(q * 32 + (tid >> 2)) * 4 + (tid & 3)];
            out[WT * WT + tid] = v;
        }
    }
}


// partial rows -> gradient buffers, several products per launch.  Thread = one element of a tile, summed over the split
// workgroups in a fixed order; added into C / the bias gradient with one fp32 atomic per element (products of one backward
// pass share gradient buffers: first- and second-order sweeps of a layer).
constexpr int DR_MAX = 16;
struct DwReduceArgs {
    int n;
    int blk_begin[DR_MAX + 1];
    const float* partials[DR_MAX]; float* C[DR_MAX]; float* colsum[DR_MAX];
    int split[DR_MAX], tiles_n[DR_MAX], Kin[DR_MAX], N[DR_MAX];
    long long ldc[DR_MAX];
};
constexpr int DR_BPT = (WPART + 255) / 256;       // blocks per tile (the last one holds the bias sums)
__global__ __launch_bounds__(256) void dense_wgrad_reduce_kernel(DwReduceArgs a) {
    int gi = 0;
    while ((int)blockIdx.x >= a.blk_begin[gi + 1]) ++gi;
    const int b = blockIdx.x - a.blk_begin[gi];
    const int tile = b / DR_BPT, e = (b - tile * DR_BPT) * 256 + threadIdx.x;
    if (e >= WPART) return;
    const int S = a.split[gi];
    const float* p = a.partials[gi] + (size_t)tile * S * WPART + e;
    // eight rows in flight per thread (a fixed order still: the same sums in every run): the plain loop waited for every load in turn
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, v5 = 0.f, v6 = 0.f, v7 = 0.f;
    int s = 0;
    for (; s + 8 <= S; s += 8) {
        const float* q = p + (size_t)s * WPART;
        const float t0 = q[0], t1 = q[(size_t)WPART], t2 = q[(size_t)2 * WPART], t3 = q[(size_t)3 * WPART];
        const float t4 = q[(size_t)4 * WPART], t5 = q[(size_t)5 * WPART], t6 = q[(size_t)6 * WPART], t7 = q[(size_t)7 * WPART];
        v0 += t0; v1 += t1; v2 += t2; v3 += t3; v4 += t4; v5 += t5; v6 += t6; v7 += t7;
    }
    for (; s < S; ++s) v0 += p[(size_t)s * WPART];
    const float v = ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7));
    const int tk = tile / a.tiles_n[gi], tn = tile - tk * a.tiles_n[gi];
    if (e < WT * WT) {
        const int k = tk * WT + (e >> 7), n = tn * WT + (e & (WT - 1));
        if (k < a.Kin[gi] && n < a.N[gi]) atomicAdd(a.C[gi] + (long long)k * a.ldc[gi] + n, v);
    } else if (tk == 0 && a.colsum[gi]) {
        const int n = tn * WT + (e - WT * WT);
        if (n < a.N[gi]) atomicAdd(a.colsum[gi] + n, v);
    }
}

}  // namespace dns
}  // namespace ptts

using namespace ptts;
using namespace ptts::dns;

// bytes of the three planes of a [K][N] operand (N rounded up to 256 columns, K to 32)
extern "C" size_t ptts_dense_planes_bytes(int N, int K) {
    if (N <= 0 || K <= 0) return 0;
    const size_t NT = (size_t)((N + NBLK - 1) / NBLK) * (NBLK / 16), KS = (size_t)(K + BK - 1) / BK;
    return 3 * NT * KS * 512 * sizeof(u16);
}

// planes of B[K][N]: transposed == 0 reads w as [K][N] (row stride ldw), transposed == 1 as [N][K]
extern "C" int ptts_split3_dense_weight(const float* w, long long ldw, int K, int N, int transposed, void* planes, void* stream) {
    PTTS_REQUIRE(w && planes, "split3_dense_weight: null pointer");
    PTTS_REQUIRE(K > 0 && N > 0 && ldw >= (transposed ? K : N), "split3_dense_weight: bad dims K=%d N=%d ldw=%lld", K, N, ldw);
    const int NT = (N + NBLK - 1) / NBLK * (NBLK / 16), KS = (K + BK - 1) / BK;
    const long long total = (long long)NT * KS * 64;
    hipLaunchKernelGGL(split3_dense_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       w, ldw, K, N, transposed, (u16*)planes, NT, KS);
    return check_launch("split3_dense_weight");
}

extern "C" int ptts_split3_dense_weight_grouped(const ptts_dense_split_desc* descs, int n, void* stream) {
    PTTS_REQUIRE(descs && n > 0, "split3_dense_weight_grouped: nothing to split");
    for (int base = 0; base < n; base += SPLIT_GROUP) {
        SplitGroupArgs a;
        const int m = n - base < SPLIT_GROUP ? n - base : SPLIT_GROUP;
        long long most = 0;
        for (int i = 0; i < m; ++i) {
            const ptts_dense_split_desc& d = descs[base + i];
            PTTS_REQUIRE(d.w && d.planes, "split3_dense_weight_grouped: null pointer (weight %d)", base + i);
            PTTS_REQUIRE(d.K > 0 && d.N > 0 && d.ldw >= (d.transposed ? d.K : d.N), "split3_dense_weight_grouped: bad dims K=%d N=%d ldw=%lld (weight %d)", d.K, d.N, d.ldw, base + i);
            a.w[i] = d.w; a.planes[i] = (u16*)d.planes; a.ldw[i] = d.ldw; a.K[i] = d.K; a.N[i] = d.N; a.transposed[i] = d.transposed;
            a.rlo[i] = 0; a.rhi[i] = d.K;
            a.NT[i] = (d.N + NBLK - 1) / NBLK * (NBLK / 16); a.KS[i] = (d.K + BK - 1) / BK;
            const long long total = (long long)a.NT[i] * a.KS[i] * 64;
            if (total > most) most = total;
        }
        hipLaunchKernelGGL(split3_dense_weight_grouped_kernel, dim3((unsigned)((most + 255) / 256), (unsigned)m), dim3(256), 0, (hipStream_t)stream, a);
    }
    return check_launch("split3_dense_weight_grouped");
}

// n weights of one shape at regular strides (w + i * stride_w floats -> planes + i * stride_planes_bytes): the operands of the
// batched products, without a descriptor array on the host side
extern "C" int ptts_split3_dense_weight_strided(const float* w, long long stride_w, void* planes, long long stride_planes_bytes, int n,
                                                long long ldw, int K, int N, int transposed, void* stream) {
    PTTS_REQUIRE(w && planes && n > 0 && n <= 65535, "split3_dense_weight_strided: nothing to split (or more than 65535 matrices)");
    PTTS_REQUIRE(K > 0 && N > 0 && ldw >= (transposed ? K : N), "split3_dense_weight_strided: bad dims K=%d N=%d ldw=%lld", K, N, ldw);
    PTTS_REQUIRE(stride_planes_bytes % 16 == 0, "split3_dense_weight_strided: the planes' stride must keep 16-byte alignment");
    SplitStridedArgs a;
    a.w = w; a.planes = (u16*)planes; a.stride_w = stride_w; a.stride_p = stride_planes_bytes / 2; a.ldw = ldw;
    a.K = K; a.N = N; a.transposed = transposed;
    a.NT = (N + NBLK - 1) / NBLK * (NBLK / 16); a.KS = (K + BK - 1) / BK;
    a.windows = 0; a.T = a.NS = a.S = a.row_off = a.kvalid = 0;
    const long long total = (long long)a.NT * a.KS * 64;
    static int lds_form = -1;
    if (lds_form < 0) { const char* e = getenv("PTTS_SPLIT_LDS"); lds_form = e ? atoi(e) : 1; }
    if (lds_form && !transposed && N % 4 == 0 && ldw % 4 == 0 && stride_w % 4 == 0 && ((uintptr_t)w & 15) == 0) {
        const unsigned cbn = (unsigned)((N + 255) / 256);
        hipLaunchKernelGGL(split3_dense_weight_strided_lds_kernel, dim3((unsigned)a.KS * cbn, (unsigned)n), dim3(256), 0, (hipStream_t)stream, a);
        return check_launch("split3_dense_weight_strided");
    }
    hipLaunchKernelGGL(split3_dense_weight_strided_kernel, dim3((unsigned)((total + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("split3_dense_weight_strided");
}

// Planes of WINDOWS of frame sequences x [B][T][C]: matrix z = b NS + s is the P rows x[b][row_off + s S + k][:], k < P, zero for rows
// outside [0, T) and for k >= kvalid (the segments of an overlap-save convolution: a window of S + KW - 1 input frames per S output
// frames; a block of S gradient frames zero-padded to P).  planes + z stride_planes_bytes, ptts_dense_planes_bytes(C, P) each.
extern "C" int ptts_split3_frame_windows(const float* x, int B, int T, int C, int NS, int S, int row_off, int P, int kvalid,
                                         void* planes, long long stride_planes_bytes, void* stream) {
    PTTS_REQUIRE(x && planes && B > 0 && T > 0 && C > 0 && NS > 0 && S > 0 && P > 0 && kvalid > 0 && kvalid <= P, "split3_frame_windows: bad arguments");
    PTTS_REQUIRE((long long)B * NS <= 65535 && stride_planes_bytes % 16 == 0, "split3_frame_windows: more than 65535 windows, or a planes stride off 16-byte alignment");
    SplitStridedArgs a;
    a.w = x; a.planes = (u16*)planes; a.stride_w = 0; a.stride_p = stride_planes_bytes / 2; a.ldw = C;
    a.K = P; a.N = C; a.transposed = 0;
    a.NT = (C + NBLK - 1) / NBLK * (NBLK / 16); a.KS = (P + BK - 1) / BK;
    a.windows = 1; a.T = T; a.NS = NS; a.S = S; a.row_off = row_off; a.kvalid = kvalid;
    const long long total = (long long)a.NT * a.KS * 64;
    hipLaunchKernelGGL(split3_dense_weight_strided_kernel, dim3((unsigned)((total + 255) / 256), (unsigned)(B * NS)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("split3_frame_windows");
}

// 1 when ptts_dense_bf16x6 takes the shape
extern "C" int ptts_dense_bf16x6_supported(int M, int N, int K, long long lda, long long ldc) {
    (void)ldc;                                    // any N / ldc: rows that are no multiple of 4 floats are stored element-wise
    return (M > 0 && N > 0 && K > 0 && K % 4 == 0 && lda % 4 == 0 && K <= 65536 &&
            (long long)((N + NBLK - 1) / NBLK) <= 65535) ? 1 : 0;
}

// C[M,N] (+)= T(A)[M,K] . B (+ bias), then C *= (out_mask > 0 ? 1 : alpha) -- the contract of ptts_gemm with transA = 0 and
// B given as the planes of ptts_split3_dense_weight.  in_mode / in_scale / in_shift / mask_src / alpha as in ptts_gemm.
extern "C" int ptts_dense_bf16x6_res(const float* A, const void* planes, const float* bias, float* C, int M, int N, int K,
                                     long long lda, long long ldc, int in_mode, const float* in_scale, const float* in_shift,
                                     const float* mask_src, float alpha, const float* res, int res_rows, long long ldr,
                                     const float* out_mask, void* stream);
extern "C" int ptts_dense_bf16x6(const float* A, const void* planes, const float* bias, float* C, int M, int N, int K,
                                 long long lda, long long ldc, int in_mode, const float* in_scale, const float* in_shift,
                                 const float* mask_src, float alpha, int accumulate, const float* out_mask, void* stream) {
    return ptts_dense_bf16x6_res(A, planes, bias, C, M, N, K, lda, ldc, in_mode, in_scale, in_shift, mask_src, alpha,
                                 accumulate ? C : nullptr, M, ldc, out_mask, stream);
}

// ... + res[m % res_rows][n] (row stride ldr) in the store: the product of a concat part that several stacked evaluations share (the
// critic's context branch, computed once at B rows) joins each of the k B-row blocks of the stacked product without an add pass of its
// own.  res == C, res_rows >= M is the plain accumulate of ptts_dense_bf16x6.
static int dense_launch(const float* A, const void* planes, const float* bias, float* C, int M, int N, int K,
                        long long lda, long long ldc, int in_mode, const float* in_scale, const float* in_shift,
                        const float* mask_src, float alpha, const float* res, int res_rows, long long ldr,
                        const float* out_mask, double* stats, int stats_capacity_rows, int* stats_rows_out, void* stream);
static thread_local const float* g_om_scale = nullptr;        // (set around dense_launch by ptts_dense_bf16x6_bwd_affine)
static thread_local const float* g_om_shift = nullptr;

extern "C" int ptts_dense_bf16x6_res(const float* A, const void* planes, const float* bias, float* C, int M, int N, int K,
                                     long long lda, long long ldc, int in_mode, const float* in_scale, const float* in_shift,
                                     const float* mask_src, float alpha, const float* res, int res_rows, long long ldr,
                                     const float* out_mask, void* stream) {
    return dense_launch(A, planes, bias, C, M, N, K, lda, ldc, in_mode, in_scale, in_shift, mask_src, alpha, res, res_rows, ldr, out_mask,
                        nullptr, 0, nullptr, stream);
}

// The product of a Dense layer that feeds a BatchNormalization layer (pFC, reference networktts.py:59-63): the launch also leaves, per
// row tile, the column sums and the column sums of squares of what it stores -- stats[*nrows_out][2 N] doubles -- which
// ptts_bn_finalize_partials finishes: no statistics pass over the [M, N] activation (ptts_colstats: two launches, 16 us at 25 600 x 256).
extern "C" int ptts_dense_bf16x6_stats(const float* A, const void* planes, const float* bias, float* C, int M, int N, int K,
                                       long long lda, long long ldc, int in_mode, const float* in_scale, const float* in_shift,
                                       float alpha, double* stats, int capacity_rows, int* nrows_out, void* stream) {
    PTTS_REQUIRE(stats && nrows_out && capacity_rows > 0, "dense_bf16x6_stats: no room for the sums");
    return dense_launch(A, planes, bias, C, M, N, K, lda, ldc, in_mode, in_scale, in_shift, nullptr, alpha, nullptr, 0, 0, nullptr,
                        stats, capacity_rows, nrows_out, stream);
}
// Backward-data product of a Dense layer whose input was lrelu(scale z + shift) (a BatchNormalization in front: pFC, networktts.py:59-63):
// C = dz = (A . B) lrelu'(scale z + shift) scale with A = dy, B = W^T (planes of the transposed kernel), and per row tile the column sums
// of (A . B) lrelu'(.) z and of (A . B) lrelu'(.) -- the gradients of scale and shift -- as stats[*nrows_out][2 N] doubles
// (ptts_partial_rows_sum adds the rows).  Replaces the product + ptts_affine_act_bwd's pass over da and z.
extern "C" int ptts_dense_bf16x6_bwd_affine(const float* A, const void* planes, float* C, int M, int N, int K, long long lda, long long ldc,
                                            const float* z, const float* scale, const float* shift, float alpha,
                                            double* stats, int capacity_rows, int* nrows_out, void* stream) {
    PTTS_REQUIRE(z && scale && shift && stats && nrows_out && capacity_rows > 0, "dense_bf16x6_bwd_affine: null pointer");
    PTTS_REQUIRE((((uintptr_t)scale | (uintptr_t)shift) & 15) == 0, "dense_bf16x6_bwd_affine: scale / shift must be 16-byte aligned");
    g_om_scale = scale; g_om_shift = shift;
    const int rc = dense_launch(A, planes, nullptr, C, M, N, K, lda, ldc, PTTS_IN_NONE, nullptr, nullptr, nullptr, alpha, nullptr, 0, 0, z,
                                stats, capacity_rows, nrows_out, stream);
    g_om_scale = g_om_shift = nullptr;
    return rc;
}
// rows of sums ptts_dense_bf16x6_stats writes for an [M, N] product (the caller's capacity)
extern "C" int ptts_dense_bf16x6_stats_rows(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    const int mt = pick_mt(M, (N + NBLK - 1) / NBLK);
    return (M + 16 * mt - 1) / (16 * mt);
}

static int dense_launch(const float* A, const void* planes, const float* bias, float* C, int M, int N, int K,
                        long long lda, long long ldc, int in_mode, const float* in_scale, const float* in_shift,
                        const float* mask_src, float alpha, const float* res, int res_rows, long long ldr,
                        const float* out_mask, double* stats, int stats_capacity_rows, int* stats_rows_out, void* stream) {
    const int accumulate = res != nullptr;
    PTTS_REQUIRE(!res || (res_rows > 0 && ldr >= N && M <= 8LL * res_rows), "dense_bf16x6: bad residual (rows %d, ldr %lld)", res_rows, ldr);
    PTTS_REQUIRE(A && planes && C, "dense_bf16x6: null matrix");
    PTTS_REQUIRE(ptts_dense_bf16x6_supported(M, N, K, lda, ldc), "dense_bf16x6: unsupported shape M=%d N=%d K=%d lda=%lld ldc=%lld (K and lda must be multiples of 4)", M, N, K, lda, ldc);
    PTTS_REQUIRE(lda >= K && ldc >= N, "dense_bf16x6: bad leading dims");
    PTTS_REQUIRE(in_mode >= 0 && in_mode <= 2, "dense_bf16x6: bad in_mode %d", in_mode);
    PTTS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "dense_bf16x6: scale/shift must come together");
    PTTS_REQUIRE(in_mode != PTTS_IN_MASKMUL || mask_src, "dense_bf16x6: MASKMUL needs mask_src");
    PTTS_REQUIRE(in_mode == PTTS_IN_LRELU || !in_scale, "dense_bf16x6: scale/shift need PTTS_IN_LRELU");
    PTTS_REQUIRE(!(out_mask && bias), "dense_bf16x6: out_mask with bias is not defined");
    PTTS_REQUIRE(alpha >= 0.f && alpha <= 1.f, "dense_bf16x6: LeakyReLU slope %g outside [0, 1]", alpha);
    const bool vec_out = N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0) &&
                         (!out_mask || ((uintptr_t)out_mask & 15) == 0) && (!res || (((uintptr_t)res & 15) == 0 && ldr % 4 == 0));
    PTTS_REQUIRE(((uintptr_t)A & 15) == 0 && (!mask_src || ((uintptr_t)mask_src & 15) == 0) &&
                 (!in_scale || (((uintptr_t)in_scale | (uintptr_t)in_shift) & 15) == 0), "dense_bf16x6: A, its mask and scale/shift must be 16-byte aligned");
    DenseArgs g;
    g.A = A; g.mask_src = mask_src; g.in_scale = in_scale; g.in_shift = in_shift; g.planes = (const u16*)planes; g.bias = bias;
    g.out_mask = out_mask; g.C = C; g.M = M; g.N = N; g.K = K;
    g.NT = (N + NBLK - 1) / NBLK * (NBLK / 16); g.KS = (K + BK - 1) / BK;
    g.lda = lda; g.ldc = ldc; g.alpha = alpha; g.out_alpha = alpha; g.accumulate = accumulate; g.has_affine = in_scale != nullptr; g.vec_out = vec_out;
    g.res = res; g.res_rows = res ? res_rows : 1; g.ldr = ldr;
    g.bsA = g.bsP = g.bsC = 0;
    const int cb = (N + NBLK - 1) / NBLK;
    const int mt = pick_mt(M, cb);
    const dim3 grid((unsigned)((M + 16 * mt - 1) / (16 * mt)), (unsigned)cb);
    g.stats = stats;
    g.om_scale = g_om_scale; g.om_shift = g_om_shift;
    if (stats) {
        PTTS_REQUIRE(vec_out, "dense_bf16x6_stats: N and ldc must be multiples of 4, C (and bias) 16-byte aligned");
        PTTS_REQUIRE(stats_capacity_rows >= (int)grid.x, "dense_bf16x6_stats: room for %d rows of sums, %d needed", stats_capacity_rows, (int)grid.x);
        *stats_rows_out = (int)grid.x;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool one = ptts::bf16_products();
#define DNS_L(MODE, AFF, MT) do { if (one) hipLaunchKernelGGL((dense_bf16x6_kernel<MODE, AFF, MT, 1>), grid, dim3(THREADS), 0, st, g); \
                                  else hipLaunchKernelGGL((dense_bf16x6_kernel<MODE, AFF, MT, 3>), grid, dim3(THREADS), 0, st, g); } while (0)
#define DNS_M(MT)                                                                \
    do {                                                                         \
        if (in_mode == PTTS_IN_LRELU) { if (in_scale) DNS_L(PTTS_IN_LRELU, true, MT); else DNS_L(PTTS_IN_LRELU, false, MT); } \
        else if (in_mode == PTTS_IN_MASKMUL) DNS_L(PTTS_IN_MASKMUL, false, MT);  \
        else DNS_L(PTTS_IN_NONE, false, MT);                                     \
    } while (0)
    switch (mt) {
        case 4: DNS_M(4); break;
        case 5: DNS_M(5); break;
        case 6: DNS_M(6); break;
        case 7: DNS_M(7); break;
        default: DNS_M(8); break;
    }
#undef DNS_M
#undef DNS_L
    return check_launch("dense_bf16x6");
}

// planes of [Wr_f ; Wi_f] [2 Kh][N] for f = 0 .. NB-1 (each ptts_dense_planes_bytes(N, 2 Kh) long) from the kernel w [KW][Cin][N] and
// the twiddle rows tw [(f, part)][KW] = cos / -sin(2 pi f (pl - k) / P)
extern "C" int ptts_conv1d_freq_kernel_planes(const float* w, const float* tw, void* planes, int NB, int KW, int Cin, int N, int Kh,
                                              void* stream) {
    PTTS_REQUIRE(w && tw && planes && NB > 0 && Cin > 0 && N > 0, "conv1d_freq_kernel_planes: bad arguments");
    PTTS_REQUIRE(Kh >= Cin && Kh % 8 == 0, "conv1d_freq_kernel_planes: Kh=%d must be a multiple of 8 and >= Cin=%d", Kh, Cin);
    const int NT = (N + NBLK - 1) / NBLK * (NBLK / 16), KS = (2 * Kh + BK - 1) / BK;
    const long long total = (long long)NT * KS * 64;
    const long long fstride = (long long)(ptts_dense_planes_bytes(N, 2 * Kh) / sizeof(u16));
    const dim3 grid((unsigned)((total + 255) / 256), (unsigned)((NB + WDFT_FCH - 1) / WDFT_FCH));
#define WDFT(KWv) hipLaunchKernelGGL(conv1d_wdft_planes_kernel<KWv>, grid, dim3(256), 0, (hipStream_t)stream, w, tw, (u16*)planes, fstride, NB, Cin, N, Kh, NT, KS)
    switch (KW) {
        case 3: WDFT(3); break;
        case 5: WDFT(5); break;
        case 7: WDFT(7); break;
        case 9: WDFT(9); break;
        case 11: WDFT(11); break;
        case 21: WDFT(21); break;
        default: set_error("conv1d_freq_kernel_planes: KW=%d not instantiated (3, 5, 7, 9, 11, 21)", KW); return PTTS_EINVAL;
    }
#undef WDFT
    return check_launch("conv1d_freq_kernel_planes");
}

// nbatch products C_z[M,N] = A_z[M,K] . B_z (+ bias) of ONE shape in one launch (blockIdx.z = z): A_z = A + z strideA, C_z = C + z strideC
// (floats; strideA = 0: the same left operand for every product), B_z = the planes at planes + z stride_planes_bytes.  No input
// transform, no masks.  planes_count: 3 = fp32 arithmetic (six products), 1 = one bf16 product -- explicit here, whatever
// ptts_set_bf16_products says.  What the frequency-domain context Conv1D is made of (ops._C1FFT): DFT, per-frequency products, inverse DFT.
extern "C" int ptts_dense_bf16x6_batched(const float* A, long long strideA, const void* planes, long long stride_planes_bytes,
                                         const float* bias, float* C, long long strideC, int nbatch, int M, int N, int K,
                                         long long lda, long long ldc, int planes_count, void* stream) {
    PTTS_REQUIRE(A && planes && C && nbatch > 0 && nbatch <= 65535, "dense_bf16x6_batched: bad arguments");
    PTTS_REQUIRE(ptts_dense_bf16x6_supported(M, N, K, lda, ldc), "dense_bf16x6_batched: unsupported shape M=%d N=%d K=%d lda=%lld ldc=%lld", M, N, K, lda, ldc);
    PTTS_REQUIRE(lda >= K && ldc >= N && planes_count >= 1 && planes_count <= 3 && planes_count != 2, "dense_bf16x6_batched: bad leading dims / planes");
    PTTS_REQUIRE(stride_planes_bytes % 16 == 0 && strideA % 4 == 0 && ((uintptr_t)A & 15) == 0, "dense_bf16x6_batched: strides of A / planes must keep 16-byte alignment");
    const bool vec_out = N % 4 == 0 && ldc % 4 == 0 && strideC % 4 == 0 && ((uintptr_t)C & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0);
    DenseArgs g;
    g.A = A; g.mask_src = nullptr; g.in_scale = nullptr; g.in_shift = nullptr; g.planes = (const u16*)planes; g.bias = bias;
    g.out_mask = nullptr; g.C = C; g.M = M; g.N = N; g.K = K;
    g.NT = (N + NBLK - 1) / NBLK * (NBLK / 16); g.KS = (K + BK - 1) / BK;
    g.lda = lda; g.ldc = ldc; g.alpha = 0.f; g.out_alpha = 0.f; g.accumulate = 0; g.has_affine = 0; g.vec_out = vec_out;
    g.res = nullptr; g.res_rows = 1; g.ldr = 0;
    g.bsA = strideA; g.bsP = stride_planes_bytes / 2; g.bsC = strideC;
    g.stats = nullptr; g.om_scale = g.om_shift = nullptr;
    const int cb = (N + NBLK - 1) / NBLK;
    // rows per workgroup: every workgroup of a product reads that product's planes, so fewer row tiles = fewer reads of the right
    // operand, which is most of the traffic of these small-M products
    int mt = 8; double best_eff = -1.0;
    {
        static int forced = -1;
        if (forced < 0) { const char* e = getenv("PTTS_DENSE_BATCHED_MT"); forced = e ? atoi(e) : 0; }
        for (int m = 8; m >= 4; --m) {          // least row padding, then the taller tile (measured: 134 / 183 us against 163 / 213 for the
            const int tiles = (M + 16 * m - 1) / (16 * m);      // round-filling choice of pick_mt on the per-frequency products)
            const double eff = (double)M / (double)(tiles * 16 * m);
            if (eff > best_eff + 1e-9) { best_eff = eff; mt = m; }
        }
        if (forced >= 4 && forced <= 8) mt = forced;
    }
    const dim3 grid((unsigned)((M + 16 * mt - 1) / (16 * mt)), (unsigned)cb, (unsigned)nbatch);
    hipStream_t st = (hipStream_t)stream;
    const bool one = planes_count == 1;
#define DNS_B(MT) do { if (one) hipLaunchKernelGGL((dense_bf16x6_kernel<PTTS_IN_NONE, false, MT, 1>), grid, dim3(THREADS), 0, st, g); \
                       else hipLaunchKernelGGL((dense_bf16x6_kernel<PTTS_IN_NONE, false, MT, 3>), grid, dim3(THREADS), 0, st, g); } while (0)
    switch (mt) {
        case 4: DNS_B(4); break;
        case 5: DNS_B(5); break;
        case 6: DNS_B(6); break;
        case 7: DNS_B(7); break;
        default: DNS_B(8); break;
    }
#undef DNS_B
    return check_launch("dense_bf16x6_batched");
}

// 1 when ptts_dense_wgrad_bf16x6 takes the shape
extern "C" int ptts_dense_wgrad_bf16x6_supported(int Kin, int N, int M, long long lda, long long ldb) {
    return (Kin > 0 && N > 0 && M > 0 && Kin % 4 == 0 && N % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
            (long long)((Kin + WT - 1) / WT) * ((N + WT - 1) / WT) <= 65535) ? 1 : 0;
}

namespace {
int wgrad_split(int Kin, int N, int M) {
    const int tiles = ((Kin + WT - 1) / WT) * ((N + WT - 1) / WT), steps = (M + 31) / 32;
    int split = (256 + tiles - 1) / tiles;                 // about one workgroup per CU
    static int env_split = -1;                              // measurement hook (tools/dense_split_probe.py)
    if (env_split < 0) { const char* e = getenv("PTTS_DENSE_WGRAD_SPLIT"); env_split = e ? atoi(e) : 0; }
    if (env_split > 0) split = env_split;
    if (split > steps) split = steps;
    return split < 1 ? 1 : split;
}
}  // namespace

extern "C" size_t ptts_dense_wgrad_workspace_bytes(int Kin, int N, int M) {
    if (Kin <= 0 || N <= 0 || M <= 0) return 0;
    const size_t tiles = (size_t)((Kin + WT - 1) / WT) * ((N + WT - 1) / WT);
    return tiles * wgrad_split(Kin, N, M) * WPART * sizeof(float);
}

// Stage 1 of a weight-gradient product dW[Kin,N] = T(A)[M,Kin]^T . dY[M,N] (+ column sums of dY): every workgroup's partial
// tile as a row of `workspace` ([tile][split][128*128 + 128] floats; *split_out = workgroups per tile).  Stage 2,
// ptts_dense_wgrad_reduce_grouped, adds the rows of several products into their gradient buffers in one launch.
extern "C" int ptts_dense_wgrad_bf16x6_partials(const float* A, const float* dY, const float* mask_src, const float* in_scale,
                                                const float* in_shift, void* workspace, size_t workspace_bytes, int* split_out,
                                                int Kin, int N, int M, long long lda, long long ldb, int in_mode, float alpha,
                                                void* stream) {
    PTTS_REQUIRE(A && dY && workspace && split_out, "dense_wgrad_bf16x6: null pointer");
    PTTS_REQUIRE(ptts_dense_wgrad_bf16x6_supported(Kin, N, M, lda, ldb), "dense_wgrad_bf16x6: unsupported shape Kin=%d N=%d M=%d lda=%lld ldb=%lld", Kin, N, M, lda, ldb);
    PTTS_REQUIRE(lda >= Kin && ldb >= N, "dense_wgrad_bf16x6: bad leading dims");
    PTTS_REQUIRE(in_mode >= 0 && in_mode <= 2, "dense_wgrad_bf16x6: bad in_mode %d", in_mode);
    PTTS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "dense_wgrad_bf16x6: scale/shift must come together");
    PTTS_REQUIRE(in_mode != PTTS_IN_MASKMUL || mask_src, "dense_wgrad_bf16x6: MASKMUL needs mask_src");
    PTTS_REQUIRE(in_mode == PTTS_IN_LRELU || !in_scale, "dense_wgrad_bf16x6: scale/shift need PTTS_IN_LRELU");
    PTTS_REQUIRE(alpha >= 0.f && alpha <= 1.f, "dense_wgrad_bf16x6: LeakyReLU slope %g outside [0, 1]", alpha);
    PTTS_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)dY & 15) == 0 && (!mask_src || ((uintptr_t)mask_src & 15) == 0) &&
                 (!in_scale || (((uintptr_t)in_scale | (uintptr_t)in_shift) & 15) == 0), "dense_wgrad_bf16x6: operands must be 16-byte aligned");
    const size_t need = ptts_dense_wgrad_workspace_bytes(Kin, N, M);
    if (workspace_bytes < need) { set_error("dense_wgrad_bf16x6: workspace %zu < %zu", workspace_bytes, need); return PTTS_EWORKSPACE; }
    WgradArgs g;
    g.A = A; g.dY = dY; g.mask_src = mask_src; g.in_scale = in_scale; g.in_shift = in_shift; g.partials = (float*)workspace;
    g.Kin = Kin; g.N = N; g.M = M; g.lda = lda; g.ldb = ldb; g.alpha = alpha;
    const int tiles_k = (Kin + WT - 1) / WT;
    g.tiles_n = (N + WT - 1) / WT;
    g.steps_total = (M + 31) / 32;
    const int tiles = tiles_k * g.tiles_n;
    g.split = wgrad_split(Kin, N, M);
    *split_out = g.split;
    const dim3 grid((unsigned)g.split, (unsigned)tiles);
    hipStream_t st = (hipStream_t)stream;
    constexpr size_t lds = (size_t)2 * 6 * WPL * sizeof(u16);
    const bool one = ptts::bf16_products();
#define DNS_W(MODE, AFF)                                                                                                  \
    do {                                                                                                                  \
        static bool attr = false;                                                                                         \
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_wgrad_bf16x6_kernel<MODE, AFF, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                     (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_wgrad_bf16x6_kernel<MODE, AFF, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; } \
        if (one) hipLaunchKernelGGL((dense_wgrad_bf16x6_kernel<MODE, AFF, 1>), grid, dim3(THREADS), lds, st, g);          \
        else hipLaunchKernelGGL((dense_wgrad_bf16x6_kernel<MODE, AFF, 3>), grid, dim3(THREADS), lds, st, g);              \
    } while (0)
    if (in_mode == PTTS_IN_LRELU) { if (in_scale) DNS_W(PTTS_IN_LRELU, true); else DNS_W(PTTS_IN_LRELU, false); }
    else if (in_mode == PTTS_IN_MASKMUL) DNS_W(PTTS_IN_MASKMUL, false);
    else DNS_W(PTTS_IN_NONE, false);
#undef DNS_W
    return check_launch("dense_wgrad_bf16x6_partials");
}

extern "C" int ptts_dense_wgrad_reduce_grouped(const ptts_dense_wgrad_reduce_desc* descs, int n, void* stream) {
    PTTS_REQUIRE(descs && n > 0, "dense_wgrad_reduce_grouped: nothing to reduce");
    for (int base = 0; base < n; base += DR_MAX) {
        DwReduceArgs a;
        a.n = n - base < DR_MAX ? n - base : DR_MAX;
        int blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            const ptts_dense_wgrad_reduce_desc& d = descs[base + i];
            PTTS_REQUIRE(d.partials && d.C && d.split > 0 && d.Kin > 0 && d.N > 0 && d.ldc >= d.N, "dense_wgrad_reduce_grouped: bad product %d", base + i);
            a.blk_begin[i] = blocks;
            a.tiles_n[i] = (d.N + WT - 1) / WT;
            blocks += ((d.Kin + WT - 1) / WT) * a.tiles_n[i] * DR_BPT;
            a.partials[i] = d.partials; a.C[i] = d.C; a.colsum[i] = d.colsum_b; a.split[i] = d.split; a.Kin[i] = d.Kin; a.N[i] = d.N; a.ldc[i] = d.ldc;
        }
        a.blk_begin[a.n] = blocks;
        hipLaunchKernelGGL(dense_wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
        int rc = check_launch("dense_wgrad_reduce_grouped");
        if (rc) return rc;
    }
    return PTTS_OK;
}

// C[Kin,N] += T(A)[M,Kin]^T . dY[M,N] and, with colsum_b, colsum_b[N] += column sums of dY -- both stages for one product.
extern "C" int ptts_dense_wgrad_bf16x6(const float* A, const float* dY, const float* mask_src, const float* in_scale,
                                       const float* in_shift, float* C, float* colsum_b, void* workspace, size_t workspace_bytes,
                                       int Kin, int N, int M, long long lda, long long ldb, long long ldc, int in_mode,
                                       float alpha, void* stream) {
    PTTS_REQUIRE(C && ldc >= N, "dense_wgrad_bf16x6: bad output");
    int split = 0;
    int rc = ptts_dense_wgrad_bf16x6_partials(A, dY, mask_src, in_scale, in_shift, workspace, workspace_bytes, &split, Kin, N, M,
                                              lda, ldb, in_mode, alpha, stream);
    if (rc) return rc;
    ptts_dense_wgrad_reduce_desc d;
    d.partials = (const float*)workspace; d.split = split; d.Kin = Kin; d.N = N; d.ldc = ldc; d.C = C; d.colsum_b = colsum_b;
    return ptts_dense_wgrad_reduce_grouped(&d, 1, stream);
}

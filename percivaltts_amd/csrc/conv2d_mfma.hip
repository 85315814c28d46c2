// 2D convolution (time x frequency, NHWC, C = 4 -> 4, 5x5) on the bf16 matrix cores of gfx950, in fp32 arithmetic.
//
// Role on the hot path: the 4 -> 4 channel layers of the critic's 8-deep Conv2D stack (reference
// networks_critic.py:66-68) -- forward, the masked forward of the gradient penalty's second-order sweep, the
// backward-data pass (with the LeakyReLU mask of the previous layer fused into the store) and the weight gradient.
// The packed-FMA stencil of conv2d.hip is issue-bound at 0.17 of the HBM roofline; here the same sums run on
// v_mfma_f32_16x16x32_bf16 through the three-way bf16 split of BOTH operands (x = x1 + x2 + x3, xi = bf16(remainder),
// exact: 3 x 8 = 24 significant bits; the six products of order >= 2^-16 are kept and accumulated in fp32 -- the same
// arithmetic as split.hip's context Conv1D, admitted as fp32 arithmetic by the round-1 verdict).
//
// Mapping (per kernel row kt): the 5 x 4 taps of a kernel row form a banded (Toeplitz) block
//     A[m = (so, co)][k = (j, ci)] = w[kt][kf = j - so][ci][co]   (0 <= j - so < 5, else 0),
// m: 4 output bins x 4 output channels, k: 8 input bins x 4 input channels (62.5 % of the block is non-zero); the
// activations are the B operand, B[k][n = time row]: a lane's 8 k-values are 8 CONSECUTIVE bf16 of the NHWC row (2 bins x
// 4 channels), no im2col.  D[m][n]: lane (n = lane & 15, q = lane >> 4) holds output pixel (row of lane n, bin 4g + q), its
// four output channels in the four accumulator registers -- one 16-byte store per lane.
// The weight gradient runs the reduction over the pixels through the same instruction (K = 16 rows x 2 bin groups per
// step) with both operands read transposed out of the row-major LDS planes by ds_read_b64_tr_b16.
//
// Tiles: (utterance, 16 time rows, block of <= 17 bin groups).  The staged tile always has the geometry of a 17-group
// block (72 bins) and the dilation is a template parameter, so that every LDS address of the inner loops is one base
// register plus an immediate: the first version spent as many vector instructions on addresses as the matrix pipe spent
// cycles on the products (rocprofv3 SQ_INSTS_VALU, gpurun_out/c2m_pmc).
//
// bf16 storage (BASELINE configs[2]): the same kernels with NPL = 1 plane per operand -- activations / gradients may lie in
// HBM as bf16 (8 bytes per pixel) or fp32, are rounded to bf16 once on their way into the LDS, the kernel's bf16 copy of
// the weights is the single table plane, ONE product per MFMA position instead of six, fp32 accumulation, and the result is
// stored as bf16 or fp32 as the caller asks.  Weight gradients stay fp32.
#include "common.h"
#include <cstdlib>
#include <algorithm>

namespace ptts {
namespace c2m {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

constexpr int THREADS = 256;
constexpr int NP = 3;          // bf16 planes per fp32 operand
constexpr int C = 4;           // channels in and out
constexpr int KF = 5;
constexpr int KT = 5;
constexpr int GPB = 17;        // bin groups (of 4 bins) per block: F = 65 is one block
constexpr int NPART = KT * KF * 16 + 4 + 8;   // row of partial sums: the layout of conv2d.hip's backward

// six products (activation plane, weight plane), smallest first; with one plane per operand (bf16 arithmetic) the last one only
#define C2M_PRODUCTS(X) X(2, 0) X(1, 1) X(0, 2) X(1, 0) X(0, 1) X(0, 0)
#define C2M_PRODUCTS_NPL(NPL, X) do { if (NPL == 3) { X(2, 0) X(1, 1) X(0, 2) X(1, 0) X(0, 1) } X(0, 0) } while (0)

// dtype flags of the entry points (bit set = that tensor is bf16 in HBM)
constexpr int DT_IN = 1, DT_OUT = 2, DT_DY = 4;

// measurement hooks (tools/conv2d_mfma_probe.py): bit 0 skip the staging, bit 1 skip the MFMA phase, bit 2 skip the
// stores, bit 3 write s_memtime stamps of the phases of every workgroup to dbg_buf[block][8], bit 6 no barriers (garbage: timing
// only).  (A bit-5 hook that made the MFMA operands up in registers was removed: its run-time branch sat inside the kernel-row
// loop, and the compiler executed its 61 constant moves per 24 MFMAs on the normal path too and issued the LDS reads of row
// kt + 1 AFTER the MFMAs of row kt instead of before them.)
constexpr int DBG_NOSTAGE = 1, DBG_NOMFMA = 2, DBG_NOSTORE = 4, DBG_STAMPS = 8, DBG_NOBAR = 64;
// bit 7 (tests/test_ops_gpu.py): the multiplying waves of the wave-specialised forward wait for a count that never comes, with a
// short bound -- the hand-off's time-out path on demand (wrong results by construction, and the device status word set)
constexpr int DBG_FORCE_TIMEOUT = 128;
constexpr int DBG_FOUR_WAVES = 1 << 16;      // host side only: launch the four-wave form although the wave-specialised one is the default (A/B in tests)
#ifndef C2M_WS_NMAX
#define C2M_WS_NMAX 0      // groups per pass of the wave-specialised kernel's multiplying waves: 0 = 5 (4 with mask values)
#endif
#ifndef C2M_WS_FLAGS
#define C2M_WS_FLAGS 1
#endif
#ifndef C2M_BW_ROT
#define C2M_BW_ROT 0       // fused backward kernel: offset of the wave that takes the odd weight-gradient pair against the one that takes the odd bin group
#endif
#ifndef C2M_BW_NMAX
#define C2M_BW_NMAX 0      // fused backward kernel: bin groups per pass of its convolution (0: fwd_piece's default, 4 with mask values)
#endif
#ifndef C2M_BW_WREG
#define C2M_BW_WREG 1      // fused backward kernel: the convolution's table fragments from global memory / registers (1) or from a copy in the LDS (0)
#endif
#ifndef C2M_PROBE_COMMIT
#define C2M_PROBE_COMMIT 0   // (timing probes of the staging waves' commit: 1 = no LDS writes, 2 = no conversions; garbage results)
#endif
#ifndef C2M_PROBE_STAMPS
#define C2M_PROBE_STAMPS 0   // (probe build: s_memtime stamps inside the wave-specialised forward kernel's piece loop; needs a debug buffer of 16 words per workgroup)
#endif
#ifndef C2M_PROBE_BREUSE
#define C2M_PROBE_BREUSE 0   // (timing probe, tools/ab_file.sh: garbage results)
#endif
#ifndef C2M_PIN_WF
// 1: the wave-specialised kernels' table fragments pinned in registers by an empty asm.  The table is `const __restrict__`, and the
// compiler RE-LOADS the lane's fifteen fragments from global memory (L2) in every pass instead of keeping 60 registers -- which is the
// faster program: pinned, the same box measured forward 22.6 -> 25.0 us, backward data 23.6 -> 29.8, fused backward 42.5 -> 45.1
// (tools/ab_c2m.sh C2M_PIN_WF=0 against the pinned build, three alternating pairs).  Default 0.
#define C2M_PIN_WF 0
#endif
static int g_dbg = 0;
static unsigned long long* g_dbg_buf = nullptr;
__device__ __forceinline__ void stamp(unsigned long long* buf, int dbg, int slot) {
    if ((dbg & DBG_STAMPS) && threadIdx.x == 0) buf[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memtime();
}

// x = h1 + h2 + h3, hi = bf16(remainder), round to nearest even (the oracle's np_split3_bf16).  Written out pair-wise: one
// v_cvt_pk_bf16_f32 per pair and plane, the pair widened again by a shift and a mask -- the generic vector conversions cost
// 30 instructions per 4 values where this takes 22.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = pk_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, p1 << 16), rb = b - __builtin_bit_cast(float, p1 & 0xffff0000u);
    p2 = pk_bf16(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, p2 << 16), sb = rb - __builtin_bit_cast(float, p2 & 0xffff0000u);
    p3 = pk_bf16(sa, sb);
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split3(f32x4 v, bf16x4& h1, bf16x4& h2, bf16x4& h3) {
    unsigned a1, a2, a3, b1, b2, b3;
    split3_pair(v[0], v[1], a1, a2, a3);
    split3_pair(v[2], v[3], b1, b2, b3);
    h1 = __builtin_bit_cast(bf16x4, (u32x2){a1, b1}); h2 = __builtin_bit_cast(bf16x4, (u32x2){a2, b2}); h3 = __builtin_bit_cast(bf16x4, (u32x2){a3, b3});
}
__device__ __forceinline__ bf16x8 cat(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
// max(a, b) for finite operands in one instruction (fmaxf canonicalises its operands first)
__device__ __forceinline__ float max_fast(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ------------------------------------------------------------------------------------------------------------
// Operand tables of a 5x5 4->4 kernel, three bf16 planes.  The MFMA A-operand fragment of kernel row kt is the banded block
//   transposed == 0 (forward):        A[(so,co)][(j,ci)] = w[kt][j - so][ci][co]
//   transposed == 1 (backward data):  A[(so,ci)][(j,co)] = w[KT-1-kt][KF-1-(j - so)][ci][co]   (dx = conv(dy, flipped w^T))
// lane (li = lane & 15, lg = lane >> 4): m = li -> so = li >> 2, oc = li & 3; element e: j = 2 lg + (e >> 2), ic = e & 3 --
// i.e. the eight elements of a lane are TWO consecutive taps kf = 2 lg - so, + 1 of one (kt, oc) row, four ic each.  The
// table therefore stores every (kt, plane, oc) row once, zero-padded to the taps -3 .. 7:
//   tab[kt][plane][oc][kf + 3 (11 slots)][ic (4)]          (5280 bytes instead of 15 KB of per-lane fragments)
// and a lane reads its fragment as 16 bytes at slot 2 lg - so + 3 (8-byte aligned).
// One launch builds both tables: block 0 the forward one, block 1 the transposed one.
// ------------------------------------------------------------------------------------------------------------
constexpr int TSLOTS = 11;                       // taps -3 .. 7
constexpr int TROW = TSLOTS * C;                 // elements of one (kt, plane, oc) row: 44
constexpr int TKP = C * TROW;                    // elements per (kt, plane): 176
__global__ void toeplitz_table_kernel(const float* __restrict__ w, u16* __restrict__ tab_fwd, u16* __restrict__ tab_bwd, int npl) {
    const int transposed = blockIdx.x;
    u16* tab = transposed ? tab_bwd : tab_fwd;
    if (!tab) return;
    // thread = (kt, oc, slot): four ic values
    const int idx = threadIdx.x;
    if (idx >= KT * C * TSLOTS) return;
    const int slot = idx % TSLOTS, oc = (idx / TSLOTS) % C, kt = idx / (TSLOTS * C);
    const int kf = slot - 3;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (kf >= 0 && kf < KF) {
#pragma unroll
        for (int ic = 0; ic < C; ++ic)
            v[ic] = transposed ? w[(((KT - 1 - kt) * KF + (KF - 1 - kf)) * C + oc) * C + ic] : w[((kt * KF + kf) * C + ic) * C + oc];
    }
    bf16x4 h1, h2, h3;
    split3(v, h1, h2, h3);
    u16* d = tab + (kt * npl * C + oc) * TROW + slot * C;
    *reinterpret_cast<bf16x4*>(d) = h1;
    if (npl == 3) {
        *reinterpret_cast<bf16x4*>(d + TKP) = h2;
        *reinterpret_cast<bf16x4*>(d + 2 * TKP) = h3;
    }
}

// ------------------------------------------------------------------------------------------------------------
// tile geometry
// ------------------------------------------------------------------------------------------------------------
struct Shape {
    int T, F, NG, nfb, ntt, ntiles;          // bin groups, blocks of <= 17 groups per row, time tiles, all tiles
    unsigned magic_nfb, magic_ntt;           // ceil(2^32 / d) for d > 1: mulhi(n, magic) == n / d for n < 2^20, d < 2^12
    int pad_t;
};

// LDS row layout.  A staged row holds SB bins x 4 channels = SB/2 sixteen-byte units (two bins each).  ds_read_b128 serves a
// wave in four groups of 16 lanes, each made of 8 lanes (lg = a) and 8 lanes (lg = a + 1) that together cover every
// li = 0..15 once (li in {0-3, 12-15} on one side, {4-11} on the other); a lane's unit is 2 g + lg, so the two sides of a
// group read the two units of one aligned pair.  With the pair's units TWO positions apart (the two low bits of the unit
// index swapped) and an odd number of units per row, the 16 lanes hit 16 different 16-byte bank slots iff the lanes
// li = 4..11 own the even time rows and the others the odd ones: row_of_lane().
__host__ __device__ constexpr int unit_pos(int u) { return (u & ~3) | ((u & 1) << 1) | ((u >> 1) & 1); }
__host__ __device__ constexpr int bin_off(int c) { return unit_pos(c >> 1) * 8 + (c & 1) * 4; }      // elements, staged bin c
// time row (0..15) of the MFMA column li
__device__ __forceinline__ int row_of_lane(int li) { return (li >= 4 && li < 12) ? 2 * (li - 4) : (li < 4 ? 2 * li + 1 : 2 * li - 15); }

template <int SB_, int ROWS_>
struct Stage {
    static constexpr int SB = SB_;                     // staged bins (-2 .. SB-3 relative to the block's first bin)
    static constexpr int RS = SB_ * C + 8;             // row stride in bf16 elements: an odd number of 16-byte units
    static constexpr int ROWS = ROWS_;
    static constexpr int PS = ROWS_ * RS;              // plane stride
    static constexpr int TOTAL = ROWS_ * SB_;
    static constexpr int NB = (TOTAL + THREADS - 1) / THREADS;
};

struct TilePos { long long img; int t0, g_base, ng; };

__device__ __forceinline__ TilePos tile_pos(const Shape& s, int tile) {
    TilePos p;
    const int q = s.nfb > 1 ? (int)__umulhi((unsigned)tile, s.magic_nfb) : tile, fb = tile - q * s.nfb;
    const int b = s.ntt > 1 ? (int)__umulhi((unsigned)q, s.magic_ntt) : q, tb = q - b * s.ntt;
    p.img = (long long)b * s.T * s.F;
    p.t0 = tb * 16;
    p.g_base = fb * GPB;
    p.ng = min(GPB, s.NG - p.g_base);
    return p;
}

// The work list of a persistent workgroup.  With G workgroups and n tiles: R = n / G full rounds (tile = wg + r G) and
// L = n % G left-over tiles.  A left-over round that occupies few workgroups for a whole tile time is the most expensive part
// of the launch ([64,400,65,4]: 1600 tiles on 768 workgroups -> 64 of them would run a third tile while 704 idle), so every
// left-over tile is cut into S pieces along the frequency axis for S times as many workgroups.  The FIRST tile of a workgroup
// is cut into FS pieces as well: all workgroups start at once, nothing overlaps the loads of their first piece, and the
// smaller it is the sooner the matrix pipe has work.  Pieces are multiples of `unit` bin groups wide.
struct Sched { int G, R, L, S, FS, unit; };
struct Work { int tile, g0, ng; };       // ng == 0: nothing
__device__ __forceinline__ int sched_items(const Sched& c, int wg) {
    return (c.R >= 1 ? c.FS + c.R - 1 : 0) + (wg < c.L * c.S ? 1 : 0);
}
__device__ __forceinline__ Work piece_of(int tile, int ngt, int p, int k, int unit) {
    Work w; w.tile = tile; w.g0 = 0; w.ng = ngt;
    if (k > 1) {                       // (whole tiles, the common case, stay clear of the integer divisions)
        int per = (ngt + k - 1) / k;
        per = (per + unit - 1) & ~(unit - 1);          // unit is 1 or 2
        w.g0 = p * per;
        w.ng = max(0, min(per, ngt - w.g0));
    }
    return w;
}
// item `it` of workgroup `wg` (0 <= it < sched_items); ngt_of(tile) = bin groups of that tile's block
__device__ __forceinline__ Work work_item(const Shape& s, const Sched& c, int wg, int it) {
    const int nfull = c.R >= 1 ? c.FS + c.R - 1 : 0;
    int tile, p = 0, k = 1;
    if (it < nfull) {
        if (it < c.FS) { tile = wg; p = it; k = c.FS; }
        else tile = wg + (it - c.FS + 1) * c.G;
    } else {
        tile = c.R * c.G + wg / c.S; p = wg % c.S; k = c.S;
    }
    const int fb = s.nfb > 1 ? tile - (int)__umulhi((unsigned)tile, s.magic_nfb) * s.nfb : 0;
    const int ngt = min(GPB, s.NG - fb * GPB);
    return piece_of(tile, ngt, p, k, c.unit);
}
__device__ __forceinline__ TilePos work_pos(const Shape& s, const Work& w) {
    TilePos p = tile_pos(s, w.tile);
    p.g_base += w.g0;
    p.ng = w.ng;
    return p;
}
// the next non-empty item after `it` (it is advanced), or ng == 0 at the end of the list
__device__ __forceinline__ Work next_work(const Shape& s, const Sched& c, int wg, int& it, int nitems) {
    Work w; w.tile = 0; w.g0 = 0; w.ng = 0;
    while (++it < nitems) {
        w = work_item(s, c, wg, it);
        if (w.ng > 0) break;
    }
    if (it >= nitems) w.ng = 0;
    return w;
}

// The same list, walked: next() fills the position of the next non-empty item.  The whole tiles of the full rounds -- all but the
// last item of a workgroup when the first tile is not cut (FS == 1) and a row is one block (nfb == 1: F <= 68) -- are wg, wg + G,
// wg + 2 G, ...: their (utterance, time tile) advance by (G / ntt, G % ntt) with one carry, a handful of scalar instructions.  The
// general path costs ~100 scalar instructions and a dozen branches per item, which the wave-specialised kernels' multiplying waves
// -- ONE per SIMD, nothing else feeds the matrix pipe meanwhile -- paid between every two pieces: 0.39 of the 2.6 us a piece of the
// forward kernel took (s_memtime stamps, tools/c2m_ws_stamps.py).
struct PieceWalk {
    int it, nitems, nwhole, b, tb, db, dtb, TF;
    __device__ __forceinline__ void init(const Shape& s, const Sched& c, int wg, int nitems_) {
        it = -1; nitems = nitems_;
        nwhole = (c.FS == 1 && s.nfb == 1 && c.R >= 1) ? min(c.R, nitems_) : 0;
        b = s.ntt > 1 ? (int)__umulhi((unsigned)wg, s.magic_ntt) : wg; tb = wg - b * s.ntt;
        db = s.ntt > 1 ? (int)__umulhi((unsigned)c.G, s.magic_ntt) : c.G; dtb = c.G - db * s.ntt;
        TF = s.T * s.F;
    }
    __device__ __forceinline__ bool next(const Shape& s, const Sched& c, int wg, TilePos& p) {
        if (it + 1 < nwhole) {
            if (++it > 0) {
                b += db; tb += dtb;
                if (tb >= s.ntt) { tb -= s.ntt; ++b; }
            }
            p.img = (long long)b * TF; p.t0 = tb * 16; p.g_base = 0; p.ng = min(GPB, s.NG);
            return true;
        }
        const Work w = next_work(s, c, wg, it, nitems);
        if (w.ng <= 0) return false;
        p = work_pos(s, w);
        return true;
    }
};

// What a lane needs to know about its NB staging slots, computed once per workgroup (tile-independent):
// slot idx = tid + 256 u -> staged row r = idx / SB, staged bin c = idx % SB
constexpr int ROWS_OOB = (int)0x80000000u;      // a byte offset no image reaches (pref_load_rows wants T F 16 < 2^30)
template <class ST>
struct Slots {
    int rc[ST::NB];        // r << 8 | c, or -1 for a slot beyond the tile
    int dst[ST::NB];       // LDS element offset inside a plane
    int boff[ST::NB];      // (init_rows) byte offset of the slot's pixel from (staged row 0, bin 0 of the image), or ROWS_OOB
    // Tiles that span whole rows of the image (one block per row: F <= 68): a slot's bin does not depend on the tile, so its
    // byte offset from the tile's first row is fixed and "this bin is outside the image" is a property of the slot -- see pref_load_rows
    __device__ __forceinline__ void init_rows(int F) {
#pragma unroll
        for (int u = 0; u < ST::NB; ++u) {
            const int r = rc[u] >> 8, c = rc[u] & 255;
            boff[u] = (rc[u] >= 0 && c >= 2 && c - 2 < F) ? (r * F + c - 2) * (C * 4) : ROWS_OOB;
        }
    }
    __device__ __forceinline__ void init(int t = threadIdx.x) {
#pragma unroll
        for (int u = 0; u < ST::NB; ++u) {
            const int idx = t + u * THREADS;
            const int r = idx / ST::SB, c = idx - r * ST::SB;
            rc[u] = idx < ST::TOTAL ? (r << 8 | c) : -1;
            dst[u] = r * ST::RS + bin_off(c);
        }
    }
};

template <int NB, bool MASK>
struct Pref {
    f32x4 v[NB];
    f32x4 m[MASK ? NB : 1];
};

// the loads of one tile: rows t_org .., bins f_org ..; zero outside the image
__device__ __forceinline__ f32x4 load_px(const void* base, int off /*elements*/, bool bf16) {
    if (bf16) return __builtin_convertvector(*reinterpret_cast<const bf16x4*>(reinterpret_cast<const u16*>(base) + off), f32x4);
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + off);
}
__device__ __forceinline__ void store_px(void* base, long long off /*elements*/, f32x4 v, bool bf16) {
    if (bf16) *reinterpret_cast<bf16x4*>(reinterpret_cast<u16*>(base) + off) = __builtin_convertvector(v, bf16x4);
    else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + off) = v;
}
__device__ __forceinline__ const void* ptr_at(const void* p, long long off /*elements*/, bool bf16) {
    return bf16 ? (const void*)(reinterpret_cast<const u16*>(p) + off) : (const void*)(reinterpret_cast<const float*>(p) + off);
}

template <class ST, bool MASK>
__device__ __forceinline__ void pref_load(Pref<ST::NB, MASK>& pf, const Slots<ST>& sl, const void* __restrict__ src,
                                          const void* __restrict__ msk, bool bf16, long long img, int t_org, int f_org, int T, int F,
                                          int f_end /*bins >= f_end are staged as zeros (f_end <= F)*/,
                                          int c_end = 1 << 20 /*staged bins >= c_end are not needed: zeros, no load*/) {
    const void* base = ptr_at(src, (img + (long long)t_org * F + f_org) * C, bf16);         // wave-uniform
    const void* mbase = MASK ? ptr_at(msk, (img + (long long)t_org * F + f_org) * C, bf16) : nullptr;
#pragma unroll
    for (int u = 0; u < ST::NB; ++u) {
        const int r = sl.rc[u] >> 8, c = sl.rc[u] & 255;
        const bool ok = sl.rc[u] >= 0 && c < c_end && (unsigned)(t_org + r) < (unsigned)T && (unsigned)(f_org + c) < (unsigned)f_end;
        pf.v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (MASK) pf.m[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const int off = (r * F + c) * C;
            pf.v[u] = load_px(base, off, bf16);
            if (MASK) pf.m[u] = load_px(mbase, off, bf16);
        }
    }
}

// The same loads for a tile that spans whole rows of an fp32 image (f_org == -2, every bin of the image staged), as RAW BUFFER loads
// over the utterance [T][F][4]: a 16-byte load whose byte offset lies outside [0, T F 16) returns zeros (checked on gfx950 for negative,
// straddling and far offsets: tools/hip/buffer_oob_test.hip), so rows above and below the utterance need no test, and the bins left and
// right of it carry the out-of-range offset in their slot.  One add and one load per slot -- the general form spends four compares,
// a branch and a 64-bit address on each, and the staging waves, which bound the second-order launch and the forward pass
// (tools/c2m_fused_stamps.py), spent a fifth of their loop ISSUING loads.
typedef unsigned u32x4r __attribute__((ext_vector_type(4)));
template <class ST, bool MASK>
__device__ __forceinline__ void pref_load_rows(Pref<ST::NB, MASK>& pf, const Slots<ST>& sl, const void* __restrict__ src,
                                               const void* __restrict__ msk, long long img, int t_org, int T, int F) {
    const int nbytes = T * F * (C * 4), tb = t_org * F * (C * 4);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(reinterpret_cast<const float*>(src)) + img * C, 0, nbytes, 0x00020000);
#pragma unroll
    for (int u = 0; u < ST::NB; ++u)
        pf.v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, sl.boff[u] + tb, 0, 0));
    if (MASK) {
        __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(reinterpret_cast<const float*>(msk)) + img * C, 0, nbytes, 0x00020000);
#pragma unroll
        for (int u = 0; u < ST::NB; ++u)
            pf.m[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, sl.boff[u] + tb, 0, 0));
    }
}
// One slot of the same loads, as an object: pref_commit calls it for slot u right after it has consumed that slot's registers, with
// the tile AFTER the one being committed.  The loads of a tile are then in flight for a whole piece time; issued all together behind
// the commit they had only the staging wave's wait for its buffer (0.2-0.3 us) before the next commit needed them -- the "commit" of
// the stamps was, for half of its time, a wait for HBM.
struct NoReload {
    template <class P> __device__ __forceinline__ void operator()(P&, int) const {}
};
template <class ST, bool MASK>
struct RowsReload {
    __amdgpu_buffer_rsrc_t rs, rm;
    int tb;
    const Slots<ST>* sl;
    __device__ __forceinline__ RowsReload(const Slots<ST>& sl_, const void* src, const void* msk, long long img, int t_org, int T, int F) {
        const int nbytes = T * F * (C * 4);
        tb = t_org * F * (C * 4);
        sl = &sl_;
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(reinterpret_cast<const float*>(src)) + img * C, 0, nbytes, 0x00020000);
        rm = rs;
        if (MASK) rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(reinterpret_cast<const float*>(msk)) + img * C, 0, nbytes, 0x00020000);
    }
    __device__ __forceinline__ void operator()(Pref<ST::NB, MASK>& pf, int u) const {
        pf.v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, sl->boff[u] + tb, 0, 0));
        if (MASK) pf.m[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, sl->boff[u] + tb, 0, 0));
    }
};
// is the piece a whole-row tile of an image small enough for pref_load_rows?
__device__ __forceinline__ bool rows_tile(const Shape& s, const TilePos& p) {
    return s.nfb == 1 && p.g_base == 0 && p.ng == s.NG && (long long)s.T * s.F * (C * 4) < (1LL << 30);
}

// registers -> transform -> three bf16 planes in LDS.  Slots outside the image were loaded as zeros and stay zero under
// every transform but the BatchNorm-affine one, which gets its own select.  `sum` (optional) accumulates the raw values
// of staged rows [sum_r0, sum_r0 + 16), staged bins [2, sum_c1).
template <class ST, int MODE, int NPL, class RL = NoReload>
__device__ __forceinline__ void pref_commit(Pref<ST::NB, MODE == PTTS_IN_MASKMUL>& pf, const Slots<ST>& sl,
                                            u16* __restrict__ planes, int t_org, int f_org, int T, int F,
                                            const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                            float alpha, int sum_r0, int sum_c1, f32x4* sum, const RL& reload = RL()) {
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    const bool affine = MODE == PTTS_IN_LRELU && in_scale != nullptr;
    if (affine) {
        sc = *reinterpret_cast<const f32x4*>(in_scale);
        sh = *reinterpret_cast<const f32x4*>(in_shift);
    }
#pragma unroll
    for (int u = 0; u < ST::NB; ++u) {
        // Every lane owns its slots u < TOTAL / THREADS; only the last one can lie beyond the tile.  The test is folded away for the others (the
        // loop is unrolled): with a branch per slot the compiler kept every slot's chain of dependent conversions in a block of its own --
        // wait for ITS load, convert, three writes -- and a staging wave, alone of its kind on its SIMD, sat out every latency (the commit
        // of a piece took 2.5 x its instruction count, tools/c2m_fused_stamps.py: the staging waves, not the multiplying ones, bound the
        // second-order launch and the forward pass).  In one block the chains of different slots interleave.
        if ((u + 1) * THREADS > ST::TOTAL && sl.rc[u] < 0) { reload(pf, u); continue; }
        f32x4 a = pf.v[u];
        f32x4 mk = {0.f, 0.f, 0.f, 0.f};
        if (MODE == PTTS_IN_MASKMUL) mk = pf.m[u];
        reload(pf, u);                 // (the slot's registers are free: the next tile's load goes out now)
        if (sum) {
            // the block's own pixels only: rows [sum_r0, sum_r0 + 16), staged bins [2, sum_c1) (the rest is halo)
            const int r = sl.rc[u] >> 8, c = sl.rc[u] & 255;
            if (r >= sum_r0 && r < sum_r0 + 16 && c >= 2 && c < sum_c1) *sum += a;
        }
        if (MODE == PTTS_IN_LRELU) {
            if (affine) {
                const int r = sl.rc[u] >> 8, c = sl.rc[u] & 255;
                const bool ok = (unsigned)(t_org + r) < (unsigned)T && (unsigned)(f_org + c) < (unsigned)F;
                a = a * sc + sh;
                if (!ok) a = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = max_fast(a[e], alpha * a[e]);      // LeakyReLU for 0 <= alpha <= 1
        } else if (MODE == PTTS_IN_MASKMUL) {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = a[e] * (mk[e] > 0.f ? 1.f : alpha);
        }
        u16* d = planes + sl.dst[u];
        if (NPL == 3) {
            bf16x4 h1, h2, h3;
#if C2M_PROBE_COMMIT == 2
            // (timing probe, garbage: no conversions -- the raw bits go to the planes)
            h1 = __builtin_bit_cast(bf16x4, (u32x2){__builtin_bit_cast(unsigned, a[0]), __builtin_bit_cast(unsigned, a[1])});
            h2 = __builtin_bit_cast(bf16x4, (u32x2){__builtin_bit_cast(unsigned, a[2]), __builtin_bit_cast(unsigned, a[3])});
            h3 = h1;
#else
            split3(a, h1, h2, h3);
#endif
#if C2M_PROBE_COMMIT == 1
            // (timing probe, garbage: the conversions without the three LDS writes)
            asm volatile("" :: "v"(h1), "v"(h2), "v"(h3), "v"(d));
#else
            *reinterpret_cast<bf16x4*>(d) = h1;
            *reinterpret_cast<bf16x4*>(d + ST::PS) = h2;
            *reinterpret_cast<bf16x4*>(d + 2 * ST::PS) = h3;
#endif
        } else {
            *reinterpret_cast<bf16x4*>(d) = __builtin_convertvector(a, bf16x4);       // the one rounding of bf16 arithmetic
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// forward / masked forward / backward data.  Persistent workgroups (256 threads) walk the tiles blockIdx.x, + gridDim.x,
// ...: the Toeplitz table is fetched once per workgroup and the global loads of tile i+1 are in flight (registers) while
// tile i is multiplied.  LDS: planes [NBUF][NP][ROWS][RS] | operand table [KT][NP][oc][11 taps][ic]
// out = bias + conv(transform(x)) ; OUTMASK: out *= (out_mask > 0 ? 1 : alpha)
// ------------------------------------------------------------------------------------------------------------
// one pass of a wave over N consecutive bin groups g0 .. g0+N-1 of the staged block: straight-line code (a branch around
// an MFMA group makes the compiler carry the accumulators through register copies), one LDS base register per parity of
// the group, everything else immediates
template <class ST, int DIL, int N, bool OUTMASK, int NPL, bool WREG, bool STATS = false>
__device__ __forceinline__ void fwd_pass(const u16* __restrict__ planes, const u16* __restrict__ wl, const bf16x8 (&wf)[KT][NPL], int g0, int lane,
                                         f32x4 bv, const void* __restrict__ mrow, void* __restrict__ yrow, bool out_bf16,
                                         int fbase, int F, bool rowok, float alpha, bool store, f32x4* stat = nullptr, const f32x4* aff = nullptr) {
    const int li = lane & 15, lg = lane >> 4;
    f32x4 acc[N], mv[OUTMASK ? N : 1];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = bv;
    // this lane's output pixels: row row_of_lane(li) (yrow / mrow point at its bin 0), bins fbase + 4 (g0 + j) + lg
    const int f0 = fbase + 4 * g0 + lg;
    if (OUTMASK) {
        // the mask source of the outputs: requested now, needed after the MFMAs
#pragma unroll
        for (int j = 0; j < N; ++j) {
            mv[j] = f32x4{1.f, 1.f, 1.f, 1.f};
            // (inside the branch on purpose: loaded unconditionally from a clamped position the backward-data launch measured 54.6 -> 65.7 us
            // at [192,400,65,4] -- the compiler then counts the loads exactly and lets the stores overtake, tools/c2m_ws_phases.py)
            if (rowok && f0 + 4 * j < F) mv[j] = load_px(mrow, (f0 + 4 * j) * C, out_bf16);
        }
    }
    // unit of the lane in group g: 2 g + lg; groups two apart are four units (32 elements) apart
    const u16* b0 = planes + row_of_lane(li) * ST::RS + unit_pos(2 * g0 + lg) * 8;
    const u16* b1 = planes + row_of_lane(li) * ST::RS + unit_pos(2 * g0 + 2 + lg) * 8;
    const u16* wa = wl + (li & 3) * TROW + (2 * lg - (li >> 2) + 3) * C;      // row oc, slot of tap 2 lg - so
    // fragments of kernel row kt: the table's (8-byte aligned: two 8-byte reads) and N x NPL of the activations.  The reads
    // of row kt + 1 are issued before the MFMAs of row kt (two register sets): a wave does not sit out an LDS round trip
    // per kernel row.
    bf16x8 a[2][NPL], bq[2][N][NPL];
    auto read_row = [&](int kt, bf16x8 (&ar)[NPL], bf16x8 (&br)[N][NPL]) {
        if (!WREG) {
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const u16* wp = wa + (kt * NPL + q) * TKP;
                ar[q] = cat(*reinterpret_cast<const bf16x4*>(wp), *reinterpret_cast<const bf16x4*>(wp + 4));
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                if (C2M_PROBE_BREUSE && kt > 0) {
                    // (timing probe only, wrong results: what the pass costs if the fragments of rows kt > 0 came out of registers -- one
                    // vector move per register -- instead of out of the LDS)
                    u32x4v v = __builtin_bit_cast(u32x4v, bq[(kt + 1) & 1][j][p]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_update_dpp(v[e], v[e], 0x101, 0xf, 0xf, false);
                    br[j][p] = __builtin_bit_cast(bf16x8, v);
                    continue;
                }
                br[j][p] = *reinterpret_cast<const bf16x8*>(((j & 1) ? b1 : b0) + p * ST::PS + kt * DIL * ST::RS + (j >> 1) * 32);
            }
    };
    read_row(0, a[0], bq[0]);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) read_row(kt + 1, a[(kt + 1) & 1], bq[(kt + 1) & 1]);
        // product-major: consecutive MFMAs go to different accumulators
#define C2M_MM(PA, PW)                                                                                  \
        _Pragma("unroll") for (int j = 0; j < N; ++j)                                                   \
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WREG ? wf[kt][PW < NPL ? PW : 0] : a[kt & 1][PW < NPL ? PW : 0],                \
                                                             bq[kt & 1][j][PA < NPL ? PA : 0], acc[j], 0, 0, 0);
        C2M_PRODUCTS_NPL(NPL, C2M_MM);
#undef C2M_MM
    }
    if (rowok && store) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (f0 + 4 * j < F) {
                f32x4 o = acc[j];
                if (OUTMASK && STATS) {
                    // backward data of a layer whose input was lrelu(sc x + sh) (a BatchNormalization in front: the generator's stack): the
                    // mask is that input's, the gradient w.r.t. x carries the factor sc, and the two sums are the gradients of sc and sh
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gd = o[e] * ((mv[j][e] * aff[0][e] + aff[1][e]) > 0.f ? 1.f : alpha);
                        stat[0][e] += gd * mv[j][e]; stat[1][e] += gd;
                        o[e] = gd * aff[0][e];
                    }
                } else if (OUTMASK) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = o[e] * (mv[j][e] > 0.f ? 1.f : alpha);
                }
                if (STATS && !OUTMASK) { stat[0] += o; stat[1] += o * o; }       // per-channel sum and sum of squares of what is stored (BatchNorm's statistics)
                store_px(yrow, (f0 + 4 * j) * C, o, out_bf16);
            }
        }
    }
}

// the MFMA phase of one staged piece: the wave's share of its bin groups, a contiguous run, in passes of at most NMAX groups
// of nearly equal size (register budget: a pass keeps N x 3 activation fragments).  The wave that takes the odd group
// changes from piece to piece: the waves of a workgroup sit on different SIMDs.
template <class ST, int DIL, bool OUTMASK, bool MASK, int NPL, bool WREG, int NW = 4, int NMAXO = 0, bool STATS = false>
__device__ __forceinline__ void fwd_piece(const u16* __restrict__ planes, const u16* __restrict__ wl, const bf16x8 (&wf)[KT][NPL], const TilePos& cur, const Shape& s,
                                          int wave, int lane, int it, f32x4 bv, const void* __restrict__ out_mask, void* __restrict__ y,
                                          bool out_bf16, float alpha, bool store, bool nomfma, f32x4* stat = nullptr, const f32x4* aff = nullptr) {
    static_assert(NW == 4 || NW == 8, "waves that share a piece");
    const int per = cur.ng / NW, rem = cur.ng & (NW - 1), wr = (wave + it) & (NW - 1);
    int gl = wr * per + min(wr, rem);
    int n = per + (wr < rem ? 1 : 0);
    if (nomfma) n = 0;
    constexpr int NMAX = NMAXO > 0 ? NMAXO : ((OUTMASK || MASK) ? 4 : 5);       // (five with the mask values as well fits the register-table form but measured 1 us slower)
    int npass = (n + NMAX - 1) / NMAX;
    const int t = cur.t0 + row_of_lane(lane & 15);
    const bool rowok = t < s.T;
    const long long rowoff = (cur.img + (long long)t * s.F) * C;
    void* yrow = const_cast<void*>(ptr_at(y, rowoff, out_bf16));
    const void* mrow = OUTMASK ? ptr_at(out_mask, rowoff, out_bf16) : nullptr;
    const int fbase = 4 * cur.g_base;
    while (n > 0) {
        const int m = npass == 1 ? n : (npass == 2 ? (n + 1) >> 1 : (n + npass - 1) / npass);      // (the division is ~35 scalar instructions)
#define C2M_PASS(NN) fwd_pass<ST, DIL, NN, OUTMASK, NPL, WREG, STATS>(planes, wl, wf, gl, lane, bv, mrow, yrow, out_bf16, fbase, s.F, rowok, alpha, store, stat, aff)
        switch (m) {
            case 1: C2M_PASS(1); break;
            case 2: if (NMAX >= 2) C2M_PASS(NMAX >= 2 ? 2 : 1); break;
            case 3: if (NMAX >= 3) C2M_PASS(NMAX >= 3 ? 3 : 1); break;
            case 4: if (NMAX >= 4) C2M_PASS(NMAX >= 4 ? 4 : 1); break;
            default: if (NMAX >= 5) C2M_PASS(NMAX >= 5 ? 5 : 1); break;
        }
#undef C2M_PASS
        gl += m; n -= m; --npass;
    }
}

// NBUF == 1: stage -> barrier -> multiply -> barrier per piece (the long stages of the dilated layers fill the LDS).
// NBUF == 2 (dilation 1): the planes are double-buffered -- the piece i+1 is activated, split and written to the other
// buffer by the same waves that multiply piece i, ONE barrier per piece, and the two workgroups of a CU (86 -> 76 KB of LDS
// each with the compact table) drift against each other, so that one's vector work runs under the other's MFMAs.
template <int DIL, int MODE, bool OUTMASK, int NPL, int NBUF, bool WREG_ = false>
__global__ __launch_bounds__(THREADS, NBUF == 2 ? 2 : (DIL == 1 ? 3 : (DIL == 2 ? 2 : 1))) void fwd_kernel(
    const void* __restrict__ x, const u16* __restrict__ tab, const float* __restrict__ bias,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, const void* __restrict__ mask_src,
    const void* __restrict__ out_mask, void* __restrict__ y, int dt, Shape s, Sched sc, float alpha, int dbg, unsigned long long* dbg_buf) {
    typedef Stage<4 * GPB + 4, 16 + (KT - 1) * DIL> ST;
    extern __shared__ __attribute__((aligned(16))) u16 lds[];
    stamp(dbg_buf, dbg, 0);
    u16* planes = lds;
    u16* wl = lds + NBUF * NPL * ST::PS;
    const bool in_bf16 = NPL == 1 && (dt & DT_IN) != 0, out_bf16 = NPL == 1 && (dt & DT_OUT) != 0;   // three planes: fp32 maps
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr bool MASK = MODE == PTTS_IN_MASKMUL;

    Slots<ST> sl;
    sl.init();
    const int wg = blockIdx.x, nitems = sched_items(sc, wg);
    int item = -1;
    Work work = next_work(s, sc, wg, item, nitems);
    Pref<ST::NB, MASK> pf;
    TilePos pos = work_pos(s, work);
    const bool stage = !(dbg & DBG_NOSTAGE), nomfma = (dbg & DBG_NOMFMA) != 0, nobar = (dbg & DBG_NOBAR) != 0;
    if (work.ng > 0 && stage)
        pref_load<ST, MASK>(pf, sl, x, mask_src, in_bf16, pos.img, pos.t0 - s.pad_t, 4 * pos.g_base - 2, s.T, s.F, s.F, 4 * pos.ng + 4);
    stamp(dbg_buf, dbg, 5);
    // the table.  Dilation 1, double-buffered (WREG): a lane's 5 x NPL operand fragments -- they do not depend on the tile -- are read from the table in
    // global memory once and stay in registers for the whole launch; out of the LDS, as before, they were a third of the
    // matrix phase's LDS reads (6 of 18 sixteen-byte reads per kernel row and pass), and the LDS pipe, not the matrix pipe, bounds
    // that phase.  The dilated kernels have no registers to spare: KT * NPL * 352 bytes into the LDS, once per workgroup.
    constexpr bool WREG = WREG_ && DIL == 1 && NBUF == 2;           // (the single-buffered form runs three workgroups per CU: 168 registers)
    bf16x8 wf[KT][NPL];
    if (WREG) {
        const int li = lane & 15, lg = lane >> 4;
        const u16* wa = tab + (li & 3) * TROW + (2 * lg - (li >> 2) + 3) * C;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const u16* wp = wa + (kt * NPL + q) * TKP;
                wf[kt][q] = cat(*reinterpret_cast<const bf16x4*>(wp), *reinterpret_cast<const bf16x4*>(wp + 4));
            }
    } else {
        for (int i = tid; i < KT * NPL * TKP / 8; i += THREADS)
            *reinterpret_cast<bf16x8*>(wl + (size_t)i * 8) = *reinterpret_cast<const bf16x8*>(tab + (size_t)i * 8);
    }
    stamp(dbg_buf, dbg, 6);
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias);
    const bool store = !(dbg & DBG_NOSTORE);
    int it = 0;
    if (NBUF == 1) {
        while (work.ng > 0) {
            if (stage)
                pref_commit<ST, MODE, NPL>(pf, sl, planes, pos.t0 - s.pad_t, 4 * pos.g_base - 2, s.T, s.F, in_scale, in_shift, alpha, 0, 0, nullptr);
            if (it == 0) stamp(dbg_buf, dbg, 1);
            __syncthreads();
            if (it == 0) stamp(dbg_buf, dbg, 2);
            const TilePos cur = pos;
            work = next_work(s, sc, wg, item, nitems);
            if (work.ng > 0) {
                pos = work_pos(s, work);
                if (stage)
                    pref_load<ST, MASK>(pf, sl, x, mask_src, in_bf16, pos.img, pos.t0 - s.pad_t, 4 * pos.g_base - 2, s.T, s.F, s.F, 4 * pos.ng + 4);
            }
            fwd_piece<ST, DIL, OUTMASK, MASK, NPL, WREG>(planes, wl, wf, cur, s, wave, lane, it, bv, out_mask, y, out_bf16, alpha, store, nomfma);
            if (it == 0) stamp(dbg_buf, dbg, 3);
            __syncthreads();       // the planes are free again
            ++it;
        }
    } else {
        // piece 0 into buffer 0, the loads of piece 1 in flight
        if (work.ng > 0 && stage)
            pref_commit<ST, MODE, NPL>(pf, sl, planes, pos.t0 - s.pad_t, 4 * pos.g_base - 2, s.T, s.F, in_scale, in_shift, alpha, 0, 0, nullptr);
        TilePos cur = pos;
        bool have = work.ng > 0;
        work = next_work(s, sc, wg, item, nitems);
        if (work.ng > 0) {
            pos = work_pos(s, work);
            if (stage)
                pref_load<ST, MASK>(pf, sl, x, mask_src, in_bf16, pos.img, pos.t0 - s.pad_t, 4 * pos.g_base - 2, s.T, s.F, s.F, 4 * pos.ng + 4);
        }
        stamp(dbg_buf, dbg, 1);
        __syncthreads();
        stamp(dbg_buf, dbg, 2);
        const bool late = ((blockIdx.x >> 8) & 1) != 0;
        while (have) {
            const u16* pcur = planes + (it & 1) * NPL * ST::PS;
            u16* pnxt = planes + ((it + 1) & 1) * NPL * ST::PS;
            const bool more = work.ng > 0;
            const TilePos nxt = pos;
            // the next piece: registers -> the other buffer (every wave left it at the last barrier), then the loads of the piece
            // after.  The two workgroups of a CU (blockIdx i and i + 256 where the dispatcher fills the CUs round-robin) start
            // together and would stay in lockstep -- vector phases together, then matrix phases together: one of them stages
            // BEFORE its MFMAs, the other AFTER.
            auto stage_next = [&]() {
                if (more && stage)
                    pref_commit<ST, MODE, NPL>(pf, sl, pnxt, nxt.t0 - s.pad_t, 4 * nxt.g_base - 2, s.T, s.F, in_scale, in_shift, alpha, 0, 0, nullptr);
                if (more) {
                    work = next_work(s, sc, wg, item, nitems);
                    if (work.ng > 0) {
                        pos = work_pos(s, work);
                        if (stage)
                            pref_load<ST, MASK>(pf, sl, x, mask_src, in_bf16, pos.img, pos.t0 - s.pad_t, 4 * pos.g_base - 2, s.T, s.F, s.F, 4 * pos.ng + 4);
                    }
                }
            };
            if (!late) stage_next();
            fwd_piece<ST, DIL, OUTMASK, MASK, NPL, WREG>(pcur, wl, wf, cur, s, wave, lane, it, bv, out_mask, y, out_bf16, alpha, store, nomfma);
            if (late) stage_next();
            if (it == 0) stamp(dbg_buf, dbg, 3);
            if (!nobar) __syncthreads();
            cur = nxt; have = more;
            ++it;
        }
    }
    stamp(dbg_buf, dbg, 4);
}

// ------------------------------------------------------------------------------------------------------------
// Dilation 1, WAVE-SPECIALISED (the default since the end of round 3): one workgroup of EIGHT waves per CU -- two per SIMD, as with two
// four-wave workgroups -- whose waves 4..7 only STAGE (global loads of piece i + 2 -> registers; activation, three-way split and
// ds_write of piece i + 1 into the other plane buffer) and whose waves 0..3 only MULTIPLY piece i (LDS reads, MFMAs, stores), one
// barrier per piece.  Every SIMD then always has one wave of each kind: the vector work of the split runs in the issue gaps of the
// other wave's MFMAs instead of before or after them (in the four-wave form a workgroup's phases alternate and the two workgroups of a
// CU overlap only when they happen to be in opposite phases), and a workgroup walks twice as many tiles behind one prologue.
// Same tiles, same pass structure, same arithmetic as fwd_kernel<1, ...>: results are bit-identical.
// (Round 4, measured: the premise holds only in part.  A SIMD of gfx950 issues another wave's vector instructions at about a tenth of
// their rate while a wave's MFMAs run -- tools/hip/mfma_valu_overlap_test.hip: both together take 0.9 of the SUM of their times -- so the
// split does not hide under the MFMAs; what the specialisation does buy is a multiplying wave whose instruction stream holds nothing but
// LDS reads, MFMAs and stores, with the loads' latency on other waves.  s_memtime stamps of both roles: tools/c2m_ws_stamps.py.)
// ------------------------------------------------------------------------------------------------------------
// STATS (round 4): the launch also leaves, per workgroup, the per-channel sum and sum of squares of the outputs it stored --
// stats[workgroup][8] doubles, [0..3] sums, [4..7] sums of squares -- for the BatchNormalization layer that follows the convolution in the
// generator's stack (reference networktts.py:122-126): its statistics no longer cost a pass of their own over the map (ptts_bn_batch_stats:
// 16 us a layer, 8 layers, in every forward of the generator -- the critic step's fake sample included), only ptts_bn_finalize_partials.
// A lane adds what it stores (fp32: 80 values per launch and channel), the workgroup's 256 lanes are added in double, in a fixed order.
template <int MODE, bool OUTMASK, int NPL, int NMW, bool STATS = false>
__global__ __launch_bounds__((NMW + 4) * 64, 1) void fwd_ws_kernel(
    const void* __restrict__ x, const u16* __restrict__ tab, const float* __restrict__ bias,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, const void* __restrict__ mask_src,
    const void* __restrict__ out_mask, void* __restrict__ y, int dt, Shape s, Sched sc, float alpha, int dbg, unsigned long long* dbg_buf,
    unsigned* status, double* __restrict__ stats = nullptr) {
    constexpr int DIL = 1;
    typedef Stage<4 * GPB + 4, 16 + (KT - 1) * DIL> ST;
    extern __shared__ __attribute__((aligned(16))) u16 lds[];
    stamp(dbg_buf, dbg, 0);
    u16* planes = lds;
    const bool in_bf16 = NPL == 1 && (dt & DT_IN) != 0, out_bf16 = NPL == 1 && (dt & DT_OUT) != 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool stager = wave8 >= NMW;
    const int wave = wave8 & (NMW - 1);
    constexpr bool MASK = MODE == PTTS_IN_MASKMUL;
    const bool stage = !(dbg & DBG_NOSTAGE), nomfma = (dbg & DBG_NOMFMA) != 0, store = !(dbg & DBG_NOSTORE);

    Slots<ST> sl;
    Pref<ST::NB, MASK> pf;
    bf16x8 wf[KT][NPL];
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    const int wg = blockIdx.x, nitems = sched_items(sc, wg);
    PieceWalk walk;
    walk.init(s, sc, wg, nitems);
    TilePos cur = {0, 0, 0, 0};
    bool havecur = walk.next(s, sc, wg, cur);
    TilePos nxt = cur;
    bool havenxt = havecur && walk.next(s, sc, wg, nxt);
    auto stage_load = [&](const TilePos& p) {
        if (!in_bf16 && rows_tile(s, p)) pref_load_rows<ST, MASK>(pf, sl, x, mask_src, p.img, p.t0 - s.pad_t, s.T, s.F);
        else pref_load<ST, MASK>(pf, sl, x, mask_src, in_bf16, p.img, p.t0 - s.pad_t, 4 * p.g_base - 2, s.T, s.F, s.F, 4 * p.ng + 4);
    };
    if (stager) {
        sl.init(tid - NMW * 64);
        sl.init_rows(s.F);
        if (havecur && stage) {
            stage_load(cur);
            pref_commit<ST, MODE, NPL>(pf, sl, planes, cur.t0 - s.pad_t, 4 * cur.g_base - 2, s.T, s.F, in_scale, in_shift, alpha, 0, 0, nullptr);
        }
        if (havenxt && stage) stage_load(nxt);
    } else {
        const int li = lane & 15, lg = lane >> 4;
        const u16* wa = tab + (li & 3) * TROW + (2 * lg - (li >> 2) + 3) * C;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const u16* wp = wa + (kt * NPL + q) * TKP;
                wf[kt][q] = cat(*reinterpret_cast<const bf16x4*>(wp), *reinterpret_cast<const bf16x4*>(wp + 4));
                if (C2M_PIN_WF == 1) asm volatile("" : "+v"(wf[kt][q]));      // (C2M_PIN_WF: see its definition)
            }
        if (bias) bv = *reinterpret_cast<const f32x4*>(bias);
        if (C2M_PIN_WF == 2) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int q = 0; q < NPL; ++q) asm volatile("" : "+v"(wf[kt][q]));
            asm volatile("" : "+v"(bv));
        }
    }
    // Hand-off of the plane buffers by COUNTERS in the LDS instead of a workgroup barrier per piece (C2M_WS_FLAGS): piece p lives in
    // buffer p & 1, use u = p >> 1.  ready[b] counts the staging waves' commits into buffer b, done[b] the multiplying waves that have
    // finished reading it: a multiplying wave starts piece p when ready[p & 1] == 4 (u + 1), a staging wave commits piece p when
    // done[p & 1] == NMW u.  The multiplying waves then drift against each other by up to a piece -- 17 bin groups over four waves is
    // 5 + 4 + 4 + 4 with the fifth rotating, and a barrier per piece makes every piece last as long as five.  Every poll is bounded:
    // all eight waves of the workgroup are co-resident and every count is produced by a wave that waits for nothing but an earlier
    // count, so the bound (~1 ms) is never reached by the protocol itself -- only by a stall from outside (preemption, a debugger,
    // counter serialisation).  A wave whose poll does run out goes on (the GPU does not hang) but the launch's results are wrong, so it
    // stores STATUS_C2M_HANDOFF into the device status word: the C ABI returns PTTS_EDEVICE from the next call on (common.h).
    int* const flags = reinterpret_cast<int*>(lds + 2 * NPL * ST::PS);       // ready[2] | done[2]  (the operand table's unused LDS slot)
    if (C2M_WS_FLAGS && tid == 0) { flags[0] = 4; flags[1] = 0; flags[2] = 0; flags[3] = 0; }
    const bool force_timeout = (dbg & DBG_FORCE_TIMEOUT) != 0;
    auto wait_for = [&](int idx, int target) {
        const int bound = force_timeout ? 8 : (1 << 14);
        if (force_timeout && !stager) target += 1 << 20;
        int r = 0;
        for (; r < bound; ++r) {
            if (__hip_atomic_load(flags + idx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (r == bound && lane == 0) raise_status(status, STATUS_SLOT_C2M, STATUS_C2M_HANDOFF);
    };
    // The counters order LDS traffic only: a staging wave's commit (ds_write) before `ready`, a multiplying wave's fragment reads before
    // `done`.  The LDS executes a wave's operations in order and lgkmcnt(0) says they have completed, so the count goes up with a
    // RELAXED add behind an explicit s_waitcnt lgkmcnt(0).  (A release add made the multiplying waves drain their GLOBAL stores as
    // well -- vmcnt(0), a write acknowledged by the L2, about a microsecond under load -- once per piece: 13.6 of the 63 us of the fused
    // backward kernel at 2B, tools/c2m_fused_probe2.py "no stores".)
    auto signal = [&](int idx) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(flags + idx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    stamp(dbg_buf, dbg, 1);
    __syncthreads();
    stamp(dbg_buf, dbg, 2);
    f32x4 stat[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    int it = 0;
#if C2M_PROBE_STAMPS
    // (probe build, tools/c2m_ws_stamps.py) the timeline of piece 3 of every workgroup: wave 0 (multiplying) slots 0..4, wave NMW (staging) 8..12
#define C2M_TS(SLOT) do { if (dbg_buf != nullptr && it == 3 && lane == 0 && (wave8 == 0 || wave8 == NMW)) dbg_buf[(size_t)blockIdx.x * 16 + (wave8 ? 8 : 0) + (SLOT)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define C2M_TS(SLOT) do { } while (0)
#endif
    while (havecur) {
        C2M_TS(0);
        // piece it + 2 (every wave walks the same list)
        TilePos p2 = nxt;
        const bool have2 = havenxt && walk.next(s, sc, wg, p2);
        if (stager) {
            u16* pnxt = planes + ((it + 1) & 1) * NPL * ST::PS;       // every multiplying wave left it at the last barrier / its done count
            bool reloaded = false;
            if (havenxt) {
                if (C2M_WS_FLAGS) wait_for(2 + ((it + 1) & 1), NMW * ((it + 1) >> 1));
                C2M_TS(1);
                if (stage) {
                    if (have2 && !in_bf16 && rows_tile(s, p2)) {
                        // the loads of piece it + 2 slot by slot, each behind the commit of the same slot of piece it + 1
                        const RowsReload<ST, MASK> rl(sl, x, mask_src, p2.img, p2.t0 - s.pad_t, s.T, s.F);
                        pref_commit<ST, MODE, NPL>(pf, sl, pnxt, nxt.t0 - s.pad_t, 4 * nxt.g_base - 2, s.T, s.F, in_scale, in_shift, alpha, 0, 0, nullptr, rl);
                        reloaded = true;
                    } else {
                        pref_commit<ST, MODE, NPL>(pf, sl, pnxt, nxt.t0 - s.pad_t, 4 * nxt.g_base - 2, s.T, s.F, in_scale, in_shift, alpha, 0, 0, nullptr);
                    }
                }
                C2M_TS(2);
                if (C2M_WS_FLAGS) signal((it + 1) & 1);
                C2M_TS(3);
            }
            if (have2 && stage && !reloaded) stage_load(p2);
            C2M_TS(4);
        } else {
            const u16* pcur = planes + (it & 1) * NPL * ST::PS;
            if (C2M_WS_FLAGS) wait_for(it & 1, 4 * ((it >> 1) + 1));
            C2M_TS(1);
            fwd_piece<ST, DIL, OUTMASK, MASK, NPL, true, NMW, (NMW == 8 ? 2 : C2M_WS_NMAX), STATS>(pcur, nullptr, wf, cur, s, wave, lane, it, bv, out_mask, y, out_bf16, alpha, store, nomfma, stat);
            C2M_TS(2);
            if (C2M_WS_FLAGS) signal(2 + (it & 1));
            C2M_TS(3);
        }
        if (it == 0) stamp(dbg_buf, dbg, 3);
        if (!C2M_WS_FLAGS) __syncthreads();
        cur = nxt; havecur = havenxt; nxt = p2; havenxt = have2;
        ++it;
    }
    if (STATS) {
        // the workgroup's row: every multiplying lane's eight fp32 sums into the LDS (the plane buffers are free: every piece is done when
        // every wave is here), then eight lanes add the 64 NMW values of their column in double, in index order
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);
        if (!stager) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { red[e * (NMW * 64) + tid] = stat[0][e]; red[(4 + e) * (NMW * 64) + tid] = stat[1][e]; }
        }
        __syncthreads();
        if (tid < 256) {
            // column c = tid / 32, 32 lanes each: a strided partial sum, then a butterfly -- a fixed order of additions
            constexpr int NV = NMW * 64;
            const int c = tid >> 5, k = tid & 31;
            double a = 0.0;
            for (int i = k; i < NV; i += 32) a += (double)red[c * NV + i];
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
            if (k == 0) stats[(size_t)blockIdx.x * 8 + c] = a;
        }
    }
    stamp(dbg_buf, dbg, 4);
}

// ------------------------------------------------------------------------------------------------------------
// weight gradient: dW[kt][kf][ci][co] = sum_{b,t,f} a[t][f][ci] dy[t - kt dil + pad][f - kf + 2][co], a = transform(x);
// dbias[co] = sum dy.  Persistent workgroups walk the tiles; a wave keeps its ten accumulators over all its tiles and the
// workgroup reduces ONCE, in a fixed order (no atomics): partials[workgroup][NPART] (the row layout of conv2d.hip's
// backward, so that ptts_conv2d_reduce_grouped serves both).
// LDS: dy planes [NP][16 + (KT-1) dil][RS] (halo) | a planes [NP][16][RS]; reduction scratch over them at the end.
// Blocks are 18 bin groups wide here (9 pairs; 17 used).  K step = 16 rows x 2 adjacent bin groups; per kernel row kt and
// half hb of the 8 dy bins a lane's accumulator holds
//   Ck[(fi,ci)][(fo,co)] with kf = fi + 4 - 4 hb - fo   (rows m = 4 (lane >> 4) + r -> fi = lane >> 4, ci = r; n = lane & 15)
// ------------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef s16x4 __attribute__((address_space(3)))* lds_s16x4_ptr;
__device__ __forceinline__ bf16x4 tr_read(const u16* p) {
    // ds_read_b64_tr_b16: lane i of a 16-lane group receives column i of the 4 rows x 16 columns block whose row q,
    // columns 4p..4p+3 lane 4q+p of the group points at (EXEC must be full: every caller is wave-uniform)
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p)));
}

template <int DIL, int MODE, int NPL>
__global__ __launch_bounds__(THREADS, DIL <= 2 ? 2 : 1) void wgrad_kernel(
    const void* __restrict__ x, const void* __restrict__ dy, const void* __restrict__ mask_src,
    float* __restrict__ partials, int dt, Shape s, Sched sc, float alpha, int dbg, unsigned long long* dbg_buf) {
    typedef Stage<4 * (GPB + 1) + 4, 16 + (KT - 1) * DIL> SD;      // dy tile with its time halo
    typedef Stage<4 * (GPB + 1) + 4, 16> SA;                       // activation tile
    static_assert(SD::RS == SA::RS, "one row stride");
    extern __shared__ __attribute__((aligned(16))) u16 lds[];
    stamp(dbg_buf, dbg, 0);
    u16* dpl = lds;
    u16* apl = lds + NPL * SD::PS;
    const bool x_bf16 = NPL == 1 && (dt & DT_IN) != 0, dy_bf16 = NPL == 1 && (dt & DT_DY) != 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int lo = (KT - 1) * DIL - s.pad_t;      // staged dy row r <-> t = t0 - lo + r
    constexpr bool MASK = MODE == PTTS_IN_MASKMUL;

    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[KT][2];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) { acc[kt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[kt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // transposed-read address of this lane inside a 4 rows x 16 columns block: row li >> 2, bin li & 3 of the block's four.
    // bin_off(8 gp + c) = 32 gp + bin_off(c): the pair index is an immediate-free add, everything else immediates
    const int trow = 4 * lg + (li >> 2);
    const u16* abase0 = apl + trow * SA::RS + bin_off(2 + (li & 3));
    const u16* abase1 = apl + trow * SA::RS + bin_off(6 + (li & 3));
    const u16* dbase0 = dpl + trow * SD::RS + bin_off(0 + (li & 3));
    const u16* dbase1 = dpl + trow * SD::RS + bin_off(4 + (li & 3));
    const u16* dbase2 = dpl + trow * SD::RS + bin_off(8 + (li & 3));

    Slots<SD> sd; sd.init();
    Slots<SA> sa; sa.init();
    const int wg = blockIdx.x, nitems = sched_items(sc, wg);
    int item = -1;
    Work work = next_work(s, sc, wg, item, nitems);
    Pref<SD::NB, false> pd;
    Pref<SA::NB, MASK> pa;
    TilePos pos = work_pos(s, work);
    if (work.ng > 0 && !(dbg & DBG_NOSTAGE)) {
        pref_load<SD, false>(pd, sd, dy, nullptr, dy_bf16, pos.img, pos.t0 - lo, 4 * pos.g_base - 2, s.T, s.F, s.F, 4 * pos.ng + 4);
        pref_load<SA, MASK>(pa, sa, x, mask_src, x_bf16, pos.img, pos.t0, 4 * pos.g_base - 2, s.T, s.F, min(s.F, 4 * (pos.g_base + pos.ng)), 4 * pos.ng + 4);
    }
    int it = 0;
    while (work.ng > 0) {
        if (!(dbg & DBG_NOSTAGE)) {
            pref_commit<SD, PTTS_IN_NONE, NPL>(pd, sd, dpl, pos.t0 - lo, 4 * pos.g_base - 2, s.T, s.F, nullptr, nullptr, alpha, lo, 2 + 4 * pos.ng, &bsum);
            pref_commit<SA, MODE, NPL>(pa, sa, apl, pos.t0, 4 * pos.g_base - 2, s.T, s.F, nullptr, nullptr, alpha, 0, 0, nullptr);
        }
        if (it == 0) stamp(dbg_buf, dbg, 1);
        __syncthreads();
        if (it == 0) stamp(dbg_buf, dbg, 2);
        const int ng2 = (pos.ng + 1) >> 1;
        work = next_work(s, sc, wg, item, nitems);
        if (work.ng > 0) {
            pos = work_pos(s, work);
            if (!(dbg & DBG_NOSTAGE)) {
                pref_load<SD, false>(pd, sd, dy, nullptr, dy_bf16, pos.img, pos.t0 - lo, 4 * pos.g_base - 2, s.T, s.F, s.F, 4 * pos.ng + 4);
                pref_load<SA, MASK>(pa, sa, x, mask_src, x_bf16, pos.img, pos.t0, 4 * pos.g_base - 2, s.T, s.F, min(s.F, 4 * (pos.g_base + pos.ng)), 4 * pos.ng + 4);
            }
        }
        for (int gp = (wave + it) & 3; gp < ng2 && !(dbg & DBG_NOMFMA); gp += 4) {
            // A operand: a[row 4 lg + e&3][group 2 gp + (e >> 2)][(fi, ci) = li]  -- staged bin of group g, fi: 4 g + 2 + fi
            bf16x8 af[NPL];
#pragma unroll
            for (int p = 0; p < NPL; ++p) af[p] = cat(tr_read(abase0 + 32 * gp + p * SA::PS), tr_read(abase1 + 32 * gp + p * SA::PS));
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                // B operand: dy[staged row 4 lg + e&3 + (KT-1-kt) dil][staged bin 4 (g + hb) + fo][co], (fo, co) = li;
                // the upper half of hb = 0 is the lower half of hb = 1
                bf16x4 d0[NPL], d1[NPL], d2[NPL];
#pragma unroll
                for (int p = 0; p < NPL; ++p) {
                    const int off = 32 * gp + p * SD::PS + (KT - 1 - kt) * DIL * SD::RS;
                    d0[p] = tr_read(dbase0 + off);
                    d1[p] = tr_read(dbase1 + off);
                    d2[p] = tr_read(dbase2 + off);
                }
#define C2M_MM(PA_, PW_)                                                                                                 \
                { constexpr int PA = PA_ < NPL ? PA_ : 0, PW = PW_ < NPL ? PW_ : 0;                                      \
                acc[kt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[PA], cat(d0[PW], d1[PW]), acc[kt][0], 0, 0, 0);   \
                acc[kt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[PA], cat(d1[PW], d2[PW]), acc[kt][1], 0, 0, 0); }
                C2M_PRODUCTS_NPL(NPL, C2M_MM);
#undef C2M_MM
            }
        }
        if (it == 0) stamp(dbg_buf, dbg, 3);
        __syncthreads();       // planes free (and, after the last piece, dead: the LDS becomes the reduction scratch)
        ++it;
    }
    // ---- one reduction per workgroup, fixed order.  red[wave][kt][hb][r][lane] | bs[256][4]
    float* red = reinterpret_cast<float*>(lds);
    float* bs = red + 4 * KT * 2 * 4 * 64;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((wave * KT + kt) * 2 + hb) * 4 + r) * 64 + lane] = acc[kt][hb][r];
    *reinterpret_cast<f32x4*>(bs + tid * 4) = bsum;
    __syncthreads();
    float* out = partials + (size_t)blockIdx.x * NPART;
    for (int i = tid; i < KT * KF * 16; i += THREADS) {
        // dW[kt][kf][ci][co]: lane (li = 4 fo + co, lg = fi), register r = ci of acc[kt][hb], kf = fi + 4 - 4 hb - fo
        const int co = i & 3, ci = (i >> 2) & 3, kf = (i >> 4) % KF, kt = (i >> 4) / KF;
        float sum = 0.f;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
                const int fo = fi + 4 - 4 * hb - kf;
                if (fo >= 0 && fo < 4) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) sum += red[(((w * KT + kt) * 2 + hb) * 4 + ci) * 64 + 16 * fi + 4 * fo + co];
                }
            }
        out[i] = sum;
    }
    if (tid < 64) {
        // dbias: 256 lane sums x 4 channels -> 64 lanes take 16 values each, then a butterfly over the lanes of a channel
        const int ch = tid & 3, j0 = tid >> 2;
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) v += bs[(j0 * 16 + j) * 4 + ch];
        v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (tid < 4) out[KT * KF * 16 + tid] = v;
    }
    stamp(dbg_buf, dbg, 4);
}


// ------------------------------------------------------------------------------------------------------------
// FUSED backward launches (round 4, dilation 1, fp32 = three planes): the two passes of a layer that read the same staged tile run
// as ONE wave-specialised launch -- one prologue, one staging (activation, three-way split, ds_write) of the shared operand, one read
// of it from HBM -- instead of a backward-data / forward launch plus a weight-gradient launch:
//   KIND 1, first-order backward of a layer (TF's Conv2DBackpropInput + Conv2DBackpropFilter + bias gradient):
//       P = dy (20 rows with the time halo)      dx = conv(P, flipped / transposed table) . lrelu'(x)      [fwd_ws_kernel<NONE, OUTMASK>]
//       Q = a = lrelu(x) (16 rows)               dW = corr(Q, P), dbias = sum P                            [wgrad_kernel<LRELU>]
//   KIND 2, second-order sweep of the gradient penalty (backward of the backward-data pass):
//       P = a = u . lrelu'(x) (20 rows)          cot_dy = conv(P, forward table)                           [fwd_ws_kernel<MASKMUL>]
//       Q = dy (20 rows)                         dW = corr(P rows 2..17, Q)                                [wgrad_kernel<MASKMUL>]
// Same tiles, passes and MFMA order as the separate kernels, so dx / cot_dy are bit-identical to theirs; the dW partial rows are one
// per workgroup (256 instead of 512: sums grouped differently, same tolerance), reduced by ptts_conv2d_reduce_grouped as before.
// Structure = fwd_ws_kernel: waves NMW.. stage piece i + 1 (both tiles) while waves 0..NMW-1 multiply piece i (the convolution over P,
// then the wave's share of the weight-gradient pairs), buffers handed over by the LDS counters (bounded polls -> status word).
// Rows are 76 staged bins wide (18 bin groups: the weight gradient's K step takes two adjacent groups).  The second group of a
// piece's last pair may lie beyond the piece (odd number of groups): its half of the A fragment is zeroed by a wave-uniform select
// (P carries the convolution's frequency halo there, which the weight gradient must not count).
// ------------------------------------------------------------------------------------------------------------
// SPLIT (round 4, second form): TWELVE waves -- 0..3 only convolve, 4..7 only run the weight-gradient products, 8..11 stage -- three per SIMD
// (<= 168 registers each), so that every SIMD has one wave of each kind: a SIMD's single multiplying wave spends as long on its LDS reads as on
// its MFMAs and nothing overlaps the two (tools/c2m_fused_probe2.py: 5 us per piece for 1.75 us of matrix work).
#if C2M_PROBE_STAMPS
// (probe build, tools/c2m_fused_stamps.py) the timeline of piece 3 of every workgroup of the fused backward kernel: wave 0 (multiplying)
// slots 0..5, the first staging wave 8..12; read back with ptts_conv2d_mfma_probe_stamps
__device__ unsigned long long g_bw_stamps[256 * 16];
#define C2M_BTS(SLOT) do { if (it == 3 && lane == 0 && blockIdx.x < 256 && (wave8 == 0 || wave8 == NCW)) g_bw_stamps[blockIdx.x * 16 + (wave8 ? 8 : 0) + (SLOT)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define C2M_BTS(SLOT) do { } while (0)
#endif
// AFF (KIND 1, round 4): the layer's input was lrelu(q_scale x + q_shift) -- a BatchNormalization in front of it, the generator's stack
// (networktts.py:122-126).  Q is staged with that affine, dx = conv^T(dy) lrelu'(q_scale x + q_shift) q_scale, and the workgroup's row
// carries, behind dW and dbias, the sums of conv^T(dy) lrelu'(.) x and conv^T(dy) lrelu'(.) over its pixels: the gradients of scale and shift.
template <int KIND, int NPL, bool SPLIT, bool AFF = false>
__global__ __launch_bounds__((SPLIT ? 12 : 8) * 64, 1) void bwd_ws_kernel(
    const void* __restrict__ psrc, const void* __restrict__ qsrc, const void* __restrict__ mask_src, const u16* __restrict__ tab,
    void* __restrict__ y, float* __restrict__ partials, Shape s, Sched sc, float alpha, int dbg, unsigned* status,
    const float* __restrict__ q_scale = nullptr, const float* __restrict__ q_shift = nullptr) {
    static_assert(!AFF || (KIND == 1 && !SPLIT), "the affine form is the first-order backward's");
    constexpr int DIL = 1, NMW = 4, NCW = SPLIT ? 8 : 4, NT = (NCW + 4) * 64;      // waves that share a piece's groups / pairs; consumer waves; threads
    typedef Stage<4 * (GPB + 1) + 4, 16 + (KT - 1) * DIL> SP;                              // the convolved tile, with its time halo
    typedef Stage<4 * (GPB + 1) + 4, KIND == 1 ? 16 : 16 + (KT - 1) * DIL> SQ;             // the other operand of the weight gradient
    static_assert(SP::RS == SQ::RS, "one row stride");
    constexpr int MODE_P = KIND == 1 ? PTTS_IN_NONE : PTTS_IN_MASKMUL;
    constexpr int MODE_Q = KIND == 1 ? PTTS_IN_LRELU : PTTS_IN_NONE;
    constexpr bool OUTMASK = KIND == 1, PMASK = KIND == 2;
    constexpr int BUF = NPL * (SP::PS + SQ::PS);                                           // elements of one buffer (P planes | Q planes)
    extern __shared__ __attribute__((aligned(16))) u16 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool stager = wave8 >= NCW;
    const int wave = wave8 & (NMW - 1);
    const int lo = (KT - 1) * DIL - s.pad_t;          // staged P row r <-> t = t0 - lo + r (pad_t = 2: symmetric halo of two rows)
    const int wg = blockIdx.x, nitems = sched_items(sc, wg);

    int* const flags = reinterpret_cast<int*>(lds + 2 * BUF);       // ready[2] | done[2]
    u16* const wl = lds + 2 * BUF + 32;                             // (C2M_BW_WREG == 0: the operand table, behind the counters)
    if (tid == 0) { flags[0] = 0; flags[1] = 0; flags[2] = 0; flags[3] = 0; }
    if (!C2M_BW_WREG)
        for (int i = tid; i < KT * NPL * TKP / 8; i += NT)
            *reinterpret_cast<bf16x8*>(wl + (size_t)i * 8) = *reinterpret_cast<const bf16x8*>(tab + (size_t)i * 8);
    const bool force_timeout = (dbg & DBG_FORCE_TIMEOUT) != 0;
    auto wait_for = [&](int idx, int target) {
        const int bound = force_timeout ? 8 : (1 << 14);
        if (force_timeout && !stager) target += 1 << 20;
        int r = 0;
        for (; r < bound; ++r) {
            if (__hip_atomic_load(flags + idx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (r == bound && lane == 0) raise_status(status, STATUS_SLOT_C2M, STATUS_C2M_HANDOFF);
    };
    // The counters order LDS traffic only: a staging wave's commit (ds_write) before `ready`, a multiplying wave's fragment reads before
    // `done`.  The LDS executes a wave's operations in order and lgkmcnt(0) says they have completed, so the count goes up with a
    // RELAXED add behind an explicit s_waitcnt lgkmcnt(0) (a release add also drains the wave's global stores: vmcnt(0)).
    auto signal = [&](int idx) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(flags + idx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);          // the end-of-launch reduction scratch lies over the (then dead) planes
    float* bs = red + 4 * KT * 2 * 4 * 64;

    // The two roles are two separate programs (no variable of one is live in the other: the register allocation is the maximum of the
    // two, not their sum); both end with the same pair of workgroup barriers.  ready[b] counts the staging waves' commits into buffer
    // b (piece p lives in buffer p & 1, use u = p >> 1: a multiplying wave starts it at ready == 4 (u + 1)), done[b] the multiplying
    // waves that have finished with it (a staging wave overwrites it at done == NMW u).
    if (stager) {
        Slots<SP> slp; slp.init(tid - NCW * 64); slp.init_rows(s.F);
        Slots<SQ> slq; slq.init(tid - NCW * 64); slq.init_rows(s.F);
        Pref<SP::NB, PMASK> pp;
        Pref<SQ::NB, false> pq;
        f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
        // the two tiles of a piece: P from t0 - lo (20 rows); Q from t0 (KIND 1: the layer input, 16 rows, zero beyond the piece's own
        // bins) or from t0 - lo (KIND 2: the incoming gradient with its halo)
        auto load_piece = [&](const TilePos& p) {
            if (rows_tile(s, p)) {
                pref_load_rows<SP, PMASK>(pp, slp, psrc, mask_src, p.img, p.t0 - lo, s.T, s.F);
                pref_load_rows<SQ, false>(pq, slq, qsrc, nullptr, p.img, KIND == 1 ? p.t0 : p.t0 - lo, s.T, s.F);
                return;
            }
            pref_load<SP, PMASK>(pp, slp, psrc, mask_src, false, p.img, p.t0 - lo, 4 * p.g_base - 2, s.T, s.F, s.F, 4 * p.ng + 4);
            if (KIND == 1) pref_load<SQ, false>(pq, slq, qsrc, nullptr, false, p.img, p.t0, 4 * p.g_base - 2, s.T, s.F, min(s.F, 4 * (p.g_base + p.ng)), 4 * p.ng + 4);
            else pref_load<SQ, false>(pq, slq, qsrc, nullptr, false, p.img, p.t0 - lo, 4 * p.g_base - 2, s.T, s.F, s.F, 4 * p.ng + 4);
        };
        auto commit_piece = [&](const TilePos& p, u16* buf) {
            // the bias gradient rides on the staging of dy (KIND 1 only: the second-order sweep has none): its own 16 rows and bins
            pref_commit<SP, MODE_P, NPL>(pp, slp, buf, p.t0 - lo, 4 * p.g_base - 2, s.T, s.F, nullptr, nullptr, alpha, lo, 2 + 4 * p.ng, KIND == 1 ? &bsum : nullptr);
            pref_commit<SQ, MODE_Q, NPL>(pq, slq, buf + NPL * SP::PS, KIND == 1 ? p.t0 : p.t0 - lo, 4 * p.g_base - 2, s.T, s.F, AFF ? q_scale : nullptr, AFF ? q_shift : nullptr, alpha, 0, 0, nullptr);
        };
        // the same with the loads of the NEXT piece (a whole-row tile) going out slot by slot behind the commits (RowsReload)
        auto commit_reload = [&](const TilePos& p, u16* buf, const TilePos& pn) {
            const RowsReload<SP, PMASK> rp(slp, psrc, mask_src, pn.img, pn.t0 - lo, s.T, s.F);
            const RowsReload<SQ, false> rq(slq, qsrc, nullptr, pn.img, KIND == 1 ? pn.t0 : pn.t0 - lo, s.T, s.F);
            pref_commit<SP, MODE_P, NPL>(pp, slp, buf, p.t0 - lo, 4 * p.g_base - 2, s.T, s.F, nullptr, nullptr, alpha, lo, 2 + 4 * p.ng, KIND == 1 ? &bsum : nullptr, rp);
            pref_commit<SQ, MODE_Q, NPL>(pq, slq, buf + NPL * SP::PS, KIND == 1 ? p.t0 : p.t0 - lo, 4 * p.g_base - 2, s.T, s.F, AFF ? q_scale : nullptr, AFF ? q_shift : nullptr, alpha, 0, 0, nullptr, rq);
        };
        const bool stage = !(dbg & DBG_NOSTAGE);              // (measurement hooks of tools/c2m_fused_probe2.py: garbage results)
        int it = 0;
        PieceWalk walk;
        walk.init(s, sc, wg, nitems);
        TilePos p = {0, 0, 0, 0};
        bool have = walk.next(s, sc, wg, p);
        if (have && stage) load_piece(p);
        while (have) {
            C2M_BTS(0);
            TilePos pn = p;
            const bool haven = walk.next(s, sc, wg, pn);
            const bool reload = haven && stage && rows_tile(s, pn);
            wait_for(2 + (it & 1), NCW * (it >> 1));
            C2M_BTS(1);
            if (stage) {
                if (reload) commit_reload(p, lds + (it & 1) * BUF, pn);
                else commit_piece(p, lds + (it & 1) * BUF);
            }
            C2M_BTS(2);
            signal(it & 1);
            C2M_BTS(3);
            if (haven && stage && !reload) load_piece(pn);
            C2M_BTS(4);
            p = pn; have = haven;
            ++it;
        }
        __syncthreads();
        *reinterpret_cast<f32x4*>(bs + (tid - NCW * 64) * 4) = bsum;
    } else {
        const int li = lane & 15, lg = lane >> 4;
        const bool do_conv = !SPLIT || wave8 < NMW, do_wg = !SPLIT || wave8 >= NMW;
        f32x4 astat[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};       // (AFF) this lane's sums for the gradients of scale / shift
        // ---- the convolution over P: dx (KIND 1, masked by the layer input) / cot_dy (KIND 2)
        auto conv_role = [&](auto&& also) {
            bf16x8 wf[KT][NPL];
            const u16* wa = tab + (li & 3) * TROW + (2 * lg - (li >> 2) + 3) * C;
            if (C2M_BW_WREG) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int q = 0; q < NPL; ++q) {
                        const u16* wp = wa + (kt * NPL + q) * TKP;
                        wf[kt][q] = cat(*reinterpret_cast<const bf16x4*>(wp), *reinterpret_cast<const bf16x4*>(wp + 4));
                        if (C2M_PIN_WF == 1) asm volatile("" : "+v"(wf[kt][q]));      // (see its definition)
                    }
                if (C2M_PIN_WF == 2) {
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                        for (int q = 0; q < NPL; ++q) asm volatile("" : "+v"(wf[kt][q]));
                }
            }
            const f32x4 bv0 = {0.f, 0.f, 0.f, 0.f};
            f32x4 aff[2] = {f32x4{1.f, 1.f, 1.f, 1.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            if (AFF) { aff[0] = *reinterpret_cast<const f32x4*>(q_scale); aff[1] = *reinterpret_cast<const f32x4*>(q_shift); }
            int it = 0;
            PieceWalk walk;
            walk.init(s, sc, wg, nitems);
            TilePos cur = {0, 0, 0, 0};
            bool have = walk.next(s, sc, wg, cur);
            while (have) {
                const u16* pcur = lds + (it & 1) * BUF;
                C2M_BTS(0);
                wait_for(it & 1, 4 * ((it >> 1) + 1));
                C2M_BTS(1);
                fwd_piece<SP, DIL, OUTMASK, PMASK, NPL, C2M_BW_WREG != 0, NMW, C2M_BW_NMAX, AFF>(pcur, wl, wf, cur, s, wave, lane, it, bv0, qsrc, y, false, alpha, !(dbg & DBG_NOSTORE), (dbg & (DBG_NOMFMA | 16)) != 0, AFF ? astat : nullptr, AFF ? aff : nullptr);
                C2M_BTS(2);
                also(cur, pcur, it);
                C2M_BTS(3);
                signal(2 + (it & 1));
                C2M_BTS(4);
                have = walk.next(s, sc, wg, cur);
                C2M_BTS(5);
                ++it;
            }
        };
        // ---- the weight gradient: K step = 16 rows x 2 adjacent bin groups, both operands read transposed
        f32x4 acc[KT][2];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) { acc[kt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[kt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // transposed-read addresses of this lane inside a 4 rows x 16 columns block: bin li & 3 of the block's row li >> 2.  WHICH four
        // time rows make up the block of lane group lg is free (the reduction index of both operands is permuted alike): the 32 lanes
        // the LDS serves per cycle (lg = 0, 1, then 2, 3) take the EVEN rows 0, 2, .. 14, then the odd ones -- the a-fragments' 8-bank
        // windows of rows two apart (312 dwords = 56 mod 64 banks) tile the 64 banks.  (The dy-fragments' windows are split in two by
        // the forward pass's unit swizzle and stay two-way conflicted either way: SQ_LDS_BANK_CONFLICT 30 -> 28 % of the LDS cycles.)
        const int trow = 2 * (4 * (lg & 1) + (li >> 2)) + (lg >> 1);
        const int aoff = (KIND == 1 ? NPL * SP::PS : lo * SP::RS) + trow * SP::RS;          // a: Q (16 rows) / P's own rows
        constexpr int APS = KIND == 1 ? SQ::PS : SP::PS;                                     // ... and its plane stride
        const int doff = (KIND == 1 ? 0 : NPL * SP::PS) + trow * SP::RS;                     // dy with its halo: P / Q
        auto wgrad_piece = [&](const TilePos& cur, const u16* pcur, int it) {
            const u16* abase0 = pcur + aoff + bin_off(2 + (li & 3));
            const u16* abase1 = pcur + aoff + bin_off(6 + (li & 3));
            const u16* dbase0 = pcur + doff + bin_off(0 + (li & 3));
            const u16* dbase1 = pcur + doff + bin_off(4 + (li & 3));
            const u16* dbase2 = pcur + doff + bin_off(8 + (li & 3));
            const int ng2 = (dbg & (DBG_NOMFMA | 32)) ? 0 : (cur.ng + 1) >> 1;
            for (int gp = (wave + it + C2M_BW_ROT) & 3; gp < ng2; gp += 4) {
                const bool second = 2 * gp + 1 < cur.ng;         // wave-uniform
                bf16x8 af[NPL];
#pragma unroll
                for (int p = 0; p < NPL; ++p) {
                    bf16x4 h0 = tr_read(abase0 + 32 * gp + p * APS), h1 = tr_read(abase1 + 32 * gp + p * APS);
                    if (!second) h1 = __builtin_bit_cast(bf16x4, (u32x2){0u, 0u});
                    af[p] = cat(h0, h1);
                }
                // the dy fragments of kernel row kt + 1 are requested before the MFMAs of row kt (two register sets)
                bf16x4 dq[2][3][NPL];
                auto read_d = [&](int kt, bf16x4 (&d)[3][NPL]) {
#pragma unroll
                    for (int p = 0; p < NPL; ++p) {
                        const int off = 32 * gp + p * SP::PS + (KT - 1 - kt) * DIL * SP::RS;
                        d[0][p] = tr_read(dbase0 + off);
                        d[1][p] = tr_read(dbase1 + off);
                        d[2][p] = tr_read(dbase2 + off);
                    }
                };
                read_d(0, dq[0]);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    if (kt + 1 < KT) read_d(kt + 1, dq[(kt + 1) & 1]);
#define C2M_MM(PA_, PW_)                                                                                                 \
                    { constexpr int PA = PA_ < NPL ? PA_ : 0, PW = PW_ < NPL ? PW_ : 0;                                  \
                    acc[kt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[PA], cat(dq[kt & 1][0][PW], dq[kt & 1][1][PW]), acc[kt][0], 0, 0, 0);   \
                    acc[kt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[PA], cat(dq[kt & 1][1][PW], dq[kt & 1][2][PW]), acc[kt][1], 0, 0, 0); }
                    C2M_PRODUCTS_NPL(NPL, C2M_MM);
#undef C2M_MM
                }
            }
        };
        if (!SPLIT) {
            conv_role(wgrad_piece);                        // one wave does both, piece by piece
        } else if (do_conv) {
            conv_role([](const TilePos&, const u16*, int) {});
        } else {
            int it = 0;
            PieceWalk walk;
            walk.init(s, sc, wg, nitems);
            TilePos cur = {0, 0, 0, 0};
            bool have = walk.next(s, sc, wg, cur);
            while (have) {
                const u16* pcur = lds + (it & 1) * BUF;
                wait_for(it & 1, 4 * ((it >> 1) + 1));
                wgrad_piece(cur, pcur, it);
                signal(2 + (it & 1));
                have = walk.next(s, sc, wg, cur);
                ++it;
            }
        }
        __syncthreads();
        // ---- one reduction per workgroup, fixed order (wgrad_kernel's): red[wave][kt][hb][r][lane]
        if (do_wg) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(((wave * KT + kt) * 2 + hb) * 4 + r) * 64 + lane] = acc[kt][hb][r];
        }
        if (AFF) {
            float* stl = bs + 256 * 4;                     // behind the bias sums: [8][256 multiplying lanes]
#pragma unroll
            for (int e = 0; e < 4; ++e) { stl[e * 256 + tid] = astat[0][e]; stl[(4 + e) * 256 + tid] = astat[1][e]; }
        }
    }
    __syncthreads();
    float* out = partials + (size_t)blockIdx.x * NPART;
    if (AFF && tid >= 256 && tid < 512) {
        // the eight affine sums of the workgroup: column c = 32 lanes, a strided partial sum and a butterfly (a fixed order)
        const float* stl = bs + 256 * 4;
        const int t2 = tid - 256, c = t2 >> 5, k = t2 & 31;
        double a = 0.0;
        for (int i = k; i < 256; i += 32) a += (double)stl[c * 256 + i];
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
        if (k == 0) out[KT * KF * 16 + 4 + c] = (float)a;
    }
    for (int i = tid; i < KT * KF * 16; i += NT) {
        const int co = i & 3, ci = (i >> 2) & 3, kf = (i >> 4) % KF, kt = (i >> 4) / KF;
        float sum = 0.f;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
                const int fo = fi + 4 - 4 * hb - kf;
                if (fo >= 0 && fo < 4) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) sum += red[(((w * KT + kt) * 2 + hb) * 4 + ci) * 64 + 16 * fi + 4 * fo + co];
                }
            }
        out[i] = sum;
    }
    if (tid < 64) {
        const int ch = tid & 3, j0 = tid >> 2;
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) v += bs[(j0 * 16 + j) * 4 + ch];
        v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (tid < 4) out[KT * KF * 16 + tid] = v;
    }
}

}  // namespace c2m
}  // namespace ptts

using namespace ptts;
using namespace ptts::c2m;

// measurement hook: phase switches / per-workgroup stamps (buffer of gridDim.x * 8 uint64, or NULL); 0 restores the product path
extern "C" int ptts_conv2d_mfma_debug(int flags, void* stamp_buf) {
    g_dbg = flags;
    g_dbg_buf = (unsigned long long*)stamp_buf;
    if ((flags & DBG_STAMPS) && !stamp_buf) g_dbg &= ~DBG_STAMPS;
    return PTTS_OK;
}

extern "C" size_t ptts_conv2d_mfma_table_bytes(int KT_) { return (size_t)KT_ * NP * TKP * sizeof(u16); }

// planes: 3 = fp32 arithmetic (three-way split), 1 = bf16 arithmetic (the kernel rounded to bf16: the "weights copy in bf16")
extern "C" int ptts_conv2d_mfma_tables(const float* w, void* table_fwd, void* table_bwd, int KT_, int KF_, int Cin, int Cout,
                                       int planes, void* stream) {
    PTTS_REQUIRE(w && (table_fwd || table_bwd), "conv2d_mfma_tables: null pointer");
    PTTS_REQUIRE(KT_ == 5 && KF_ == 5 && Cin == 4 && Cout == 4, "conv2d_mfma_tables: only 5x5, 4 -> 4 channels (got %dx%d, %d -> %d)", KT_, KF_, Cin, Cout);
    PTTS_REQUIRE(planes == 1 || planes == 3, "conv2d_mfma_tables: planes must be 1 (bf16) or 3 (fp32 split), got %d", planes);
    hipLaunchKernelGGL(toeplitz_table_kernel, dim3(2), dim3(256), 0, (hipStream_t)stream, w, (u16*)table_fwd, (u16*)table_bwd, planes);
    return check_launch("conv2d_mfma_tables");
}

// the tables of up to TAB_GROUP kernels in one launch (blockIdx.y = kernel): a stack's 7 layers need theirs after every update
constexpr int TAB_GROUP = 16;
struct TabGroupArgs { const float* w[TAB_GROUP]; u16* fwd[TAB_GROUP]; u16* bwd[TAB_GROUP]; int npl; };
__global__ void toeplitz_table_grouped_kernel(TabGroupArgs a) {
    const int i = blockIdx.y;
    const int transposed = blockIdx.x;
    u16* tab = transposed ? a.bwd[i] : a.fwd[i];
    if (!tab) return;
    const float* __restrict__ w = a.w[i];
    const int npl = a.npl;
    const int idx = threadIdx.x;
    if (idx >= KT * C * TSLOTS) return;
    const int slot = idx % TSLOTS, oc = (idx / TSLOTS) % C, kt = idx / (TSLOTS * C);
    const int kf = slot - 3;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (kf >= 0 && kf < KF) {
#pragma unroll
        for (int ic = 0; ic < C; ++ic)
            v[ic] = transposed ? w[(((KT - 1 - kt) * KF + (KF - 1 - kf)) * C + oc) * C + ic] : w[((kt * KF + kf) * C + ic) * C + oc];
    }
    bf16x4 h1, h2, h3;
    split3(v, h1, h2, h3);
    u16* d = tab + (kt * npl * C + oc) * TROW + slot * C;
    *reinterpret_cast<bf16x4*>(d) = h1;
    if (npl == 3) {
        *reinterpret_cast<bf16x4*>(d + TKP) = h2;
        *reinterpret_cast<bf16x4*>(d + 2 * TKP) = h3;
    }
}

extern "C" int ptts_conv2d_mfma_tables_grouped(const float* const* w, void* const* table_fwd, void* const* table_bwd, int n, int planes,
                                               void* stream) {
    PTTS_REQUIRE(w && table_fwd && table_bwd && n > 0, "conv2d_mfma_tables_grouped: nothing to build");
    PTTS_REQUIRE(planes == 1 || planes == 3, "conv2d_mfma_tables_grouped: planes must be 1 (bf16) or 3 (fp32 split), got %d", planes);
    for (int base = 0; base < n; base += TAB_GROUP) {
        TabGroupArgs a;
        const int m = n - base < TAB_GROUP ? n - base : TAB_GROUP;
        for (int i = 0; i < m; ++i) {
            PTTS_REQUIRE(w[base + i] && (table_fwd[base + i] || table_bwd[base + i]), "conv2d_mfma_tables_grouped: null pointer (kernel %d)", base + i);
            a.w[i] = w[base + i]; a.fwd[i] = (u16*)table_fwd[base + i]; a.bwd[i] = (u16*)table_bwd[base + i];
        }
        a.npl = planes;
        hipLaunchKernelGGL(toeplitz_table_grouped_kernel, dim3(2, (unsigned)m), dim3(256), 0, (hipStream_t)stream, a);
    }
    return check_launch("conv2d_mfma_tables_grouped");
}

namespace {
constexpr size_t LDS_MAX = 160 * 1024;
constexpr int NCU = 256;

unsigned magic32(int d) { return (unsigned)(((1ULL << 32) + (unsigned)d - 1) / (unsigned)d); }

Shape make_shape(int B, int T, int F, int pad_t) {
    Shape s;
    s.T = T; s.F = F; s.NG = (F + 3) / 4;
    s.nfb = (s.NG + GPB - 1) / GPB;
    s.ntt = (T + 15) / 16;
    s.ntiles = B * s.ntt * s.nfb;
    s.magic_nfb = s.nfb > 1 ? magic32(s.nfb) : 0u;
    s.magic_ntt = s.ntt > 1 ? magic32(s.ntt) : 0u;
    s.pad_t = pad_t;
    return s;
}
bool shape_ok(const Shape& s) { return s.ntiles > 0 && s.ntiles < (1 << 20) && s.nfb < 4096 && s.ntt < 4096; }

// stage buffers of the dilation-1 forward kernel: 2 (the default: two workgroups per CU, one barrier per piece) or 1
// (PTTS_C2M_DOUBLE=0: three per CU).  Launched back to back the two forms take the same time per layer (20-22 us); inside
// the critic step, where the three evaluations run on three streams, the double-buffered form leaves a third of the
// register file to the other streams' kernels and the step is 3 % shorter (7.00 against 7.23 ms, same box).
bool fwd_double_buffered() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PTTS_C2M_DOUBLE"); v = e ? atoi(e) : 1; }
    return v != 0;
}
// dilation 1, fp32 (three planes): the wave-specialised eight-wave kernel (PTTS_C2M_WS, default on) or the four-wave forms below
int fwd_wave_specialised() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PTTS_C2M_WS"); v = e ? atoi(e) : 1; }
    return v;           // 0 never, 1 always, 2 only the layers without a BatchNorm affine on their input (the critic's)
}
// multiplying waves of the wave-specialised kernel: 4 (one per SIMD) or 8 (two per SIMD, at most two bin groups per pass: 170 registers)
int fwd_ws_waves() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PTTS_C2M_WS_WAVES"); v = e ? atoi(e) : 4; }
    return v == 8 ? 8 : 4;
}
// the dilation-1 double-buffered kernel with its operand table in registers (the default) or in the LDS (PTTS_C2M_WREG=0):
// same box, whole train step: critic step 5.07 -> 4.97 ms with the registers
bool fwd_table_in_registers() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PTTS_C2M_WREG"); v = e ? atoi(e) : 1; }
    return v != 0;
}
template <int DIL, int NPL, int NB> constexpr size_t lds_fwd() { return ((size_t)NB * NPL * Stage<4 * GPB + 4, 16 + (KT - 1) * DIL>::PS + (size_t)KT * NPL * TKP) * sizeof(u16); }
template <int DIL, int NPL> constexpr size_t lds_wgrad() {
    const size_t planes = (size_t)NPL * (Stage<4 * (GPB + 1) + 4, 16 + (KT - 1) * DIL>::PS + Stage<4 * (GPB + 1) + 4, 16>::PS) * sizeof(u16);
    const size_t red = (size_t)(4 * KT * 2 * 4 * 64 + THREADS * 4) * sizeof(float);
    return planes > red ? planes : red;
}
// grid and work list (Sched, above).  Measurement hook: bits 8-10 of the debug flags override the number of pieces of a
// workgroup's first tile (default 1: cutting it measured slower, the start-up is not bandwidth-bound), bits 12-14 the largest number of pieces of a left-over tile (default 4).
Sched sched_for(int ntiles, size_t lds, int max_per_cu, int unit) {
    int per_cu = (int)(LDS_MAX / lds);
    if (per_cu > max_per_cu) per_cu = max_per_cu;
    if (per_cu < 1) per_cu = 1;
    const int maxg = NCU * per_cu;
    int fs = (g_dbg >> 8) & 7, ts = (g_dbg >> 12) & 7;
    if (fs == 0) fs = 1;
    if (ts == 0) ts = 4;
    Sched c;
    c.unit = unit;
    if (ntiles >= maxg) {
        c.G = maxg; c.R = ntiles / maxg; c.L = ntiles % maxg;
        c.S = c.L > 0 ? std::max(1, std::min(ts, maxg / c.L)) : 1;
        c.FS = fs;
    } else {
        // fewer tiles than workgroup slots: every tile in S pieces, one piece per workgroup
        c.S = std::max(1, std::min(ts, maxg / ntiles));
        c.G = ntiles * c.S; c.R = 0; c.L = ntiles; c.FS = 1;
    }
    return c;
}
}  // namespace

// The forward pass of a 4 -> 4 layer (fp32 maps, dilation 1, LeakyReLU / BatchNorm-affine + LeakyReLU or no input transform) that also leaves
// the per-workgroup sums of its outputs for the BatchNormalization layer behind it: stats[*nrows_out][8] doubles (channel sums, channel sums of
// squares), to be finished by ptts_bn_finalize_partials.  capacity_rows >= 256.
extern "C" int ptts_conv2d_mfma_fwd_stats_supported(int F, int dil_t, int in_mode) {
    return (F >= 1 && dil_t == 1 && (in_mode == PTTS_IN_NONE || in_mode == PTTS_IN_LRELU) && fwd_wave_specialised() == 1 && fwd_ws_waves() != 8) ? 1 : 0;
}
extern "C" int ptts_conv2d_mfma_fwd_stats(const float* x, const void* table, const float* bias, const float* in_scale, const float* in_shift,
                                          float* y, int B, int T, int F, int KT_, int pad_t, int in_mode, float alpha,
                                          double* stats, int capacity_rows, int* nrows_out, void* stream) {
    PTTS_REQUIRE(x && table && y && stats && nrows_out, "conv2d_mfma_fwd_stats: null pointer");
    PTTS_REQUIRE(KT_ == KT && B > 0 && T > 0 && F > 0, "conv2d_mfma_fwd_stats: bad shape B=%d T=%d F=%d KT=%d", B, T, F, KT_);
    PTTS_REQUIRE(ptts_conv2d_mfma_fwd_stats_supported(F, 1, in_mode), "conv2d_mfma_fwd_stats: unsupported (in_mode %d)", in_mode);
    PTTS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv2d_mfma_fwd_stats: scale/shift must come together");
    PTTS_REQUIRE(in_mode == PTTS_IN_LRELU || !in_scale, "conv2d_mfma_fwd_stats: scale/shift need PTTS_IN_LRELU");
    PTTS_REQUIRE(pad_t >= 0 && pad_t <= KT - 1, "conv2d_mfma_fwd_stats: bad pad_t %d", pad_t);
    PTTS_REQUIRE(alpha >= 0.f && alpha <= 1.f, "conv2d_mfma_fwd_stats: LeakyReLU slope %g outside [0, 1]", alpha);
    PTTS_REQUIRE((long long)(T + 64) * F * C < (1LL << 31), "conv2d_mfma_fwd_stats: utterance too large for 32-bit tile offsets");
    if (int rc = check_status("conv2d_mfma_fwd_stats")) return rc;
    const Shape s = make_shape(B, T, F, pad_t);
    PTTS_REQUIRE(shape_ok(s), "conv2d_mfma_fwd_stats: too many tiles");
    constexpr size_t lds = lds_fwd<1, 3, 2>();
    const Sched sc = sched_for(s.ntiles, lds, 1, 1);
    PTTS_REQUIRE(capacity_rows >= sc.G, "conv2d_mfma_fwd_stats: room for %d rows of sums, %d needed", capacity_rows, sc.G);
    hipStream_t st = (hipStream_t)stream;
#define C2M_ST(MODE)                                                                                                         \
    do {                                                                                                                     \
        static bool attr = false;                                                                                            \
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_ws_kernel<MODE, false, 3, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX); attr = true; } \
        hipLaunchKernelGGL((fwd_ws_kernel<MODE, false, 3, 4, true>), dim3(sc.G), dim3(8 * 64), lds, st, x, (const u16*)table, bias,    \
                           in_scale, in_shift, nullptr, nullptr, y, 0, s, sc, alpha, g_dbg, g_dbg_buf, status_words(), stats);           \
    } while (0)
    if (in_mode == PTTS_IN_LRELU) C2M_ST(PTTS_IN_LRELU); else C2M_ST(PTTS_IN_NONE);
#undef C2M_ST
    *nrows_out = sc.G;
    return check_launch("conv2d_mfma_fwd_stats");
}

#if C2M_PROBE_STAMPS
extern "C" int ptts_conv2d_mfma_probe_stamps(void* host_dst, int clear) {
    if (clear) {
        static unsigned long long zeros[256 * 16];
        return hipMemcpyToSymbol(HIP_SYMBOL(ptts::c2m::g_bw_stamps), zeros, sizeof(zeros)) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(ptts::c2m::g_bw_stamps), sizeof(unsigned long long) * 256 * 16) == hipSuccess ? 0 : -1;
}
#endif
// 1 when the shape has a matrix-core kernel, else 0: 5x5, 4 -> 4 channels, time dilation 1, 2, 4 or 8
extern "C" int ptts_conv2d_mfma_supported(int F, int Cin, int Cout, int KT_, int KF_, int dil_t) {
    if (!(Cin == 4 && Cout == 4 && KT_ == 5 && KF_ == 5 && F >= 1)) return 0;
    return (dil_t == 1 || dil_t == 2 || dil_t == 4 || dil_t == 8) ? 1 : 0;
}

// y = bias + conv(transform(x), w) through the forward table; with out_mask: y *= (out_mask > 0 ? 1 : alpha)
// (the backward-data pass: x = dy, table = the transposed table, pad_t = (KT-1) dil - pad_t(forward), out_mask = the
// forward layer's pre-activation input).
// planes 3: fp32 arithmetic, x / y fp32.  planes 1: bf16 arithmetic (dilation 1): x and mask_src are bf16 when
// in_bf16, y and out_mask are bf16 when out_bf16, fp32 otherwise.
extern "C" int ptts_conv2d_mfma_fwd(const void* x, const void* table, const float* bias, const float* in_scale,
                                    const float* in_shift, const void* mask_src, const void* out_mask, void* y,
                                    int B, int T, int F, int KT_, int dil_t, int pad_t, int in_mode, float alpha,
                                    int planes, int in_bf16, int out_bf16, void* stream) {
    if (int rc = check_status("conv2d_mfma_fwd")) return rc;      // an earlier launch reported a failed hand-off: sticky
    PTTS_REQUIRE(x && table && y, "conv2d_mfma_fwd: null tensor");
    PTTS_REQUIRE(B > 0 && T > 0 && F > 0 && KT_ == 5, "conv2d_mfma_fwd: bad dims B=%d T=%d F=%d KT=%d", B, T, F, KT_);
    PTTS_REQUIRE(dil_t == 1 || dil_t == 2 || dil_t == 4 || dil_t == 8, "conv2d_mfma_fwd: time dilation %d has no kernel (1, 2, 4, 8)", dil_t);
    PTTS_REQUIRE(in_mode >= 0 && in_mode <= 2, "conv2d_mfma_fwd: bad in_mode %d", in_mode);
    PTTS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv2d_mfma_fwd: scale/shift must come together");
    PTTS_REQUIRE(in_mode != PTTS_IN_MASKMUL || mask_src, "conv2d_mfma_fwd: MASKMUL needs mask_src");
    PTTS_REQUIRE(in_mode == PTTS_IN_LRELU || !in_scale, "conv2d_mfma_fwd: scale/shift need PTTS_IN_LRELU");
    PTTS_REQUIRE(pad_t >= 0 && pad_t <= (KT - 1) * dil_t, "conv2d_mfma_fwd: bad pad_t %d", pad_t);
    PTTS_REQUIRE(alpha >= 0.f && alpha <= 1.f, "conv2d_mfma_fwd: LeakyReLU slope %g outside [0, 1]", alpha);
    PTTS_REQUIRE((long long)(T + 64) * F * C < (1LL << 31), "conv2d_mfma_fwd: utterance too large for 32-bit tile offsets");
    PTTS_REQUIRE(planes == 3 || (planes == 1 && dil_t == 1), "conv2d_mfma_fwd: planes %d / dilation %d: bf16 arithmetic is built for dilation 1", planes, dil_t);
    PTTS_REQUIRE(planes == 1 || (!in_bf16 && !out_bf16), "conv2d_mfma_fwd: bf16 tensors need planes == 1");
    const Shape s = make_shape(B, T, F, pad_t);
    PTTS_REQUIRE(shape_ok(s), "conv2d_mfma_fwd: too many tiles");
    hipStream_t st = (hipStream_t)stream;
    const bool om = out_mask != nullptr;
    const int dt = (in_bf16 ? DT_IN : 0) | (out_bf16 ? DT_OUT : 0);
#define C2M_LB(DIL, MODE, OM, NPL, NB, WR)                                                                               \
    do {                                                                                                                 \
        constexpr size_t lds = lds_fwd<DIL, NPL, NB>();                                                                  \
        static_assert(lds <= LDS_MAX, "tile does not fit the LDS");                                                      \
        static bool attr = false;                                                                                        \
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_kernel<DIL, MODE, OM, NPL, NB, WR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX); attr = true; } \
        const Sched sc = sched_for(s.ntiles, lds, NB == 2 ? 2 : (DIL == 1 ? 3 : (DIL == 2 ? 2 : 1)), 1);                  \
        hipLaunchKernelGGL((fwd_kernel<DIL, MODE, OM, NPL, NB, WR>), dim3(sc.G), dim3(THREADS), lds, st, x, (const u16*)table, bias, \
                           in_scale, in_shift, mask_src, out_mask, y, dt, s, sc, alpha, g_dbg, g_dbg_buf);               \
    } while (0)
#define C2M_WSN(MODE, OM, NPL, NMW)                                                                                      \
    do {                                                                                                                 \
        constexpr size_t lds = lds_fwd<1, NPL, 2>();                                                                     \
        static bool attr = false;                                                                                        \
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_ws_kernel<MODE, OM, NPL, NMW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX); attr = true; } \
        const Sched sc = sched_for(s.ntiles, lds, 1, 1);                                                                 \
        hipLaunchKernelGGL((fwd_ws_kernel<MODE, OM, NPL, NMW>), dim3(sc.G), dim3((NMW + 4) * 64), lds, st, x, (const u16*)table, bias, \
                           in_scale, in_shift, mask_src, out_mask, y, dt, s, sc, alpha, g_dbg, g_dbg_buf, status_words()); \
    } while (0)
#define C2M_WS(MODE, OM, NPL) do { if (fwd_ws_waves() == 8) C2M_WSN(MODE, OM, NPL, 8); else C2M_WSN(MODE, OM, NPL, 4); } while (0)
#define C2M_L(DIL, MODE, OM, NPL)                                                                                        \
    do {                                                                                                                 \
        if (DIL == 1 && NPL == 3 && !(g_dbg & DBG_FOUR_WAVES) && (fwd_wave_specialised() == 1 || (fwd_wave_specialised() == 2 && !in_scale))) C2M_WS(MODE, OM, 3);                                         \
        else if (DIL == 1 && fwd_double_buffered()) { if (fwd_table_in_registers()) C2M_LB(1, MODE, OM, NPL, 2, true); else C2M_LB(1, MODE, OM, NPL, 2, false); } \
        else C2M_LB(DIL, MODE, OM, NPL, 1, false);                                                                       \
    } while (0)
#define C2M_M(DIL, NPL)                                                                                                  \
    do {                                                                                                                 \
        if (in_mode == PTTS_IN_LRELU) { if (om) C2M_L(DIL, PTTS_IN_LRELU, true, NPL); else C2M_L(DIL, PTTS_IN_LRELU, false, NPL); } \
        else if (in_mode == PTTS_IN_MASKMUL) { PTTS_REQUIRE(!om, "conv2d_mfma_fwd: MASKMUL with an output mask has no kernel"); C2M_L(DIL, PTTS_IN_MASKMUL, false, NPL); } \
        else { if (om) C2M_L(DIL, PTTS_IN_NONE, true, NPL); else C2M_L(DIL, PTTS_IN_NONE, false, NPL); }                 \
    } while (0)
    if (planes == 1) C2M_M(1, 1);
    else if (dil_t == 1) C2M_M(1, 3); else if (dil_t == 2) C2M_M(2, 3); else if (dil_t == 4) C2M_M(4, 3); else C2M_M(8, 3);
#undef C2M_M
#undef C2M_L
#undef C2M_WS
#undef C2M_WSN
    return check_launch("conv2d_mfma_fwd");
}

extern "C" size_t ptts_conv2d_mfma_wgrad_workspace_bytes(int B, int T) {
    (void)B; (void)T;
    return 4096 + (size_t)NCU * 3 * NPART * sizeof(float);
}

// per-workgroup partial sums of dW / dbias as rows [nblocks][npart] behind a 4096-byte head of `workspace` (the layout of
// ptts_conv2d_bwd_partials: ptts_conv2d_reduce_grouped adds them into the gradient buffers).  planes / x_bf16 / dy_bf16 as
// in ptts_conv2d_mfma_fwd (mask_src has x's type); the sums are fp32 either way.
extern "C" int ptts_conv2d_mfma_wgrad_partials(const void* dy, const void* x, const void* mask_src, void* workspace,
                                               size_t workspace_bytes, int* nblocks_out, int* npart_out, int B, int T, int F,
                                               int KT_, int dil_t, int pad_t, int in_mode, float alpha,
                                               int planes, int x_bf16, int dy_bf16, void* stream) {
    PTTS_REQUIRE(dy && x && workspace && nblocks_out && npart_out, "conv2d_mfma_wgrad: null pointer");
    PTTS_REQUIRE(B > 0 && T > 0 && F > 0 && KT_ == 5, "conv2d_mfma_wgrad: bad dims");
    PTTS_REQUIRE(dil_t == 1 || dil_t == 2 || dil_t == 4 || dil_t == 8, "conv2d_mfma_wgrad: time dilation %d has no kernel (1, 2, 4, 8)", dil_t);
    PTTS_REQUIRE(in_mode >= 0 && in_mode <= 2, "conv2d_mfma_wgrad: bad in_mode %d", in_mode);
    PTTS_REQUIRE(in_mode != PTTS_IN_MASKMUL || mask_src, "conv2d_mfma_wgrad: MASKMUL needs mask_src");
    PTTS_REQUIRE(pad_t >= 0 && pad_t <= (KT - 1) * dil_t, "conv2d_mfma_wgrad: bad pad_t %d", pad_t);
    PTTS_REQUIRE(alpha >= 0.f && alpha <= 1.f, "conv2d_mfma_wgrad: LeakyReLU slope %g outside [0, 1]", alpha);
    PTTS_REQUIRE((long long)(T + 64) * F * C < (1LL << 31), "conv2d_mfma_wgrad: utterance too large for 32-bit tile offsets");
    PTTS_REQUIRE(planes == 3 || (planes == 1 && dil_t == 1), "conv2d_mfma_wgrad: planes %d / dilation %d: bf16 arithmetic is built for dilation 1", planes, dil_t);
    PTTS_REQUIRE(planes == 1 || (!x_bf16 && !dy_bf16), "conv2d_mfma_wgrad: bf16 tensors need planes == 1");
    const Shape s = make_shape(B, T, F, pad_t);
    PTTS_REQUIRE(shape_ok(s), "conv2d_mfma_wgrad: too many tiles");
    const size_t need = ptts_conv2d_mfma_wgrad_workspace_bytes(B, T);
    if (workspace_bytes < need) { set_error("conv2d_mfma_wgrad: workspace %zu < %zu", workspace_bytes, need); return PTTS_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    float* parts = reinterpret_cast<float*>((char*)workspace + 4096);
    const int dt = (x_bf16 ? DT_IN : 0) | (dy_bf16 ? DT_DY : 0);
    int grid = 0;
#define C2M_L(DIL, MODE, NPL)                                                                                            \
    do {                                                                                                                 \
        constexpr size_t lds = lds_wgrad<DIL, NPL>();                                                                    \
        static_assert(lds <= LDS_MAX, "tile does not fit the LDS");                                                      \
        static bool attr = false;                                                                                        \
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<DIL, MODE, NPL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX); attr = true; } \
        const Sched sc = sched_for(s.ntiles, lds, DIL <= 2 ? 2 : 1, 2);                                                  \
        grid = sc.G;                                                                                                     \
        hipLaunchKernelGGL((wgrad_kernel<DIL, MODE, NPL>), dim3(grid), dim3(THREADS), lds, st, x, dy, mask_src, parts, dt, s, sc, alpha, g_dbg, g_dbg_buf); \
    } while (0)
#define C2M_M(DIL, NPL)                                                                                                  \
    do {                                                                                                                 \
        if (in_mode == PTTS_IN_LRELU) C2M_L(DIL, PTTS_IN_LRELU, NPL);                                                    \
        else if (in_mode == PTTS_IN_MASKMUL) C2M_L(DIL, PTTS_IN_MASKMUL, NPL);                                           \
        else C2M_L(DIL, PTTS_IN_NONE, NPL);                                                                              \
    } while (0)
    if (planes == 1) C2M_M(1, 1);
    else if (dil_t == 1) C2M_M(1, 3); else if (dil_t == 2) C2M_M(2, 3); else if (dil_t == 4) C2M_M(4, 3); else C2M_M(8, 3);
#undef C2M_M
#undef C2M_L
    *nblocks_out = grid;
    *npart_out = NPART;
    return check_launch("conv2d_mfma_wgrad");
}

// the fused backward kernel with its multiplying waves split by role (twelve waves: PTTS_C2M_BW_SPLIT=1) or not (eight waves)
static bool bwd_split() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PTTS_C2M_BW_SPLIT"); v = e ? atoi(e) : 0; }
    return v != 0;
}

extern "C" size_t ptts_conv2d_mfma_bwd_fused_workspace_bytes(int B, int T) {
    (void)B; (void)T;
    return 4096 + (size_t)NCU * NPART * sizeof(float);
}

// 1 when the fused backward launch exists for the shape: 5x5, 4 -> 4 channels, time dilation 1, fp32 maps (three planes)
extern "C" int ptts_conv2d_mfma_bwd_fused_supported(int F, int dil_t, int planes) {
    return (F >= 1 && dil_t == 1 && planes == 3) ? 1 : 0;
}

// Two passes of a 4 -> 4 5x5 layer that share a staged tile, as ONE launch (c2m::bwd_ws_kernel):
//   kind 1 (first-order backward):  p = dy, q = x (the layer's pre-activation input; a = lrelu(x)), table = the transposed table:
//           y = dx = conv^T(dy) . lrelu'(x);  partial rows of dW = corr(lrelu(x), dy) and dbias = sum dy
//   kind 2 (second-order sweep):    p = u with mask_src = x (a = u . lrelu'(x)), q = dy, table = the forward table:
//           y = cot_dy = conv(a);             partial rows of dW = corr(a, dy) (no bias gradient: the row's bias sums are zero)
// pad_t is the FORWARD layer's ('same': 2).  Partial rows as ptts_conv2d_mfma_wgrad_partials writes them (4096-byte head).
extern "C" int ptts_conv2d_mfma_bwd_fused(const void* p, const void* q, const void* mask_src, const void* table, void* y,
                                          void* workspace, size_t workspace_bytes, int* nblocks_out, int* npart_out,
                                          int B, int T, int F, int KT_, int pad_t, int kind, float alpha, void* stream) {
    if (int rc = check_status("conv2d_mfma_bwd_fused")) return rc;
    PTTS_REQUIRE(p && q && table && y && workspace && nblocks_out && npart_out, "conv2d_mfma_bwd_fused: null pointer");
    PTTS_REQUIRE(B > 0 && T > 0 && F > 0 && KT_ == 5, "conv2d_mfma_bwd_fused: bad dims B=%d T=%d F=%d KT=%d", B, T, F, KT_);
    PTTS_REQUIRE(kind == 1 || kind == 2, "conv2d_mfma_bwd_fused: kind %d (1 = first-order backward, 2 = second-order sweep)", kind);
    PTTS_REQUIRE(kind == 1 || mask_src, "conv2d_mfma_bwd_fused: the second-order sweep needs mask_src");
    PTTS_REQUIRE(pad_t == 2, "conv2d_mfma_bwd_fused: built for 'same' padding at dilation 1 (pad_t 2, got %d)", pad_t);
    PTTS_REQUIRE(alpha >= 0.f && alpha <= 1.f, "conv2d_mfma_bwd_fused: LeakyReLU slope %g outside [0, 1]", alpha);
    PTTS_REQUIRE((long long)(T + 64) * F * C < (1LL << 31), "conv2d_mfma_bwd_fused: utterance too large for 32-bit tile offsets");
    const Shape s = make_shape(B, T, F, pad_t);
    PTTS_REQUIRE(shape_ok(s), "conv2d_mfma_bwd_fused: too many tiles");
    const size_t need = ptts_conv2d_mfma_bwd_fused_workspace_bytes(B, T);
    if (workspace_bytes < need) { set_error("conv2d_mfma_bwd_fused: workspace %zu < %zu", workspace_bytes, need); return PTTS_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    float* parts = reinterpret_cast<float*>((char*)workspace + 4096);
    typedef Stage<4 * (GPB + 1) + 4, 16 + (KT - 1)> SP;
    int grid = 0;
#define C2M_BF(KIND)                                                                                                     \
    do {                                                                                                                 \
        typedef Stage<4 * (GPB + 1) + 4, KIND == 1 ? 16 : 16 + (KT - 1)> SQ;                                             \
        constexpr size_t lds = (size_t)2 * NP * (SP::PS + SQ::PS) * sizeof(u16) + 64 + (size_t)KT * NP * TKP * sizeof(u16);   \
        static_assert(lds <= LDS_MAX, "tile does not fit the LDS");                                                      \
        static_assert((size_t)(4 * KT * 2 * 4 * 64 + 256 * 4) * sizeof(float) <= lds, "reduction scratch");             \
        static bool attr = false;                                                                                        \
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_ws_kernel<KIND, NP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX); \
                     (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_ws_kernel<KIND, NP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX); attr = true; } \
        const Sched sc = sched_for(s.ntiles, lds, 1, 2);                                                                 \
        grid = sc.G;                                                                                                     \
        if (bwd_split()) hipLaunchKernelGGL((bwd_ws_kernel<KIND, NP, true>), dim3(grid), dim3(12 * 64), lds, st, p, q, mask_src, (const u16*)table, y, parts, s, sc, alpha, g_dbg, status_words()); \
        else hipLaunchKernelGGL((bwd_ws_kernel<KIND, NP, false>), dim3(grid), dim3(8 * 64), lds, st, p, q, mask_src, (const u16*)table, y, parts, s, sc, alpha, g_dbg, status_words()); \
    } while (0)
    if (kind == 1) C2M_BF(1); else C2M_BF(2);
#undef C2M_BF
    *nblocks_out = grid;
    *npart_out = NPART;
    return check_launch("conv2d_mfma_bwd_fused");
}

// Kind 1 for a layer whose input was lrelu(scale x + shift) -- the kl.BatchNormalization + kl.LeakyReLU in front of the generator's
// kl.Conv2D layers (networktts.py:122-126): q is the raw map x, y = dx = conv^T(dy) lrelu'(scale x + shift) scale, and every partial row
// carries, behind the 400 kernel-gradient sums and the 4 bias-gradient sums, the 4 + 4 sums that are the gradients of scale and shift
// (columns 404 .. 411: reduce them with a second descriptor of ptts_conv2d_reduce_grouped -- partials + 404 floats, nw = 4, cout = 4).
extern "C" int ptts_conv2d_mfma_bwd_fused_affine(const float* p, const float* q, const void* table, float* y, void* workspace, size_t workspace_bytes,
                                                 int* nblocks_out, int* npart_out, int B, int T, int F, int KT_, int pad_t, float alpha,
                                                 const float* q_scale, const float* q_shift, void* stream) {
    if (int rc = check_status("conv2d_mfma_bwd_fused_affine")) return rc;
    PTTS_REQUIRE(p && q && table && y && workspace && nblocks_out && npart_out && q_scale && q_shift, "conv2d_mfma_bwd_fused_affine: null pointer");
    PTTS_REQUIRE(B > 0 && T > 0 && F > 0 && KT_ == 5, "conv2d_mfma_bwd_fused_affine: bad dims B=%d T=%d F=%d KT=%d", B, T, F, KT_);
    PTTS_REQUIRE(pad_t == 2, "conv2d_mfma_bwd_fused_affine: built for 'same' padding at dilation 1 (pad_t 2, got %d)", pad_t);
    PTTS_REQUIRE(alpha >= 0.f && alpha <= 1.f, "conv2d_mfma_bwd_fused_affine: LeakyReLU slope %g outside [0, 1]", alpha);
    PTTS_REQUIRE((long long)(T + 64) * F * C < (1LL << 31), "conv2d_mfma_bwd_fused_affine: utterance too large for 32-bit tile offsets");
    PTTS_REQUIRE((((uintptr_t)q_scale | (uintptr_t)q_shift) & 15) == 0, "conv2d_mfma_bwd_fused_affine: scale / shift must be 16-byte aligned");
    const Shape s = make_shape(B, T, F, pad_t);
    PTTS_REQUIRE(shape_ok(s), "conv2d_mfma_bwd_fused_affine: too many tiles");
    const size_t need = ptts_conv2d_mfma_bwd_fused_workspace_bytes(B, T);
    if (workspace_bytes < need) { set_error("conv2d_mfma_bwd_fused_affine: workspace %zu < %zu", workspace_bytes, need); return PTTS_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    float* parts = reinterpret_cast<float*>((char*)workspace + 4096);
    typedef Stage<4 * (GPB + 1) + 4, 16 + (KT - 1)> SP;
    typedef Stage<4 * (GPB + 1) + 4, 16> SQ;
    constexpr size_t lds = (size_t)2 * NP * (SP::PS + SQ::PS) * sizeof(u16) + 64 + (size_t)KT * NP * TKP * sizeof(u16);
    static_assert(lds <= LDS_MAX, "tile does not fit the LDS");
    static_assert((size_t)(4 * KT * 2 * 4 * 64 + 256 * 4 + 8 * 256) * sizeof(float) <= lds, "reduction scratch");
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_ws_kernel<1, NP, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX); attr = true; }
    const Sched sc = sched_for(s.ntiles, lds, 1, 2);
    hipLaunchKernelGGL((bwd_ws_kernel<1, NP, false, true>), dim3(sc.G), dim3(8 * 64), lds, st, p, q, nullptr, (const u16*)table, y, parts, s, sc, alpha,
                       g_dbg, status_words(), q_scale, q_shift);
    *nblocks_out = sc.G;
    *npart_out = NPART;
    return check_launch("conv2d_mfma_bwd_fused_affine");
}

// The critic's whole Conv2D stack as ONE launch per pass: L x (5x5 Conv2D, 4 filters, bias, LeakyReLU) with the maps between
// the layers held in the LDS, bf16 storage / bf16 products / fp32 accumulation (BASELINE configs[2]).
//
// Role on the hot path: reference networks_critic.py:64-70 (the spectral branch of the critic: 1 -> 4 -> ... -> 4 channels over
// [B, T, F] spectra), its first-order backward, the backward-data pass that the gradient penalty differentiates
// (optimizertts_wgan.py:53-68, K.gradients of a gradient) and that pass's own backward (the second-order sweep).  The layer-wise
// kernels of conv2d_mfma.hip spend 71 launches per critic step on it and move every map between HBM and the LDS two to three
// times; with bf16 maps the matrix-pipe work of a layer is 1.2 us, so the launches' prologues and the map traffic are all that is
// left (0.106 of the HBM roofline).  Here a workgroup owns a tile of TR = 32 time rows of one utterance over the
// whole frequency axis (F <= 68: one block of 17 bin groups), carries it through ALL layers with a halo of two rows per layer and
// side (recomputed: the products are cheap, the bytes are not), and touches HBM only for what the algorithm must keep:
//     forward        reads the spectrum, writes the L activation maps a_l = lrelu(z_l) (the backward's masks and operands);
//     backward       reads dL/da_L and the maps, keeps the gradient maps in the LDS, accumulates dW / db of all layers in
//                    registers (a wave per kernel row, 8 accumulators per layer) -- writes per-workgroup partial sums only;
//     backward-data  (gradient penalty, generator step) the same chain without weight gradients, writes d/dx and -- for the
//                    second-order sweep -- the masked gradient maps gamma_l;
//     second order   reads d/d(G_0), gamma_l and the maps, writes d/d(dL/da_L) and the partial sums of dW.
// Stored maps are POST-activation (a_l, not z_l): LeakyReLU with slope > 0 keeps the sign, so a_l serves as the next layer's
// operand, as the weight gradient's operand and as the mask source, and no consumer has any vector work to do on it.
//
// Arithmetic per layer (the Toeplitz arrangement of conv2d_mfma.hip, one bf16 plane): per kernel row kt the 5 x 4 taps form a banded
// block A[(so, co)][(j, ci)] = w[kt][j - so][ci][co], M = 4 output bins x 4 co, K = 8 input bins x 4 ci; the activations are the B
// operand with N = 16 time rows; v_mfma_f32_16x16x32_bf16, fp32 accumulation.  The first layer (1 -> 4) and the last backward-data
// step (4 -> 1) run through the same instruction with the missing channels zero.  Weight gradients: K = 16 rows x 2 adjacent bin
// groups, both operands read transposed out of the row-major LDS tiles by ds_read_b64_tr_b16.
//
// LDS tile: rows of 37 sixteen-byte units (bins -2 .. 71, four bf16 channels each), the unit order of conv2d_mfma.hip (two low
// bits of the unit index swapped, lanes 4..11 of an MFMA column own the even rows): conflict-free ds_read_b128 fragments.
// Internal maps lie in HBM as [B][T][FP][4] bf16 with FP = F rounded up to even and the pad bin zero, so that a row is a whole
// number of 16-byte units; the last map a_L is written unpadded ([B][T][F][4]): the dense layers read it as [B*T, 4F].
#include "common.h"
#include <algorithm>
#include <cstdlib>

namespace ptts {
namespace c2c {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

constexpr int C = 4, KT = 5, KF = 5;
constexpr int LMAX = 8;            // layers per chain
constexpr int TR = 32;             // a tile's own time rows
constexpr int NGMAX = 17;          // bin groups of 4: F <= 68
constexpr int RSU = 37;            // 16-byte units per LDS row (odd): staged bins -2 .. 71
constexpr int RS = RSU * 8;        // bf16 elements per LDS row
constexpr int TSLOTS = 11, TROW = TSLOTS * C, TKP = C * TROW, TLAY = KT * TKP;    // operand table of one layer and direction
constexpr int NPART = KT * KF * 16 + 4 + 8;     // row of partial sums (the layout of ptts_conv2d_reduce_grouped)
constexpr int NCU = 256;
constexpr size_t LDS_MAX = 160 * 1024;
// LDS header: tables [LMAX][TLAY] bf16 | biases [LMAX][4] f32 | 64 zero bytes | per-wave bias-gradient sums
constexpr int HDR_BIAS = LMAX * TLAY * 2, HDR_ZERO = HDR_BIAS + LMAX * 16, HDR_BS = 14336, HDR_BYTES = 16384;
static_assert(HDR_ZERO + 64 <= HDR_BS, "header layout");
// device table buffer: [LMAX][2 directions][TLAY] bf16, then [LMAX][4] fp32 biases
constexpr size_t TAB_BIAS_OFF = (size_t)LMAX * 2 * TLAY * sizeof(u16);
constexpr size_t TAB_ZERO_OFF = TAB_BIAS_OFF + LMAX * 4 * sizeof(float);          // 64 zero bytes: the source of halo units in the map DMAs
constexpr size_t TAB_BYTES = TAB_ZERO_OFF + 64;

__host__ __device__ constexpr int unit_pos(int u) { return (u & ~3) | ((u & 1) << 1) | ((u >> 1) & 1); }
__host__ __device__ constexpr int bin_off(int c) { return unit_pos(c >> 1) * 8 + (c & 1) * 4; }      // elements, staged bin c = f + 2
__device__ __forceinline__ int row_of_lane(int li) { return (li >= 4 && li < 12) ? 2 * (li - 4) : (li < 4 ? 2 * li + 1 : 2 * li - 15); }
__device__ __forceinline__ bf16x8 cat(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ float max_fast(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef s16x4 __attribute__((address_space(3)))* lds_s16x4_ptr;
__device__ __forceinline__ bf16x4 tr_read(const u16* p) {
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p)));
}
__device__ __forceinline__ bf16x4 to_bf16(f32x4 v) { return __builtin_convertvector(v, bf16x4); }
__device__ __forceinline__ f32x4 to_f32(bf16x4 v) { return __builtin_convertvector(v, f32x4); }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
// workgroup barrier that orders the LDS only: __syncthreads() also waits for every global store of the wave to be acknowledged
// (vmcnt counts stores on gfx950) -- the maps a step writes to HBM would be drained at every layer boundary
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// measurement hook (tools/chain_probe.py): 100 MHz time stamps of the phases of every workgroup's first tile, dbg[block][32]
__device__ __forceinline__ void stamp(unsigned long long* dbg, int slot) {
    if (dbg && threadIdx.x == 0) dbg[(size_t)blockIdx.x * 32 + slot] = __builtin_amdgcn_s_memrealtime();
}
// ... and of the parts of one step (the second) as waves 0 (weight-gradient wave) and 7 see them: slots 16 + 8 (wave == 7) + i
__device__ __forceinline__ void stamp_w(unsigned long long* dbg, int s, int wave, int lane, int i) {
    if (dbg && s == 1 && lane == 0 && (wave == 0 || wave == 7)) dbg[(size_t)blockIdx.x * 32 + 16 + (wave == 7 ? 8 : 0) + i] = __builtin_amdgcn_s_memrealtime();
}
static unsigned long long* g_dbg = nullptr;

// ------------------------------------------------------------------------------------------------------------
// operand tables of all layers, both directions (layout of conv2d_mfma.hip with one plane):
//   tab[l][dir][kt][oc][kf + 3 (11 slots)][ic]   dir 0: A[(so,co)][(j,ci)] = w[kt][j - so][ci][co]
//                                                 dir 1: A[(so,ci)][(j,co)] = w[KT-1-kt][KF-1-(j - so)][ci][co]
// a layer with fewer than four input channels (the first one) has the missing channels zero.
// ------------------------------------------------------------------------------------------------------------
struct TabArgs { const float* w[LMAX]; const float* b[LMAX]; int cin[LMAX]; };
__global__ void chain_tables_kernel(TabArgs a, u16* __restrict__ tab, float* __restrict__ bias) {
    const int l = blockIdx.x >> 1, transposed = blockIdx.x & 1;
    const float* __restrict__ w = a.w[l];
    const int Cin = a.cin[l];
    const int idx = threadIdx.x;
    if (idx < KT * C * TSLOTS) {
        const int slot = idx % TSLOTS, oc = (idx / TSLOTS) % C, kt = idx / (TSLOTS * C);
        const int kf = slot - 3;
        f32x4 v = zero4();
        if (kf >= 0 && kf < KF) {
#pragma unroll
            for (int ic = 0; ic < C; ++ic) {
                if (!transposed) { if (ic < Cin) v[ic] = w[((kt * KF + kf) * Cin + ic) * C + oc]; }
                else { if (oc < Cin) v[ic] = w[(((KT - 1 - kt) * KF + (KF - 1 - kf)) * Cin + oc) * C + ic]; }
            }
        }
        *reinterpret_cast<bf16x4*>(tab + (size_t)(l * 2 + transposed) * TLAY + (kt * C + oc) * TROW + slot * C) = to_bf16(v);
    }
    if (!transposed && idx < C) bias[l * C + idx] = a.b[l] ? a.b[l][idx] : 0.f;
    if (blockIdx.x == 0 && idx < 16) reinterpret_cast<unsigned*>(bias + LMAX * C)[idx] = 0u;       // the zero block
}

// ------------------------------------------------------------------------------------------------------------
// geometry of a launch
// ------------------------------------------------------------------------------------------------------------
struct Geo {
    int B, T, F, FP, L, ng;            // FP: padded bins of the internal maps; ng: bin groups
    int ntt, ntiles;                   // time tiles per utterance, all tiles
    unsigned magic_fp2;                // ceil(2^32 / (FP / 2))
    float alpha;
    long long map_stride;              // elements between the internal maps of consecutive layers: B * T * FP * 4
};

// ------------------------------------------------------------------------------------------------------------
// tile loaders (global -> registers -> LDS; the loads of the next step's tile are in flight while this step multiplies)
// ------------------------------------------------------------------------------------------------------------
// (a) a padded bf16 map: whole 16-byte units, global -> LDS by DMA (global_load_lds_dwordx4: every lane names its own 16
// source bytes, a wave-instruction fills 64 consecutive units of the tile's unit image; no staging registers, no ds_write --
// the register-staged form held 20 VGPRs across a step, which the accumulator kernel spilled to scratch: the "prefetch" was
// waited for right after its issue).  Slot k of a thread: unit image index tid + NT k -> (row r, position p); units without
// data (the row halo, rows outside the utterance) read a zero block.  The caller waits (dma_wait) before the barrier that
// publishes the tile.
typedef void __attribute__((address_space(3)))* lds_void_ptr;
__device__ __forceinline__ void dma16(const void* src, const u16* lds_wave_base) {
    const unsigned lds_off = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long)(lds_void_ptr)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_off) : "memory", "m0");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int NS, int NT>
struct MapDma {
    int pk[NS];        // tile-independent: row << 16 | element offset of the unit inside a map row, or -1 (no data)
    __device__ __forceinline__ void init(const Geo& g) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int idx = threadIdx.x + k * NT;
            const int r = (int)__umulhi((unsigned)idx, 116080198u /* ceil(2^32 / 37) */), p = idx - r * RSU;
            const int u = unit_pos(p);
            pk[k] = (r << 16) | ((u >= 1 && 2 * u <= g.FP) ? (2 * u - 2) * C : 0xffff);
        }
    }
    // rows [t_org, t_org + nrows) of utterance b -> tile (row 0 <-> t_org)
    __device__ __forceinline__ void issue(const u16* __restrict__ map, const Geo& g, int b, int t_org, int nrows, const u16* tile, const u16* zeros) const {
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const u16* base = map + (long long)(b * g.T + t_org) * g.FP * C;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if ((k * NT + wave * 64) >= nrows * RSU) break;                  // wave-uniform: nothing of this instruction is needed
            const int r = pk[k] >> 16, uo = pk[k] & 0xffff, t = t_org + r;
            const u16* src = (uo != 0xffff && (unsigned)t < (unsigned)g.T) ? base + (long long)r * g.FP * C + uo : zeros;
            if (r < nrows) dma16(src, tile + (size_t)(k * NT + wave * 64) * 8);
        }
    }
};
// (b) a one-channel fp32 image [B][T][ld] (the spectrum, d/dG_0): pixel slots over (row, staged bin 0..71); channels 1..3 zero
template <int NS, int NT>
struct X0Pref {
    float v[NS];
    __device__ __forceinline__ void load(const float* __restrict__ x, long long ld, const Geo& g, int b, int t_org, int nrows) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int idx = threadIdx.x + k * NT;
            const int r = (int)__umulhi((unsigned)idx, 59652324u /* ceil(2^32 / 72) */), c = idx - r * 72;
            const int t = t_org + r, f = c - 2;
            v[k] = 0.f;
            if (r < nrows && (unsigned)t < (unsigned)g.T && (unsigned)f < (unsigned)g.F) v[k] = x[(long long)(b * g.T + t) * ld + f];
        }
    }
    __device__ __forceinline__ void commit(u16* tile, int nrows) const {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int idx = threadIdx.x + k * NT;
            const int r = (int)__umulhi((unsigned)idx, 59652324u), c = idx - r * 72;
            if (r < nrows) *reinterpret_cast<bf16x4*>(tile + r * RS + bin_off(c)) = to_bf16(f32x4{v[k], 0.f, 0.f, 0.f});
        }
    }
};

// own rows of an LDS tile -> a padded map in HBM: whole rows of FP / 2 units (the pad bin is zero in the tile).  The slots of a
// thread (tile-independent) are worked out once per kernel; all reads are issued before the first store.
template <int NT>
struct RowStore {
    static constexpr int NS = (TR * (4 * NGMAX / 2) + NT - 1) / NT;
    int lo_[NS];       // LDS element offset relative to the own rows' first row
    int go_[NS];       // element offset in the map relative to the own rows' first row, or -1
    int r_[NS];
    __device__ __forceinline__ void init(const Geo& g) {
        const int upr = g.FP >> 1, total = TR * upr;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int idx = threadIdx.x + k * NT;
            const int r = upr > 1 ? (int)__umulhi((unsigned)idx, g.magic_fp2) : idx, u = idx - r * upr + 1;
            r_[k] = idx < total ? r : (1 << 20);
            lo_[k] = r * RS + unit_pos(u) * 8;
            go_[k] = (r * g.FP + (2 * u - 2)) * C;
        }
    }
    __device__ __forceinline__ void run(const u16* tile_own, u16* __restrict__ map_own, int nrows) const {
        bf16x8 v[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) if (r_[k] < nrows) v[k] = *reinterpret_cast<const bf16x8*>(tile_own + lo_[k]);
#pragma unroll
        for (int k = 0; k < NS; ++k) if (r_[k] < nrows) *reinterpret_cast<bf16x8*>(map_own + go_[k]) = v[k];
    }
};
// ... -> the unpadded last map [B][T][F][4]: eight bytes per lane, consecutive lanes consecutive pixels
template <int NT>
__device__ __forceinline__ void store_rows_plain(const u16* tile, int tile_torg, u16* __restrict__ map, const Geo& g, int b, int t0) {
    const int nrows = min(TR, g.T - t0), total = nrows * g.F;
    const unsigned magic_f = (unsigned)(((1ULL << 32) + (unsigned)g.F - 1) / (unsigned)g.F);
    u16* dst = map + (long long)(b * g.T + t0) * g.F * C;
    for (int idx = threadIdx.x; idx < total; idx += NT) {
        const int r = g.F > 1 ? (int)__umulhi((unsigned)idx, magic_f) : idx, f = idx - r * g.F;
        *reinterpret_cast<bf16x4*>(dst + (size_t)idx * C) = *reinterpret_cast<const bf16x4*>(tile + (t0 + r - tile_torg) * RS + bin_off(f + 2));
    }
}

// ------------------------------------------------------------------------------------------------------------
// one layer on LDS tiles: out[t][f][:] = acc0 + sum_kt A_kt . in[t - 2 + kt][f - 2 .. f + 5][:] for t in [ta, tb), all bin groups:
// five MFMAs per (chunk of 16 rows, bin group).  The last chunk ends at tb: it overlaps its predecessor when tb - ta is no
// multiple of 16 (same values written twice); `fresh` tells the epilogue which of its rows are new.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_afrag(const u16* tabl, int lane, bf16x8 (&a)[KT]) {
    const int li = lane & 15, lg = lane >> 4;
    const u16* wa = tabl + (li & 3) * TROW + (2 * lg - (li >> 2) + 3) * C;      // row oc, slot of tap 2 lg - so
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
        a[kt] = cat(*reinterpret_cast<const bf16x4*>(wa + kt * TKP), *reinterpret_cast<const bf16x4*>(wa + kt * TKP + 4));
}

// per-lane element offsets inside a 16-row chunk, relative to the chunk's first row and to bin-group pair 0
struct LaneOff {
    int rdA, rdB;      // B-operand fragment of the even / odd group of a pair: row of the lane, unit 2 g + lg
    int pxA, pxB;      // the lane's output pixel: row of the lane, staged bin 4 g + lg + 2
    int rl, lg;
    __device__ __forceinline__ void init(int lane) {
        const int li = lane & 15;
        lg = lane >> 4; rl = row_of_lane(li);
        rdA = rl * RS + unit_pos(lg) * 8; rdB = rl * RS + unit_pos(2 + lg) * 8;
        pxA = rl * RS + bin_off(lg + 2); pxB = rl * RS + bin_off(lg + 6);
    }
};

// Work of a step = (chunk of 16 rows, PAIR of bin groups): the even and the odd group of a pair differ in their unit positions
// only by per-lane constants (LaneOff); a wave owns fixed pairs for all chunks of a step, so everything else of an address is a
// scalar and the loop carries no per-item index arithmetic (the first version spent 16 instructions per MFMA: rocprofv3
// SQ_INSTS_VALU / SQ_INSTS_SALU, gpurun_out/chain_pmc).
// pair_of: the bin-group pair of slot j (0, 1) of this wave at chunk c, or -1.
//   all waves alike (WG = false):   pairs 0 .. NW-1 fixed on slot 0, pair NW (the ninth of a 17-group row) goes round the waves
//   waves 0 .. KT-1 also run weight-gradient products (WG = true): waves KT .. NW-1 own pairs 0 .. 2 (NW-KT) - 1 (two each),
//                                   the remaining three pairs go round waves 0 .. KT-1
template <int NW, bool WG>
__device__ __forceinline__ int pair_of(int j, int c, int wave, int ng2, int rot) {
    int gp;
    if (!WG) gp = j == 0 ? wave : ((((c + rot) & (NW - 1)) == wave) ? NW : -1);
    else if (wave >= KT) gp = (wave - KT) + j * (NW - KT);
    else {
        if (j) return -1;
        // (round 4: shifting work from the convolution-only waves to these -- five pairs on waves KT.., four going round here, as the
        // per-wave stamps of tools/chain_probe.py suggested -- made the step LONGER, 329 -> 339 us backward, 336 -> 371 second order at
        // 2B: the waves are not independent, the step is bound by what the CU issues in all, not by its longest wave)
        const int k = (c + rot + wave) % KT;
        gp = k < 3 ? 2 * (NW - KT) + k : -1;
    }
    return gp < ng2 ? gp : -1;
}

// SEQ: the two groups of a pair one after the other (the kernels that carry the weight-gradient accumulators: register budget),
// else both groups' ten fragment reads are in flight together.  More than eight waves: wave & 7 owns the pairs, wave >> 3 the
// chunks c = wave >> 3, + NW / 8, ...
//   pre(tc, t, f, px)                      -> what the epilogue wants fetched before the MFMAs (a mask source); px = the
//                                             pixel's element offset relative to tile row tc
//   epi(tc, t, f, px, acc, pre, ok, fresh)    ok: the pixel lies inside the image; fresh: not produced by the previous chunk
// (A hand-pipelined stream of single units -- reads of unit i + 1 before the MFMAs of unit i -- was built and measured 1.7 x
// SLOWER: a wave issues one instruction per four cycles, and the scalar bookkeeping of the stream cost more than the LDS latency
// it hid.  What counts here is the number of instructions per wave and step.)
template <int NW, bool WG, bool SEQ, bool DUAL = false, class Pre, class Epi>
__device__ __forceinline__ void conv_chunks(const u16* in, int in_torg, const bf16x8 (&a)[KT], int ta, int tb, const Geo& g,
                                            int wave_, int rot, const LaneOff& lo, f32x4 acc0, Pre&& pre, Epi&& epi) {
    constexpr int NWP = NW > 8 ? 8 : NW, CST = NW / NWP;
    const int wave = wave_ & (NWP - 1), c0 = wave_ / NWP;
    const int ng2 = (g.ng + 1) >> 1, nch = (tb - ta + 15) >> 4;
    for (int c = c0; c < nch; c += CST) {
        const int tc = min(ta + 16 * c, tb - 16);
        const int t = tc + lo.rl;
        const bool tin = (unsigned)t < (unsigned)g.T, fresh = t >= ta + 16 * c;
        if (DUAL) {
            // a wave that owns TWO full pairs at this chunk (the waves without weight-gradient work): all twenty fragment reads
            // and the four mask fetches first, then four interleaved MFMA chains -- one exposed LDS latency per chunk instead of
            // two, and the epilogue of one group under the MFMAs of the next
            const int g0 = pair_of<NWP, WG>(0, c, wave, ng2, rot), g1 = pair_of<NWP, WG>(1, c, wave, ng2, rot);
            if (g0 >= 0 && g1 >= 0 && 2 * g0 + 1 < g.ng && 2 * g1 + 1 < g.ng) {
                const u16* bp0 = in + (tc - 2 - in_torg) * RS + 32 * g0;
                const u16* bp1 = in + (tc - 2 - in_torg) * RS + 32 * g1;
                const int f0 = 8 * g0 + lo.lg, f1 = 8 * g1 + lo.lg;
                const int px0 = lo.pxA + 32 * g0, px1 = lo.pxB + 32 * g0, px2 = lo.pxA + 32 * g1, px3 = lo.pxB + 32 * g1;
                auto p0 = pre(tc, t, f0, px0); auto p1 = pre(tc, t, f0 + 4, px1);
                auto p2 = pre(tc, t, f1, px2); auto p3 = pre(tc, t, f1 + 4, px3);
                bf16x8 b0[KT], b1[KT], b2[KT], b3[KT];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    b0[kt] = *reinterpret_cast<const bf16x8*>(bp0 + lo.rdA + kt * RS);
                    b1[kt] = *reinterpret_cast<const bf16x8*>(bp0 + lo.rdB + kt * RS);
                }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    b2[kt] = *reinterpret_cast<const bf16x8*>(bp1 + lo.rdA + kt * RS);
                    b3[kt] = *reinterpret_cast<const bf16x8*>(bp1 + lo.rdB + kt * RS);
                }
                f32x4 x0 = acc0, x1 = acc0, x2 = acc0, x3 = acc0;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    x0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kt], b0[kt], x0, 0, 0, 0);
                    x1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kt], b1[kt], x1, 0, 0, 0);
                }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    x2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kt], b2[kt], x2, 0, 0, 0);
                    x3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kt], b3[kt], x3, 0, 0, 0);
                }
                epi(tc, t, f0, px0, x0, p0, tin && f0 < g.F, fresh);
                epi(tc, t, f0 + 4, px1, x1, p1, tin && f0 + 4 < g.F, fresh);
                epi(tc, t, f1, px2, x2, p2, tin && f1 < g.F, fresh);
                epi(tc, t, f1 + 4, px3, x3, p3, tin && f1 + 4 < g.F, fresh);
                continue;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gp = pair_of<NWP, WG>(j, c, wave, ng2, rot);
            if (gp < 0) continue;                                              // wave-uniform
            const bool hasB = 2 * gp + 1 < g.ng;
            const u16* bp = in + (tc - 2 - in_torg) * RS + 32 * gp;
            const int fA = 8 * gp + lo.lg, fB = fA + 4;
            const int pxA = lo.pxA + 32 * gp, pxB = lo.pxB + 32 * gp;
            const bool okA = tin && fA < g.F, okB = tin && fB < g.F;
            auto pA = pre(tc, t, fA, pxA);
            bf16x8 bA[KT], bB[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) bA[kt] = *reinterpret_cast<const bf16x8*>(bp + lo.rdA + kt * RS);
            if (!SEQ && hasB) {
                auto pB = pre(tc, t, fB, pxB);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) bB[kt] = *reinterpret_cast<const bf16x8*>(bp + lo.rdB + kt * RS);
                f32x4 xA = acc0, xB = acc0;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    xA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kt], bA[kt], xA, 0, 0, 0);
                    xB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kt], bB[kt], xB, 0, 0, 0);
                }
                epi(tc, t, fA, pxA, xA, pA, okA, fresh);
                epi(tc, t, fB, pxB, xB, pB, okB, fresh);
            } else {
                f32x4 xA = acc0;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) xA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kt], bA[kt], xA, 0, 0, 0);
                epi(tc, t, fA, pxA, xA, pA, okA, fresh);
                if (hasB) {
                    auto pB = pre(tc, t, fB, pxB);
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) bB[kt] = *reinterpret_cast<const bf16x8*>(bp + lo.rdB + kt * RS);
                    f32x4 xB = acc0;
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) xB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kt], bB[kt], xB, 0, 0, 0);
                    epi(tc, t, fB, pxB, xB, pB, okB, fresh);
                }
            }
        }
    }
}

// weight-gradient products of one layer for this wave's kernel row kt, own rows of the tile:
//   C_hb[(fi, ci)][(fo, co)] += sum_{t, g} a[t + kt - 2][4 g + fi][ci] * d[t][4 (g + hb) + fo - 2][co],   kf = fi + 4 - 4 hb - fo
// `at` / `dt` point at the tile rows of time t0 + kt - 2 / t0.  K step = 16 rows x 2 adjacent bin groups; with an odd number of
// groups the last pair's second group does not exist: its fragments are read from the zero block.
__device__ __forceinline__ void dw_step(f32x4& c0, f32x4& c1, const u16* at, const u16* dt, int ng, const u16* zero, int lane) {
    const int li = lane & 15, lg = lane >> 4;
    const int trow = 4 * lg + (li >> 2), fb = li & 3;
    const int ng2 = (ng + 1) >> 1, nk = (TR / 16) * ng2;
    const bool odd = (ng & 1) != 0;
    const u16* ab = at + trow * RS;
    const u16* db = dt + trow * RS;
    const int oa0 = bin_off(2 + fb), oa1 = bin_off(6 + fb), od0 = bin_off(fb), od1 = bin_off(4 + fb), od2 = bin_off(8 + fb);
    // K step k = (row chunk rc, bin group pair gp); the five fragment reads of step k + 1 are issued before the MFMAs of step k
    auto rd = [&](int k, bf16x4 (&v)[5]) {
        const int rc = k >= ng2 ? 1 : 0, gp = k - rc * ng2;              // TR / 16 == 2
        const bool ph = odd && gp == ng2 - 1;                            // wave-uniform
        const int o = rc * 16 * RS + 32 * gp;
        v[0] = tr_read(ab + o + oa0);
        v[1] = tr_read(ph ? zero : ab + o + oa1);
        v[2] = tr_read(db + o + od0);
        v[3] = tr_read(db + o + od1);
        v[4] = tr_read(ph ? zero : db + o + od2);
    };
    auto mm = [&](const bf16x4 (&v)[5]) {
        const bf16x8 af = cat(v[0], v[1]);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, cat(v[2], v[3]), c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, cat(v[3], v[4]), c1, 0, 0, 0);
    };
    static_assert(TR == 32, "dw_step: two row chunks");
    bf16x4 va[5], vb[5];
    rd(0, va);
    for (int k = 0; k < nk; k += 2) {          // nk is even
        rd(k + 1, vb);
        mm(va);
        if (k + 2 < nk) rd(k + 2, va);
        mm(vb);
    }
}

// acc[s] += (c0, c1) for a run-time step index: the accumulators keep static register names behind a wave-uniform switch, so
// the step loop itself need not be unrolled (unrolled eight-fold it drove the kernel past its register budget)
__device__ __forceinline__ void acc_add(f32x4 (&acc)[LMAX][2], int s, f32x4 c0, f32x4 c1) {
    switch (s) {
// (the empty asm pins the two registers inside their own case: without it the optimiser sinks the eight identical additions out of
// the switch into ONE addition through a run-time index -- half of the accumulators then live in scratch memory, one 16-byte scratch
// load + store per step and lane: 144 bytes of scratch per lane, 88-141 MB of scratch write-backs per launch, profiles/r03_*)
#define C2C_CASE(S) case S: acc[S][0] += c0; acc[S][1] += c1; asm volatile("" : "+v"(acc[S][0]), "+v"(acc[S][1])); break;
        C2C_CASE(0) C2C_CASE(1) C2C_CASE(2) C2C_CASE(3) C2C_CASE(4) C2C_CASE(5) C2C_CASE(6) C2C_CASE(7)
#undef C2C_CASE
        default: break;
    }
}
static_assert(LMAX == 8, "acc_add: eight cases");

// dst[0..3] += the sums of v over the wave, by lane 0.  Data-parallel-primitive rotations inside the rows of 16 lanes and four
// v_readlane: no LDS round trips (the butterfly of __shfl_xor -- six dependent ds_bpermute per component -- cost 1 us per step).
__device__ __forceinline__ float row_sum16(float v) {
    // after rotations by 8, 4, 2, 1 every lane of a row holds the row's sum
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}
__device__ __forceinline__ void wave_add4(float* dst, f32x4 v, int lane) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float r = row_sum16(v[e]);
        const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 0)) +
                        __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 16)) +
                        __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 32)) +
                        __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 48));
        if (lane == 0) dst[e] += s;
    }
}

// LDS set-up shared by the kernels: zero everything, then the tables of one direction (and the biases)
template <int NT>
__device__ __forceinline__ void lds_init(u16* lds, size_t lds_bytes, const u16* __restrict__ tab, const float* __restrict__ bias, int L, int dir) {
    for (size_t i = threadIdx.x; i < lds_bytes / 16; i += NT) reinterpret_cast<u32x4*>(lds)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    for (int i = threadIdx.x; i < L * (TLAY / 8); i += NT) {
        const int l = i / (TLAY / 8), j = i - l * (TLAY / 8);
        reinterpret_cast<bf16x8*>(lds)[i] = *reinterpret_cast<const bf16x8*>(tab + (size_t)(l * 2 + dir) * TLAY + j * 8);
    }
    if (bias && (int)threadIdx.x < L * C) reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + HDR_BIAS)[threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------------------
// forward: x0 [B][T][ldx] fp32 (one channel) -> a_1 .. a_{L-1} (padded maps), a_L (plain)
// tiles P0, P1 of TR + 4 L rows, row i <-> t = t0 - 2 L + i; a_l is valid on [t0 - 2 (L - l), t0 + TR + 2 (L - l))
// ------------------------------------------------------------------------------------------------------------
struct FwdArgs {
    const float* x0; long long ldx;
    const u16* tab; const float* bias;
    u16* maps; u16* a_last;
    Geo g; size_t lds_bytes;
    unsigned long long* dbg;
};

template <int NT>
__global__ __launch_bounds__(NT) void chain_fwd_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) u16 lds[];
    constexpr int NW = NT / 64;
    const Geo g = a.g;
    const int L = g.L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    stamp(a.dbg, 0);
    lds_init<NT>(lds, a.lds_bytes, a.tab, a.bias, L, 0);
    stamp(a.dbg, 1);
    LaneOff lo;
    lo.init(lane);
    RowStore<NT> rs;
    rs.init(g);
    const float* biasl = reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds) + HDR_BIAS);
    const int RB = TR + 4 * L;
    u16* P0 = lds + HDR_BYTES / 2;
    u16* P1 = P0 + RB * RS;
    constexpr int NSX = ((TR + 4 * LMAX) * 72 + NT - 1) / NT;
    X0Pref<NSX, NT> xp;
    int tile = blockIdx.x;
    if (tile < g.ntiles) {
        const int b = tile / g.ntt, t0 = (tile - b * g.ntt) * TR;
        xp.load(a.x0, a.ldx, g, b, t0 - 2 * L, RB);
    }
    for (; tile < g.ntiles; tile += gridDim.x) {
        const int b = tile / g.ntt, t0 = (tile - b * g.ntt) * TR;
        const int torg = t0 - 2 * L;
        xp.commit(P0, RB);
        const int nxt = tile + gridDim.x;
        if (nxt < g.ntiles) {
            const int nb = nxt / g.ntt, nt0 = (nxt - nb * g.ntt) * TR;
            xp.load(a.x0, a.ldx, g, nb, nt0 - 2 * L, RB);
        }
        lds_barrier();
        stamp(a.dbg, 2);
        for (int l = 1; l <= L; ++l) {
            const u16* in = (l & 1) ? P0 : P1;
            u16* out = (l & 1) ? P1 : P0;
            bf16x8 af[KT];
            load_afrag(lds + (l - 1) * TLAY, lane, af);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(biasl + (l - 1) * C);
            if (l >= 2) rs.run(in + (t0 - torg) * RS, a.maps + (long long)(l - 2) * g.map_stride + (long long)(b * g.T + t0) * g.FP * C, min(TR, g.T - t0));
            const int ext = 2 * (L - l);
            const int ta = t0 - ext, tb = t0 + TR + ext;
            conv_chunks<NW, false, false>(in, torg, af, ta, tb, g, wave, l, lo, bv,
                       [](int, int, int, int) { return 0; },
                       [&](int tc, int, int, int px, f32x4 v, int, bool ok, bool) {
#pragma unroll
                           for (int e = 0; e < 4; ++e) v[e] = max_fast(v[e], g.alpha * v[e]);       // LeakyReLU, 0 <= alpha <= 1
                           if (!ok) v = zero4();                                                    // zero padding of the next layer
                           *reinterpret_cast<bf16x4*>(out + (tc - torg) * RS + px) = to_bf16(v);
                       });
            stamp(a.dbg, 2 + l);
            lds_barrier();
        }
        store_rows_plain<NT>((L & 1) ? P1 : P0, torg, a.a_last, g, b, t0);
        lds_barrier();
        stamp(a.dbg, 12);
        a.dbg = nullptr;           // stamps: the first tile of a workgroup only
    }
}

// ------------------------------------------------------------------------------------------------------------
// backward chains.  d_l = lrelu'(a_l) . dL/da_l lives in tiles D0 / D1 (row i <-> t0 - EXT + i), the maps a_{l-1} arrive in tiles
// A0 / A1 one step ahead.  MODE_BWD: weight gradients, d_l valid on [t0 - 2 (l - 1), ...); MODE_DATA: no weight gradients, d_l valid on
// [t0 - 2 l, ...), gamma_l = d_l stored (optional), last step d/dx0 = W_1^T * d_1 (channel 0) -> g0 [B][T][F] fp32.
// ------------------------------------------------------------------------------------------------------------
struct BwdArgs {
    const void* d_last; int d_bf16;            // dL/da_L [B][T][F][4], fp32 or bf16
    const float* x0; long long ldx;            // the stack's input (weight gradient of the first layer); MODE_BWD only
    const u16* maps; const u16* a_last;
    const u16* tab;
    u16* gmaps;                                // MODE_DATA: gamma_1 .. gamma_L (padded maps) or NULL
    float* g0;                                 // MODE_DATA: d/dx0 [B][T][F] or NULL
    float* partials;                           // MODE_BWD: [L][gridDim][NPART]
    int cin0;                                  // input channels of the first layer (1)
    Geo g; size_t lds_bytes;
    unsigned long long* dbg;
};

constexpr int MODE_BWD = 0, MODE_DATA = 1;

// the first stage: d_L = lrelu'(a_L) . dL/da_L on rows [ta, tb) -> tile D (zero outside the image); db_L from the own rows
template <int NT, bool WANT_B>
__device__ __forceinline__ void stage_d_last(const BwdArgs& a, const Geo& g, u16* D, int d_torg, int b, int ta, int tb, int t0, f32x4& bsum) {
    const int nb = 4 * g.ng, total = (tb - ta) * nb;
    const unsigned magic = (unsigned)(((1ULL << 32) + (unsigned)nb - 1) / (unsigned)nb);
#pragma unroll 1
    for (int base = threadIdx.x; base < total; base += 4 * NT) {
        f32x4 dv[4]; bf16x4 av[4]; int rr[4], ff[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = base + k * NT;
            const int r = (int)__umulhi((unsigned)idx, magic), f = idx - r * nb, t = ta + r;
            rr[k] = idx < total ? t : (1 << 30); ff[k] = f;
            dv[k] = zero4(); av[k] = to_bf16(zero4());
            if (idx < total && (unsigned)t < (unsigned)g.T && f < g.F) {
                const long long off = ((long long)(b * g.T + t) * g.F + f) * C;
                if (a.d_bf16) dv[k] = to_f32(*reinterpret_cast<const bf16x4*>(reinterpret_cast<const u16*>(a.d_last) + off));
                else dv[k] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.d_last) + off);
                av[k] = *reinterpret_cast<const bf16x4*>(a.a_last + off);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (rr[k] == (1 << 30)) continue;
            const f32x4 m = to_f32(av[k]);
            f32x4 v = dv[k];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * (m[e] > 0.f ? 1.f : g.alpha);
            if (WANT_B && rr[k] >= t0 && rr[k] < t0 + TR) bsum += v;          // (zero outside the image)
            *reinterpret_cast<bf16x4*>(D + (rr[k] - d_torg) * RS + bin_off(ff[k] + 2)) = to_bf16(v);
        }
    }
}

template <int NT, int MODE>
__global__ __launch_bounds__(NT) void chain_bwd_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) u16 lds[];
    constexpr int NW = NT / 64;
    constexpr bool WG = MODE == MODE_BWD;
    const Geo g = a.g;
    const int L = g.L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    stamp(a.dbg, 0);
    lds_init<NT>(lds, a.lds_bytes, a.tab, nullptr, L, 1);
    stamp(a.dbg, 1);
    LaneOff lo;
    lo.init(lane);
    RowStore<WG ? NT * 64 : NT> rs;         // (the weight-gradient kernel stores no rows: no slots)
    if (!WG) rs.init(g);
    const u16* zero = lds + HDR_ZERO / 2;
    float* bsl = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + HDR_BS);      // [LMAX][NW][4]
    // d_l valid on t0 -+ ext(l): MODE_BWD 2 (l - 1), MODE_DATA 2 l
    const int EXTD = WG ? 2 * (L - 1) : 2 * L;
    const int EXTA = WG ? max(2 * (L - 2), 2) : 2 * (L - 1);
    const int RD = TR + 2 * EXTD, RA = TR + 2 * EXTA;
    u16* D0 = lds + HDR_BYTES / 2;
    u16* D1 = D0 + RD * RS;
    u16* A0 = D1 + RD * RS;
    u16* A1 = A0 + RA * RS;
    constexpr int NSA = ((TR + 4 * (LMAX - 1)) * RSU + NT - 1) / NT;
    constexpr int NSX = ((TR + 4) * 72 + NT - 1) / NT;
    MapDma<NSA, NT> mp;
    mp.init(g);
    const u16* zpage = reinterpret_cast<const u16*>(reinterpret_cast<const char*>(a.tab) + TAB_ZERO_OFF);
    X0Pref<NSX, NT> xp;
    f32x4 acc[LMAX][2];
#pragma unroll
    for (int s = 0; s < LMAX; ++s) { acc[s][0] = zero4(); acc[s][1] = zero4(); }

    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        const int b = tile / g.ntt, t0 = (tile - b * g.ntt) * TR;
        const int dorg = t0 - EXTD, aorg = t0 - EXTA;
        // ---- stage 0: d_L -> D0, a_{L-1} -> A0
        {
            f32x4 bs = zero4();
            const int e = WG ? 2 * (L - 1) : 2 * L;
            if (L >= 2) mp.issue(a.maps + (long long)(L - 2) * g.map_stride, g, b, aorg, RA, A0, zpage);       // in flight under the first stage
            stage_d_last<NT, WG>(a, g, D0, dorg, b, t0 - e, t0 + TR + e, t0, bs);
            if (WG) wave_add4(bsl + ((L - 1) * NW + wave) * 4, bs, lane);
            if (L >= 2) {
                dma_wait();
            } else if (WG) {
                xp.load(a.x0, a.ldx, g, b, aorg, RA);
                xp.commit(A0, RA);
            }
        }
        lds_barrier();
        stamp(a.dbg, 2);
#pragma unroll 1
        for (int s = 0; s < L; ++s) {
            const int l = L - s;                       // this step's layer
            {
                const u16* Dc = (s & 1) ? D1 : D0;
                u16* Dn = (s & 1) ? D0 : D1;
                const u16* Ac = (s & 1) ? A1 : A0;
                u16* An = (s & 1) ? A0 : A1;
                // the map of the next step: a_{l-2} (l - 2 >= 1: a padded map; l == 2: the stack's input, for the first layer's dW)
                stamp_w(a.dbg, s, wave, lane, 0);
                if (l >= 3) mp.issue(a.maps + (long long)(l - 3) * g.map_stride, g, b, aorg, RA, An, zpage);
                else if (l == 2 && WG) xp.load(a.x0, a.ldx, g, b, t0 - 2, TR + 4);        // rows t0 - 2 .. t0 + TR + 2 of the input
                stamp_w(a.dbg, s, wave, lane, 1);
                if (!WG && a.gmaps) rs.run(Dc + (t0 - dorg) * RS, a.gmaps + (long long)(l - 1) * g.map_stride + (long long)(b * g.T + t0) * g.FP * C, min(TR, g.T - t0));
                if (WG && wave < KT) {
                    f32x4 c0 = zero4(), c1 = zero4();
                    dw_step(c0, c1, Ac + (t0 + wave - 2 - aorg) * RS, Dc + (t0 - dorg) * RS, g.ng, zero, lane);
                    stamp_w(a.dbg, s, wave, lane, 2);
                    acc_add(acc, s, c0, c1);
                }
                stamp_w(a.dbg, s, wave, lane, 3);
                if (l >= 2 || !WG) {
                    bf16x8 af[KT];
                    load_afrag(lds + (l - 1) * TLAY, lane, af);
                    const int ext = WG ? 2 * (l - 2) : 2 * (l - 1);
                    const int ta = t0 - ext, tb = t0 + TR + ext;
                    if (l >= 2) {
                        f32x4 bs = zero4();
                        conv_chunks<NW, WG, false, false>(Dc, dorg, af, ta, tb, g, wave, s, lo, zero4(),      // (both pairs' reads in one burst measured 9 % slower here)
                                   [&](int tc, int, int, int px) { return *reinterpret_cast<const bf16x4*>(Ac + (tc - aorg) * RS + px); },
                                   [&](int tc, int t, int, int px, f32x4 v, bf16x4 mk, bool ok, bool fresh) {
                                       const f32x4 m = to_f32(mk);
#pragma unroll
                                       for (int e = 0; e < 4; ++e) v[e] = v[e] * (m[e] > 0.f ? 1.f : g.alpha);
                                       if (!ok) v = zero4();
                                       if (WG && fresh && t >= t0 && t < t0 + TR) bs += v;
                                       *reinterpret_cast<bf16x4*>(Dn + (tc - dorg) * RS + px) = to_bf16(v);
                                   });
                        stamp_w(a.dbg, s, wave, lane, 4);
                        if (WG) wave_add4(bsl + ((l - 2) * NW + wave) * 4, bs, lane);
                        stamp_w(a.dbg, s, wave, lane, 5);
                    } else if (a.g0) {
                        // d/dx0: channel 0 of the transposed first layer, own rows, straight to HBM
                        conv_chunks<NW, false, false>(Dc, dorg, af, t0, t0 + TR, g, wave, s, lo, zero4(),
                                   [](int, int, int, int) { return 0; },
                                   [&](int, int t, int f, int, f32x4 v, int, bool ok, bool) {
                                       if (ok) a.g0[(long long)(b * g.T + t) * g.F + f] = v[0];
                                   });
                    }
                }
                if (l >= 3) dma_wait();
                else if (l == 2 && WG) xp.commit(An + (EXTA - 2) * RS, TR + 4);
                stamp_w(a.dbg, s, wave, lane, 6);
                lds_barrier();
                stamp_w(a.dbg, s, wave, lane, 7);
                stamp(a.dbg, 3 + s);
            }
        }
        a.dbg = nullptr;
    }
    if (!WG) return;
    // ---- one reduction per workgroup, fixed order: red[s][kt][hb][r][lane]
    float* red = reinterpret_cast<float*>(lds + HDR_BYTES / 2);
    if (wave < KT) {
#pragma unroll
        for (int s = 0; s < LMAX; ++s)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(((s * KT + wave) * 2 + hb) * 4 + r) * 64 + lane] = acc[s][hb][r];
    }
    __syncthreads();
    for (int i = tid; i < L * KT * KF * 16; i += NT) {
        const int li = i / (KT * KF * 16), j = i - li * (KT * KF * 16);       // layer index li = l - 1 <-> step s = L - 1 - li
        const int s = L - 1 - li;
        const int co = j & 3, ci = (j >> 2) & 3, kf = (j >> 4) % KF, kt = (j >> 4) / KF;
        const int cin = li == 0 ? a.cin0 : C;
        if (ci >= cin) continue;
        float sum = 0.f;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
                const int fo = fi + 4 - 4 * hb - kf;
                if (fo >= 0 && fo < 4) sum += red[(((s * KT + kt) * 2 + hb) * 4 + ci) * 64 + 16 * fi + 4 * fo + co];
            }
        a.partials[((size_t)li * gridDim.x + blockIdx.x) * NPART + ((kt * KF + kf) * cin + ci) * C + co] = sum;
    }
    if (tid < L * C) {
        const int li = tid >> 2, co = tid & 3;
        const int cin = li == 0 ? a.cin0 : C;
        float sum = 0.f;
        for (int w = 0; w < NW; ++w) sum += bsl[(li * NW + w) * 4 + co];
        a.partials[((size_t)li * gridDim.x + blockIdx.x) * NPART + KT * KF * cin * C + co] = sum;
    }
}

// ------------------------------------------------------------------------------------------------------------
// second-order sweep (the backward of MODE_DATA): u_0 = d/d(g0) [B][T][F] fp32;  for l = 1 .. L
//     dW_l += corr(u_{l-1}, gamma_l)           (u_{l-1} in the role of the layer input, gamma_l in the role of d_l)
//     u_l   = lrelu'(a_l) . (W_l * u_{l-1})    (forward table, no bias)
// u_L [B][T][F][4] is the gradient w.r.t. dL/da_L.  Tiles U0 / U1 (row i <-> t0 - 2 L + i; u_l valid on t0 -+ 2 (L - l)),
// gamma tiles G0 / G1 (own rows); the mask source a_l is read from HBM in the epilogue (eight bytes per pixel).
// ------------------------------------------------------------------------------------------------------------
struct SecArgs {
    const float* u0;
    const u16* gmaps; const u16* maps; const u16* a_last;
    const u16* tab;
    void* out; int out_bf16;
    float* partials;
    int cin0;
    Geo g; size_t lds_bytes;
    unsigned long long* dbg;
};

template <int NT>
__global__ __launch_bounds__(NT) void chain_second_kernel(SecArgs a) {
    extern __shared__ __attribute__((aligned(16))) u16 lds[];
    constexpr int NW = NT / 64;
    const Geo g = a.g;
    const int L = g.L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    stamp(a.dbg, 0);
    lds_init<NT>(lds, a.lds_bytes, a.tab, nullptr, L, 0);
    stamp(a.dbg, 1);
    LaneOff lo;
    lo.init(lane);
    const u16* zero = lds + HDR_ZERO / 2;
    const int RU = TR + 4 * L;
    u16* U0 = lds + HDR_BYTES / 2;
    u16* U1 = U0 + RU * RS;
    u16* G0 = U1 + RU * RS;
    u16* G1 = G0 + TR * RS;
    constexpr int NSG = (TR * RSU + NT - 1) / NT;
    constexpr int NSX = ((TR + 4 * LMAX) * 72 + NT - 1) / NT;
    MapDma<NSG, NT> mp;
    mp.init(g);
    const u16* zpage = reinterpret_cast<const u16*>(reinterpret_cast<const char*>(a.tab) + TAB_ZERO_OFF);
    f32x4 acc[LMAX][2];
#pragma unroll
    for (int s = 0; s < LMAX; ++s) { acc[s][0] = zero4(); acc[s][1] = zero4(); }

    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        const int b = tile / g.ntt, t0 = (tile - b * g.ntt) * TR;
        const int uorg = t0 - 2 * L;
        {
            X0Pref<NSX, NT> xp;
            xp.load(a.u0, g.F, g, b, uorg, RU);
            mp.issue(a.gmaps, g, b, t0, TR, G0, zpage);
            xp.commit(U0, RU);
            dma_wait();
        }
        lds_barrier();
        stamp(a.dbg, 2);
#pragma unroll 1
        for (int s = 0; s < L; ++s) {
            const int l = s + 1;
            {
                const u16* Uc = (s & 1) ? U1 : U0;
                u16* Un = (s & 1) ? U0 : U1;
                const u16* Gc = (s & 1) ? G1 : G0;
                u16* Gn = (s & 1) ? G0 : G1;
                if (l < L) mp.issue(a.gmaps + (long long)l * g.map_stride, g, b, t0, TR, Gn, zpage);
                if (wave < KT) {
                    f32x4 c0 = zero4(), c1 = zero4();
                    dw_step(c0, c1, Uc + (t0 + wave - 2 - uorg) * RS, Gc, g.ng, zero, lane);
                    acc_add(acc, s, c0, c1);
                }
                bf16x8 af[KT];
                load_afrag(lds + (l - 1) * TLAY, lane, af);
                const int ext = 2 * (L - l);
                const int ta = t0 - ext, tb = t0 + TR + ext;
                const bool last = l == L;
                const u16* am = last ? a.a_last : a.maps + (long long)(l - 1) * g.map_stride;
                const int apitch = last ? g.F : g.FP;
                conv_chunks<NW, true, false, true>(Uc, uorg, af, ta, tb, g, wave, s, lo, zero4(),
                           [&](int, int t, int f, int) {
                               bf16x4 m = to_bf16(zero4());
                               if ((unsigned)t < (unsigned)g.T && f < g.F) m = *reinterpret_cast<const bf16x4*>(am + ((long long)(b * g.T + t) * apitch + f) * C);
                               return m;
                           },
                           [&](int tc, int t, int f, int px, f32x4 v, bf16x4 mk, bool ok, bool) {
                               const f32x4 m = to_f32(mk);
#pragma unroll
                               for (int e = 0; e < 4; ++e) v[e] = v[e] * (m[e] > 0.f ? 1.f : g.alpha);
                               if (!ok) v = zero4();
                               if (!last) *reinterpret_cast<bf16x4*>(Un + (tc - uorg) * RS + px) = to_bf16(v);
                               else if (ok) {
                                   const long long off = ((long long)(b * g.T + t) * g.F + f) * C;
                                   if (a.out_bf16) *reinterpret_cast<bf16x4*>(reinterpret_cast<u16*>(a.out) + off) = to_bf16(v);
                                   else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + off) = v;
                               }
                           });
                if (l < L) dma_wait();
                lds_barrier();
                stamp(a.dbg, 3 + s);
            }
        }
        a.dbg = nullptr;
    }
    float* red = reinterpret_cast<float*>(lds + HDR_BYTES / 2);
    if (wave < KT) {
#pragma unroll
        for (int s = 0; s < LMAX; ++s)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(((s * KT + wave) * 2 + hb) * 4 + r) * 64 + lane] = acc[s][hb][r];
    }
    __syncthreads();
    for (int i = tid; i < L * KT * KF * 16; i += NT) {
        const int li = i / (KT * KF * 16), j = i - li * (KT * KF * 16);       // layer index li = l - 1 = step s
        const int co = j & 3, ci = (j >> 2) & 3, kf = (j >> 4) % KF, kt = (j >> 4) / KF;
        const int cin = li == 0 ? a.cin0 : C;
        if (ci >= cin) continue;
        float sum = 0.f;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
                const int fo = fi + 4 - 4 * hb - kf;
                if (fo >= 0 && fo < 4) sum += red[(((li * KT + kt) * 2 + hb) * 4 + ci) * 64 + 16 * fi + 4 * fo + co];
            }
        a.partials[((size_t)li * gridDim.x + blockIdx.x) * NPART + ((kt * KF + kf) * cin + ci) * C + co] = sum;
    }
    if (tid < L * C) {       // no bias term in the second-order sweep: the row's bias slots are zero
        const int li = tid >> 2, co = tid & 3;
        const int cin = li == 0 ? a.cin0 : C;
        a.partials[((size_t)li * gridDim.x + blockIdx.x) * NPART + KT * KF * cin * C + co] = 0.f;
    }
}

}  // namespace c2c
}  // namespace ptts

using namespace ptts;
using namespace ptts::c2c;

namespace {
constexpr int NT = 512;
// threads of the kernels without weight-gradient accumulators (forward, backward data): 1024 (16 waves: 7-9 % faster, a wave
// issues one instruction per four cycles and has half the units of a step) or 512 (PTTS_CHAIN_NT=512)
int wide_nt() {
    static int nt = -1;
    if (nt < 0) { const char* e = getenv("PTTS_CHAIN_NT"); nt = (e && atoi(e) == 512) ? 512 : 1024; }
    return nt;
}

bool make_geo(Geo& g, int B, int T, int F, int L, float alpha) {
    g.B = B; g.T = T; g.F = F; g.FP = (F + 1) & ~1; g.L = L; g.ng = (F + 3) / 4;
    g.ntt = (T + TR - 1) / TR;
    g.ntiles = B * g.ntt;
    const unsigned upr = (unsigned)(g.FP / 2);
    g.magic_fp2 = upr > 1 ? (unsigned)(((1ULL << 32) + upr - 1) / upr) : 0u;
    g.alpha = alpha;
    g.map_stride = (long long)B * T * g.FP * C;
    return true;
}

int check_common(const char* what, int B, int T, int F, int L, float alpha) {
    PTTS_REQUIRE(B > 0 && T > 0, "%s: bad dims B=%d T=%d", what, B, T);
    PTTS_REQUIRE(F >= 2 && F <= 4 * NGMAX, "%s: F = %d outside [2, %d] (one block of %d bin groups)", what, F, 4 * NGMAX, NGMAX);
    PTTS_REQUIRE(L >= 1 && L <= LMAX, "%s: %d layers (1 .. %d)", what, L, LMAX);
    PTTS_REQUIRE(alpha > 0.f && alpha <= 1.f, "%s: LeakyReLU slope %g outside (0, 1]: the stored activations must keep the sign", what, alpha);
    PTTS_REQUIRE((long long)B * T * ((F + 1) & ~1) * C < (1LL << 31), "%s: map too large for 32-bit frame offsets", what);
    PTTS_REQUIRE((long long)B * ((T + TR - 1) / TR) < (1LL << 30), "%s: too many tiles", what);
    return PTTS_OK;
}

template <class K>
void set_lds(K kernel) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
}
}  // namespace

// 1 when the stack (Cin0 -> C -> ... -> C, KT x KF kernels, F bins) has a chain kernel
extern "C" int ptts_conv2d_chain_supported(int F, int L, int Cin0, int Cc, int KT_, int KF_) {
    return (F >= 2 && F <= 4 * NGMAX && L >= 1 && L <= LMAX && Cin0 >= 1 && Cin0 <= 4 && Cc == 4 && KT_ == 5 && KF_ == 5) ? 1 : 0;
}

// measurement hook: a buffer of 256 x 32 uint64 that receives the phase stamps of the next launches (NULL: off)
extern "C" int ptts_conv2d_chain_debug(void* stamp_buf) { g_dbg = (unsigned long long*)stamp_buf; return PTTS_OK; }

extern "C" size_t ptts_conv2d_chain_tables_bytes(void) { return TAB_BYTES; }
extern "C" size_t ptts_conv2d_chain_partials_bytes(int L) { return (size_t)L * NCU * NPART * sizeof(float); }
// elements (bf16) of one padded internal map [B][T][FP][4]
extern "C" long long ptts_conv2d_chain_map_elems(int B, int T, int F) { return (long long)B * T * ((F + 1) & ~1) * C; }

// w[l] / b[l]: HOST arrays of L device pointers (kernels [5][5][Cin_l][4] with Cin_0 = cin0, Cin_l = 4; biases [4] or NULL)
extern "C" int ptts_conv2d_chain_tables(const float* const* w, const float* const* b, void* tables, int L, int cin0, void* stream) {
    PTTS_REQUIRE(w && tables && L >= 1 && L <= LMAX && cin0 >= 1 && cin0 <= 4, "conv2d_chain_tables: bad arguments (L=%d, cin0=%d)", L, cin0);
    TabArgs a;
    for (int l = 0; l < LMAX; ++l) { a.w[l] = nullptr; a.b[l] = nullptr; a.cin[l] = C; }
    for (int l = 0; l < L; ++l) {
        PTTS_REQUIRE(w[l], "conv2d_chain_tables: kernel %d is null", l);
        a.w[l] = w[l]; a.b[l] = b ? b[l] : nullptr; a.cin[l] = l == 0 ? cin0 : C;
    }
    hipLaunchKernelGGL(chain_tables_kernel, dim3(2 * L), dim3(256), 0, (hipStream_t)stream, a, (u16*)tables,
                       reinterpret_cast<float*>((char*)tables + TAB_BIAS_OFF));
    return check_launch("conv2d_chain_tables");
}

// forward of the stack: x0 [B][T][ldx] fp32 (the first F columns of a row are the spectrum) -> maps a_1 .. a_{L-1}
// ([L-1][B][T][FP][4] bf16, FP = F rounded up to even) and a_last = a_L [B][T][F][4] bf16
extern "C" int ptts_conv2d_chain_fwd(const float* x0, long long ldx, const void* tables, void* maps, void* a_last,
                                     int B, int T, int F, int L, float alpha, void* stream) {
    int rc = check_common("conv2d_chain_fwd", B, T, F, L, alpha);
    if (rc) return rc;
    PTTS_REQUIRE(x0 && tables && a_last && (maps || L == 1) && ldx >= F, "conv2d_chain_fwd: null tensor or ldx < F");
    FwdArgs a;
    make_geo(a.g, B, T, F, L, alpha);
    a.x0 = x0; a.ldx = ldx; a.tab = (const u16*)tables; a.bias = reinterpret_cast<const float*>((const char*)tables + TAB_BIAS_OFF);
    a.maps = (u16*)maps; a.a_last = (u16*)a_last; a.dbg = g_dbg;
    a.lds_bytes = HDR_BYTES + (size_t)2 * (TR + 4 * L) * RS * sizeof(u16);
    PTTS_REQUIRE(a.lds_bytes <= LDS_MAX, "conv2d_chain_fwd: tiles do not fit the LDS");
    static bool attr = false;
    if (!attr) { set_lds(&chain_fwd_kernel<512>); set_lds(&chain_fwd_kernel<1024>); attr = true; }
    const int grid = std::min(a.g.ntiles, NCU);
    if (wide_nt() == 1024) hipLaunchKernelGGL(chain_fwd_kernel<1024>, dim3(grid), dim3(1024), a.lds_bytes, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(chain_fwd_kernel<512>, dim3(grid), dim3(512), a.lds_bytes, (hipStream_t)stream, a);
    return check_launch("conv2d_chain_fwd");
}

// first-order backward: per-workgroup partial sums of dW_l / db_l, rows [l][nblocks][npart] (layer l's row: KT*KF*Cin_l*4 kernel
// entries, then 4 bias entries) for ptts_conv2d_reduce_grouped.  d_last = dL/da_L [B][T][F][4], fp32 or bf16.
extern "C" int ptts_conv2d_chain_bwd(const void* d_last, int d_bf16, const float* x0, long long ldx, const void* maps, const void* a_last,
                                     const void* tables, float* partials, size_t partials_bytes, int* nblocks_out, int* npart_out,
                                     int B, int T, int F, int L, int cin0, float alpha, void* stream) {
    int rc = check_common("conv2d_chain_bwd", B, T, F, L, alpha);
    if (rc) return rc;
    PTTS_REQUIRE(d_last && x0 && a_last && tables && partials && nblocks_out && npart_out && (maps || L == 1) && ldx >= F,
                 "conv2d_chain_bwd: null pointer or ldx < F");
    PTTS_REQUIRE(cin0 >= 1 && cin0 <= 4, "conv2d_chain_bwd: cin0 = %d", cin0);
    BwdArgs a;
    make_geo(a.g, B, T, F, L, alpha);
    const int grid = std::min(a.g.ntiles, NCU);
    if (partials_bytes < (size_t)L * grid * NPART * sizeof(float)) { set_error("conv2d_chain_bwd: partials buffer too small"); return PTTS_EWORKSPACE; }
    a.d_last = d_last; a.d_bf16 = d_bf16; a.x0 = x0; a.ldx = ldx; a.maps = (const u16*)maps; a.a_last = (const u16*)a_last;
    a.tab = (const u16*)tables; a.gmaps = nullptr; a.g0 = nullptr; a.partials = partials; a.cin0 = cin0; a.dbg = g_dbg;
    const int EXTD = 2 * (L - 1), EXTA = std::max(2 * (L - 2), 2);
    a.lds_bytes = HDR_BYTES + (size_t)(2 * (TR + 2 * EXTD) + 2 * (TR + 2 * EXTA)) * RS * sizeof(u16);
    a.lds_bytes = std::max(a.lds_bytes, (size_t)HDR_BYTES + (size_t)LMAX * KT * 2 * 4 * 64 * sizeof(float));
    PTTS_REQUIRE(a.lds_bytes <= LDS_MAX, "conv2d_chain_bwd: tiles do not fit the LDS");
    static bool attr = false;
    if (!attr) { set_lds(&chain_bwd_kernel<NT, MODE_BWD>); attr = true; }
    hipLaunchKernelGGL((chain_bwd_kernel<NT, MODE_BWD>), dim3(grid), dim3(NT), a.lds_bytes, (hipStream_t)stream, a);
    *nblocks_out = grid; *npart_out = NPART;
    return check_launch("conv2d_chain_bwd");
}

// backward-data chain: g0 = d(.)/dx0 [B][T][F] fp32 (or NULL) and, when gmaps is given, gamma_l = lrelu'(a_l) . dL/da_l for
// l = 1 .. L as padded maps [L][B][T][FP][4] bf16 (the operands of the second-order sweep)
extern "C" int ptts_conv2d_chain_bwd_data(const void* d_last, int d_bf16, const void* maps, const void* a_last, const void* tables,
                                          void* gmaps, float* g0, int B, int T, int F, int L, float alpha, void* stream) {
    int rc = check_common("conv2d_chain_bwd_data", B, T, F, L, alpha);
    if (rc) return rc;
    PTTS_REQUIRE(d_last && a_last && tables && (maps || L == 1) && (gmaps || g0), "conv2d_chain_bwd_data: null pointer");
    BwdArgs a;
    make_geo(a.g, B, T, F, L, alpha);
    a.d_last = d_last; a.d_bf16 = d_bf16; a.x0 = nullptr; a.ldx = 0; a.maps = (const u16*)maps; a.a_last = (const u16*)a_last;
    a.tab = (const u16*)tables; a.gmaps = (u16*)gmaps; a.g0 = g0; a.partials = nullptr; a.cin0 = 1; a.dbg = g_dbg;
    const int EXTD = 2 * L, EXTA = 2 * (L - 1);
    a.lds_bytes = HDR_BYTES + (size_t)(2 * (TR + 2 * EXTD) + 2 * (TR + 2 * EXTA)) * RS * sizeof(u16);
    PTTS_REQUIRE(a.lds_bytes <= LDS_MAX, "conv2d_chain_bwd_data: tiles do not fit the LDS");
    static bool attr = false;
    if (!attr) { set_lds(&chain_bwd_kernel<512, MODE_DATA>); set_lds(&chain_bwd_kernel<1024, MODE_DATA>); attr = true; }
    const int grid = std::min(a.g.ntiles, NCU);
    if (wide_nt() == 1024) hipLaunchKernelGGL((chain_bwd_kernel<1024, MODE_DATA>), dim3(grid), dim3(1024), a.lds_bytes, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((chain_bwd_kernel<512, MODE_DATA>), dim3(grid), dim3(512), a.lds_bytes, (hipStream_t)stream, a);
    return check_launch("conv2d_chain_bwd_data");
}

// second-order sweep: u0 = d/d(g0) [B][T][F] fp32 -> out = d/d(d_last) [B][T][F][4] (fp32 or bf16) and the partial sums of dW_l
// (rows as ptts_conv2d_chain_bwd; the bias slots are zero)
extern "C" int ptts_conv2d_chain_second(const float* u0, const void* gmaps, const void* maps, const void* a_last, const void* tables,
                                        void* out, int out_bf16, float* partials, size_t partials_bytes, int* nblocks_out, int* npart_out,
                                        int B, int T, int F, int L, int cin0, float alpha, void* stream) {
    int rc = check_common("conv2d_chain_second", B, T, F, L, alpha);
    if (rc) return rc;
    PTTS_REQUIRE(u0 && gmaps && a_last && tables && out && partials && nblocks_out && npart_out && (maps || L == 1), "conv2d_chain_second: null pointer");
    PTTS_REQUIRE(cin0 >= 1 && cin0 <= 4, "conv2d_chain_second: cin0 = %d", cin0);
    SecArgs a;
    make_geo(a.g, B, T, F, L, alpha);
    const int grid = std::min(a.g.ntiles, NCU);
    if (partials_bytes < (size_t)L * grid * NPART * sizeof(float)) { set_error("conv2d_chain_second: partials buffer too small"); return PTTS_EWORKSPACE; }
    a.u0 = u0; a.gmaps = (const u16*)gmaps; a.maps = (const u16*)maps; a.a_last = (const u16*)a_last; a.tab = (const u16*)tables;
    a.out = out; a.out_bf16 = out_bf16; a.partials = partials; a.cin0 = cin0; a.dbg = g_dbg;
    a.lds_bytes = HDR_BYTES + (size_t)(2 * (TR + 4 * L) + 2 * TR) * RS * sizeof(u16);
    a.lds_bytes = std::max(a.lds_bytes, (size_t)HDR_BYTES + (size_t)LMAX * KT * 2 * 4 * 64 * sizeof(float));
    PTTS_REQUIRE(a.lds_bytes <= LDS_MAX, "conv2d_chain_second: tiles do not fit the LDS");
    static bool attr = false;
    if (!attr) { set_lds(&chain_second_kernel<NT>); attr = true; }
    hipLaunchKernelGGL(chain_second_kernel<NT>, dim3(grid), dim3(NT), a.lds_bytes, (hipStream_t)stream, a);
    *nblocks_out = grid; *npart_out = NPART;
    return check_launch("conv2d_chain_second");
}

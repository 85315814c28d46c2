// fp32 products on the bf16 matrix cores: three-way operand split ("bf16x6").
//
// Every fp32 operand x is written as x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)
// (round to nearest even; the subtractions are exact in fp32, and 3 x 8 significant bits cover the 24 of fp32).  The
// product a.b is then the sum of the six bf16 products a1b1, a1b2, a2b1, a1b3, a2b2, a3b1 (the three left out are below
// 2^-24 of a.b), each exact in the matrix core's fp32 product and accumulated in fp32 -- the same accumulation the
// fp32 MFMA does.  Six v_mfma_f32_16x16x32_bf16 (16 cycles each, K = 32) replace eight v_mfma_f32_16x16x4_f32
// (32 cycles each) for the same 16x16x32 block: 2.7x less matrix-pipe time at fp32-level accuracy
// (tools/bf16x6_accuracy.py: max error 4.6e-6 of mean |C| at K = 12 621, against 3.8e-6 for the fp32 product).
//
// Role on the hot path: the context Conv1D forward of critic and generator (reference networktts.py:116-120), the
// largest flop item of both networks.  On by default (PTTS_CONV1D_SPLIT=0 / cfg.train_wgan_split_bf16 = False select the fp32 MFMA kernels): see DESIGN.md.
//
// Data: the split pass writes the zero-padded frames as three bf16 planes [Cp/32][B*(T+KW-1)][32] (32-channel blocks, Cp
// = C rounded up to 32) and the kernel [KW][C][N] as three TRANSPOSED planes [Cp/32][N][KW][32]: both MFMA operands
// are k-contiguous, and what a workgroup reads in a row is contiguous in memory (a tile's frames of a channel block are
// one run of 64-byte rows; the taps of (channel block, n) lie next to each other, so no half of a 128-byte line is
// fetched in vain).  The GEMM is stream-K over (128x128 tile, channel block) units, its k-step 32 wide with the tap
// innermost (see split_segment): a frame image of [176 rows][32 bf16] x 3 planes per channel block (double-buffered) and
// a B stage of [128 rows][32 bf16] x 3 planes per k-step (three stages deep) in LDS, rows of 64 B with their four
// 16-byte quads XOR-swizzled by (row>>1)&2 -> conflict-free ds_read_b128 fragments at every tap shift (the hardware
// serves that instruction in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... : 16 consecutive rows, two
// neighbouring quads); everything arrives by
// global_load_lds_dwordx4.
#include "common.h"
#include <cstdlib>

namespace ptts {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

constexpr int SBM = 128, SBN = 128, SBK = 32;
constexpr int S_THREADS = 512;               // 8 waves, two per SIMD: one wave alone issues a bf16 MFMA only every 27 cycles
constexpr int S_WAVES = S_THREADS / 64;      // (tools/mfma_clock_bf16.hip), two waves together one per 17.5
constexpr int S_IMG = SBN * SBK;                 // elements (bf16) of one B plane of a k-step: 8 KB
constexpr int S_BSTAGE = 3 * S_IMG;              // B1 B2 B3 of a k-step: 24 KB
constexpr int S_STAGES = 3;
constexpr int S_AROWS = 176;                     // frames of an A image: 128 + (KW-1) halo + pad rows of one batch boundary
constexpr int S_AIMG = 3 * S_AROWS * SBK;        // A1 A2 A3 of a channel block: 33 KB
constexpr int S_LDS_BYTES = (2 * S_AIMG + S_STAGES * S_BSTAGE) * 2;   // 138 KB of the CU's 160

struct SplitGemmArgs {
    const u16* A[3]; const u16* Bt[3]; const float* bias; float* C;
    int M, N;
    int rows_per_seg, seg_rows;                  // frames per utterance segment (T) and rows of a padded segment (T + KW - 1)
    long long a_rows;                            // rows of a padded plane (segments * seg_rows)
    int tiles_n, taps, cblocks;
    long long iters_total;                       // tiles * cblocks
    int workers;
};

__device__ __forceinline__ u16 bf16_rn(float f) {
    const unsigned u = __float_as_uint(f);
    return (u16)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_f(u16 h) { return __uint_as_float((unsigned)h << 16); }

__device__ __forceinline__ void split3(float x, u16& h1, u16& h2, u16& h3) {
    h1 = bf16_rn(x);
    const float r1 = x - bf16_f(h1);
    h2 = bf16_rn(r1);
    const float r2 = r1 - bf16_f(h2);
    h3 = bf16_rn(r2);
}

struct alignas(16) U16x8 { u16 v[8]; };

// x [B][T][C] fp32 -> planes [Cp/32][B*(T+pl+pr)][32] bf16 (zero rows in front / behind each utterance, zero channels C..Cp-1).
__global__ __launch_bounds__(256) void split3_frames_kernel(const float* __restrict__ x, u16* __restrict__ p1,
                                                            u16* __restrict__ p2, u16* __restrict__ p3, int B, int T,
                                                            int C, int pl, int pr, int Cp) {
    const int groups = Cp / 8;
    const long long total = (long long)B * (T + pl + pr) * groups;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int gq = (int)(i % groups);
        const long long row = i / groups;
        const int tp = (int)(row % (T + pl + pr));
        const int b = (int)(row / (T + pl + pr));
        const int t = tp - pl, c0 = gq * 8;
        U16x8 o1, o2, o3;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = 0.f;
            if (t >= 0 && t < T && c0 + e < C) v = x[((long long)b * T + t) * C + c0 + e];
            split3(v, o1.v[e], o2.v[e], o3.v[e]);
        }
        const long long o = ((long long)(c0 / SBK) * B * (T + pl + pr) + row) * SBK + c0 % SBK;
        *reinterpret_cast<U16x8*>(p1 + o) = o1;
        *reinterpret_cast<U16x8*>(p2 + o) = o2;
        *reinterpret_cast<U16x8*>(p3 + o) = o3;
    }
}

// w [KW][C][N] fp32 -> planes [Cp/32][N][KW][32] bf16 (transposed: k-contiguous per output channel, the taps of a channel
// block next to each other).
__global__ __launch_bounds__(256) void split3_weight_t_kernel(const float* __restrict__ w, u16* __restrict__ p1,
                                                              u16* __restrict__ p2, u16* __restrict__ p3, int KW, int C,
                                                              int N, int Cp) {
    const int groups = Cp / 8;
    const long long total = (long long)KW * groups * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % N);                 // consecutive lanes: consecutive n (coalesced reads of w)
        const int gq = (int)((i / N) % groups);
        const int kw = (int)(i / ((long long)N * groups));
        const int c0 = gq * 8;
        U16x8 o1, o2, o3;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = 0.f;
            if (c0 + e < C) v = w[((long long)kw * C + c0 + e) * N + n];
            split3(v, o1.v[e], o2.v[e], o3.v[e]);
        }
        const long long o = (((long long)(c0 / SBK) * N + n) * KW + kw) * SBK + c0 % SBK;
        *reinterpret_cast<U16x8*>(p1 + o) = o1;
        *reinterpret_cast<U16x8*>(p2 + o) = o2;
        *reinterpret_cast<U16x8*>(p3 + o) = o3;
    }
}

typedef void __attribute__((address_space(3)))* s_lptr;

// see gemm.hip dma16: inline assembly so that the kernel, not the compiler, orders the DMAs
__device__ __forceinline__ void dma16h(const u16* src, u16* lds_wave_base) {
    const unsigned lds_off = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long)(s_lptr)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_off) : "memory", "m0");
}

// padded-buffer row of frame m (its tap 0)
__device__ __forceinline__ int s_prow(const SplitGemmArgs& g, int m) {
    const int seg = m / g.rows_per_seg;
    return seg * g.seg_rows + (m - seg * g.rows_per_seg);
}

// One segment of a tile: channel blocks cb0..cb1-1 (32 channels each), all KW taps of each.  The k index runs
// (channel block, tap) with the tap innermost: the A operand of tap j is the A operand of tap 0 shifted down by j frames,
// so ONE LDS image of the tile's frames (+ KW-1 halo rows, + the pad rows of a batch boundary inside the tile) serves
// all KW k-steps of a channel block -- A is fetched once per channel block instead of once per k-step, and what
// streams per k-step is the B slice only.
__device__ __forceinline__ void split_segment(const SplitGemmArgs& g, u16* lds, int tile, int cb0, int cb1) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform for the compiler too: scalar plane / LDS addressing
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 64;             // wave tile 32 (m) x 64 (n): 2 x 4 MFMA blocks
    const int li = lane & 15, lg = lane >> 4;
    const int m0 = (tile / g.tiles_n) * SBM, n0 = (tile % g.tiles_n) * SBN;
    const int KW = g.taps;
    u16* aimg = lds;                                // 2 x A image (3 planes x S_AROWS rows)
    u16* bst = lds + 2 * S_AIMG;                    // S_STAGES x B stage (3 planes x 128 rows)

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A image: S_AROWS consecutive rows of the padded frame buffer from the tile's first frame; wave-instruction u
    // (0 .. 3*S_AROWS/16-1; this wave takes u = wave, wave+8, ...) fills 16 rows of plane u / (S_AROWS/16)
    const int prow0 = s_prow(g, m0);
    constexpr int A_INSTR = 3 * (S_AROWS / 16);     // 33
    constexpr int A_PER_WAVE = (A_INSTR + S_WAVES - 1) / S_WAVES;   // 5 (wave 0; the others issue 4)
    const int n_a = (A_INSTR - wave + S_WAVES - 1) / S_WAVES;       // instructions of this wave
    auto issue_a = [&](int cb, int buf) {
        u16* img = aimg + buf * S_AIMG;
#pragma unroll
        for (int x = 0; x < A_PER_WAVE; ++x) {
            const int u = wave + S_WAVES * x;
            if (u < A_INSTR) {
                const int p = u / (S_AROWS / 16), rb = (u % (S_AROWS / 16)) * 16;
                const int r = rb + (lane >> 2);
                const int cg = (lane & 3) ^ ((r >> 1) & 2);
                long long row = prow0 + r;
                if (row >= g.a_rows) row = g.a_rows - 1;             // rows behind the buffer: never used by a stored row
                const u16* ap = p == 0 ? g.A[0] : (p == 1 ? g.A[1] : g.A[2]);
                dma16h(ap + ((long long)cb * g.a_rows + row) * SBK + 8 * cg, img + p * (S_AROWS * SBK) + rb * SBK);
            }
        }
    };
    long long boff;                                         // this wave: rows 16 wave .. 16 wave + 15 of every B plane
    {
        const int r = 16 * wave + (lane >> 2);
        const int cg = (lane & 3) ^ ((r >> 1) & 2);
        boff = (long long)(n0 + r) * (KW * SBK) + 8 * cg;
    }
    auto issue_b = [&](int cb, int tap, int stage) {        // 3 DMA wave-instructions
        u16* st = bst + stage * S_BSTAGE;
        const long long k0 = ((long long)cb * g.N * KW + tap) * SBK;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            dma16h(g.Bt[p] + boff + k0, st + p * S_IMG + 16 * wave * SBK);
    };
    // image row of this lane's fragment rows at tap 0: frames of the tile, pad rows of a batch boundary skipped
    int arow[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int m = min(m0 + wm + 16 * f + li, g.M - 1);
        arow[f] = s_prow(g, m) - prow0;
    }
    const int bfrag = (wn + li) * SBK + 8 * (lg ^ ((li >> 1) & 2));

    const int nsteps = (cb1 - cb0) * KW;
    // step s = (cb0 + s / KW, s % KW).  Software-pipelined over the barrier: the fragments of step s+1 are read from LDS
    // into a second register set while the MFMAs of step s run from the first, so the LDS phase (144 KB of fragment
    // reads per k-step, half the MFMA time) is hidden instead of standing in front of the MFMAs of all eight waves.
    // Stage s%3 is therefore free once barrier s is passed (its fragments were read during step s-1), and B(s+3) is
    // issued into it at step s: still two steps of DMA latency from three stages.
    // Issue order: [A(cb0), B(0), B(1), B(2)], then at step s (after its barrier): A(cb+1) when s is a tap 0 and another
    // block follows, then B(s+3).
    auto step_pos = [&](int s, int& c, int& t) { c = cb0 + s / KW; t = s - (s / KW) * KW; };
    auto read_frags = [&](int s, int c, int t, bf16x8 (&a)[3][2], bf16x8 (&b)[3][4]) {
        const u16* img = aimg + ((c - cb0) & 1) * S_AIMG;
        const u16* st = bst + (s % S_STAGES) * S_BSTAGE;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int row = arow[f] + t;
            const int ao = row * SBK + 8 * (lg ^ ((row >> 1) & 2));
#pragma unroll
            for (int p = 0; p < 3; ++p) a[p][f] = *reinterpret_cast<const bf16x8*>(img + p * (S_AROWS * SBK) + ao);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int f = 0; f < 4; ++f) b[p][f] = *reinterpret_cast<const bf16x8*>(st + p * S_IMG + bfrag + 16 * f * SBK);
    };
    issue_a(cb0, 0);
    {
        int c, t;
        step_pos(0, c, t); issue_b(c, t, 0);
        if (nsteps > 1) { step_pos(1, c, t); issue_b(c, t, 1); }
        if (nsteps > 2) { step_pos(2, c, t); issue_b(c, t, 2); }
    }
    if (nsteps > 2) __builtin_amdgcn_s_waitcnt(0xF76);            // vmcnt(6): A(cb0) and B(0) have landed
    else if (nsteps > 1) __builtin_amdgcn_s_waitcnt(0xF73);       // vmcnt(3)
    else __builtin_amdgcn_s_waitcnt(0xF70);
    __syncthreads();
    bf16x8 fa0[3][2], fb0[3][4], fa1[3][2], fb1[3][4];
    read_frags(0, cb0, 0, fa0, fb0);
    int s = 0, cb = cb0, tap = 0;
    // one step: MFMAs from (a, b); fragments of the next step into (na, nb)
    auto step = [&](bf16x8 (&a)[3][2], bf16x8 (&b)[3][4], bf16x8 (&na)[3][2], bf16x8 (&nb)[3][4]) {
        if (s + 1 < nsteps) {
            // DMAs of this wave issued after B(s+1): B(s+2), and, when step s-1 was a tap 0 with a following block, the
            // A image issued there right before B(s+2)
            const bool b_next = s + 2 < nsteps;
            const bool a_after = (tap == 1) && (cb + 1 < cb1) && KW > 1 && s >= 1;
            if (b_next && a_after) { if (n_a == 5) __builtin_amdgcn_s_waitcnt(0xF78); else __builtin_amdgcn_s_waitcnt(0xF77); }   // vmcnt(3 + n_a)
            else if (b_next) __builtin_amdgcn_s_waitcnt(0xF73);       // vmcnt(3)
            else __builtin_amdgcn_s_waitcnt(0xF70);                   // vmcnt(0)
        }
        __syncthreads();
        if (tap == 0 && cb + 1 < cb1) issue_a(cb + 1, ((cb + 1 - cb0) & 1));
        if (s + 3 < nsteps) { int c, t; step_pos(s + 3, c, t); issue_b(c, t, s % S_STAGES); }
        {   // unconditional (the last step re-reads its own, still valid, stage): a branch here would keep the reads in
            // a basic block of their own, in front of the MFMAs instead of between them
            const int sn = min(s + 1, nsteps - 1);
            int c, t; step_pos(sn, c, t); read_frags(sn, c, t, na, nb);
        }
        // small terms first; consecutive MFMAs go to different accumulators
#define PTTS_PROD(PA, PB)                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                       \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                       \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA][i], b[PB][j], acc[i][j], 0, 0, 0);
        PTTS_PROD(2, 0) PTTS_PROD(1, 1) PTTS_PROD(0, 2) PTTS_PROD(1, 0) PTTS_PROD(0, 1) PTTS_PROD(0, 0)
#undef PTTS_PROD
        // the 18 fragment reads of the next step spread between the 48 MFMAs of this one (all reads issued first, the
        // eight waves queue 144 KB on the LDS before any of them starts its MFMAs)
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        ++s;
        if (++tap == KW) { tap = 0; ++cb; }
    };
    while (s < nsteps) {
        step(fa0, fb0, fa1, fb1);
        if (s < nsteps) step(fa1, fb1, fa0, fb0);
    }
    __syncthreads();

    // epilogue: C/D layout of the 16x16 MFMA: col = lane&15, row = 4 (lane>>4) + reg
    const bool whole = (cb0 == 0) && (cb1 == g.cblocks);
    const bool add_bias = g.bias != nullptr && cb0 == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn + 16 * j + li;
            const float bv = add_bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + 16 * i + 4 * lg + r;
                if (m < g.M) {
                    float* cp = g.C + (long long)m * g.N + n;
                    const float v = acc[i][j][r] + bv;
                    if (whole) *cp = v; else atomicAdd(cp, v);
                }
            }
        }
}

__global__ __launch_bounds__(S_THREADS) void gemm_bf16x6_kernel(SplitGemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) u16 s_lds[];
    long long it = g.iters_total * blockIdx.x / g.workers;
    const long long it_end = g.iters_total * (blockIdx.x + 1) / g.workers;
    while (it < it_end) {
        const int tile = (int)(it / g.cblocks);
        const int cb0 = (int)(it - (long long)tile * g.cblocks);
        const int cb1 = (int)min((long long)g.cblocks, cb0 + (it_end - it));
        it += cb1 - cb0;
        split_segment(g, s_lds, tile, cb0, cb1);
    }
}

// ONE product per position (ptts_set_bf16_products, BASELINE configs[2]): plane 1 of each operand alone, fp32 accumulation.
// The structure of split_segment is kept, with three consecutive TAPS where the three planes were: a k-step is (channel block,
// taps 3 t3 .. 3 t3 + 2), its B stage holds the three taps' kernel slices (the taps of a channel block lie next to each other in
// the transposed kernel plane), its A fragments are three row shifts of the one frame image, and the three products
// (tap 0, 1, 2) take the place of the six split products: a sixth of the MFMAs, a third of the barriers.  KW % 3 == 0, KW >= 6.
__device__ __forceinline__ void split_segment_1p(const SplitGemmArgs& g, u16* lds, int tile, int cb0, int cb1) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 64;
    const int li = lane & 15, lg = lane >> 4;
    const int m0 = (tile / g.tiles_n) * SBM, n0 = (tile % g.tiles_n) * SBN;
    const int KW = g.taps, KW3 = KW / 3;
    u16* aimg = lds;                                // 2 x A image (plane 0 used)
    u16* bst = lds + 2 * S_AIMG;                    // S_STAGES x B stage (three taps x 128 rows)

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int prow0 = s_prow(g, m0);
    constexpr int A_INSTR = S_AROWS / 16;           // 11 wave-instructions of 16 rows
    constexpr int A_PER_WAVE = (A_INSTR + S_WAVES - 1) / S_WAVES;   // 2 (waves 0..2; the others issue 1)
    const int n_a = (A_INSTR - wave + S_WAVES - 1) / S_WAVES;
    auto issue_a = [&](int cb, int buf) {
        u16* img = aimg + buf * S_AIMG;
#pragma unroll
        for (int x = 0; x < A_PER_WAVE; ++x) {
            const int u = wave + S_WAVES * x;
            if (u < A_INSTR) {
                const int rb = u * 16;
                const int r = rb + (lane >> 2);
                const int cg = (lane & 3) ^ ((r >> 1) & 2);
                long long row = prow0 + r;
                if (row >= g.a_rows) row = g.a_rows - 1;
                dma16h(g.A[0] + ((long long)cb * g.a_rows + row) * SBK + 8 * cg, img + rb * SBK);
            }
        }
    };
    long long boff;
    {
        const int r = 16 * wave + (lane >> 2);
        const int cg = (lane & 3) ^ ((r >> 1) & 2);
        boff = (long long)(n0 + r) * (KW * SBK) + 8 * cg;
    }
    auto issue_b = [&](int cb, int t3, int stage) {        // 3 DMA wave-instructions: taps 3 t3, + 1, + 2
        u16* st = bst + stage * S_BSTAGE;
        const long long k0 = ((long long)cb * g.N * KW + 3 * t3) * SBK;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            dma16h(g.Bt[0] + boff + k0 + p * SBK, st + p * S_IMG + 16 * wave * SBK);
    };
    int arow[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int m = min(m0 + wm + 16 * f + li, g.M - 1);
        arow[f] = s_prow(g, m) - prow0;
    }
    const int bfrag = (wn + li) * SBK + 8 * (lg ^ ((li >> 1) & 2));

    const int nsteps = (cb1 - cb0) * KW3;
    auto step_pos = [&](int s, int& c, int& t) { c = cb0 + s / KW3; t = s - (s / KW3) * KW3; };
    auto read_frags = [&](int s, int c, int t3, bf16x8 (&a)[3][2], bf16x8 (&b)[3][4]) {
        const u16* img = aimg + ((c - cb0) & 1) * S_AIMG;
        const u16* st = bst + (s % S_STAGES) * S_BSTAGE;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int row = arow[f] + 3 * t3 + p;
                a[p][f] = *reinterpret_cast<const bf16x8*>(img + row * SBK + 8 * (lg ^ ((row >> 1) & 2)));
            }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int f = 0; f < 4; ++f) b[p][f] = *reinterpret_cast<const bf16x8*>(st + p * S_IMG + bfrag + 16 * f * SBK);
    };
    issue_a(cb0, 0);
    {
        int c, t;
        step_pos(0, c, t); issue_b(c, t, 0);
        if (nsteps > 1) { step_pos(1, c, t); issue_b(c, t, 1); }
        if (nsteps > 2) { step_pos(2, c, t); issue_b(c, t, 2); }
    }
    if (nsteps > 2) __builtin_amdgcn_s_waitcnt(0xF76);            // vmcnt(6): A(cb0) and B(0) have landed
    else if (nsteps > 1) __builtin_amdgcn_s_waitcnt(0xF73);
    else __builtin_amdgcn_s_waitcnt(0xF70);
    __syncthreads();
    bf16x8 fa0[3][2], fb0[3][4], fa1[3][2], fb1[3][4];
    read_frags(0, cb0, 0, fa0, fb0);
    int s = 0, cb = cb0, tap = 0;
    auto step = [&](bf16x8 (&a)[3][2], bf16x8 (&b)[3][4], bf16x8 (&na)[3][2], bf16x8 (&nb)[3][4]) {
        if (s + 1 < nsteps) {
            const bool b_next = s + 2 < nsteps;
            const bool a_after = (tap == 1) && (cb + 1 < cb1) && s >= 1;
            if (b_next && a_after) { if (n_a == 2) __builtin_amdgcn_s_waitcnt(0xF75); else __builtin_amdgcn_s_waitcnt(0xF74); }   // vmcnt(3 + n_a)
            else if (b_next) __builtin_amdgcn_s_waitcnt(0xF73);
            else __builtin_amdgcn_s_waitcnt(0xF70);
        }
        __syncthreads();
        if (tap == 0 && cb + 1 < cb1) issue_a(cb + 1, ((cb + 1 - cb0) & 1));
        if (s + 3 < nsteps) { int c, t; step_pos(s + 3, c, t); issue_b(c, t, s % S_STAGES); }
        {
            const int sn = min(s + 1, nsteps - 1);
            int c, t; step_pos(sn, c, t); read_frags(sn, c, t, na, nb);
        }
#define PTTS_PROD1(P)                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                       \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                       \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[P][i], b[P][j], acc[i][j], 0, 0, 0);
        PTTS_PROD1(0) PTTS_PROD1(1) PTTS_PROD1(2)
#undef PTTS_PROD1
        // the 18 fragment reads of the next step spread between the 24 MFMAs of this one
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        ++s;
        if (++tap == KW3) { tap = 0; ++cb; }
    };
    while (s < nsteps) {
        step(fa0, fb0, fa1, fb1);
        if (s < nsteps) step(fa1, fb1, fa0, fb0);
    }
    __syncthreads();

    const bool whole = (cb0 == 0) && (cb1 == g.cblocks);
    const bool add_bias = g.bias != nullptr && cb0 == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn + 16 * j + li;
            const float bv = add_bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + 16 * i + 4 * lg + r;
                if (m < g.M) {
                    float* cp = g.C + (long long)m * g.N + n;
                    const float v = acc[i][j][r] + bv;
                    if (whole) *cp = v; else atomicAdd(cp, v);
                }
            }
        }
}

__global__ __launch_bounds__(S_THREADS) void gemm_bf16x1_kernel(SplitGemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) u16 s_lds[];
    long long it = g.iters_total * blockIdx.x / g.workers;
    const long long it_end = g.iters_total * (blockIdx.x + 1) / g.workers;
    while (it < it_end) {
        const int tile = (int)(it / g.cblocks);
        const int cb0 = (int)(it - (long long)tile * g.cblocks);
        const int cb1 = (int)min((long long)g.cblocks, cb0 + (it_end - it));
        it += cb1 - cb0;
        split_segment_1p(g, s_lds, tile, cb0, cb1);
    }
}

static int split_grid(long long total, int threads) {
    long long b = (total + threads - 1) / threads;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace ptts

using namespace ptts;

extern "C" int ptts_split3_frames(const float* x, void* p1, void* p2, void* p3, int B, int T, int C, int pad_left,
                                  int pad_right, int Cp, void* stream) {
    PTTS_REQUIRE(x && p1 && p2 && p3, "split3_frames: null pointer");
    PTTS_REQUIRE(B > 0 && T > 0 && C > 0 && pad_left >= 0 && pad_right >= 0, "split3_frames: bad dims");
    PTTS_REQUIRE(Cp >= C && Cp % SBK == 0, "split3_frames: Cp=%d must be a multiple of %d and >= C=%d", Cp, SBK, C);
    const long long total = (long long)B * (T + pad_left + pad_right) * (Cp / 8);
    hipLaunchKernelGGL(split3_frames_kernel, dim3(split_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, (u16*)p1,
                       (u16*)p2, (u16*)p3, B, T, C, pad_left, pad_right, Cp);
    return check_launch("split3_frames");
}

extern "C" int ptts_split3_weight_t(const float* w, void* p1, void* p2, void* p3, int KW, int C, int N, int Cp,
                                    void* stream) {
    PTTS_REQUIRE(w && p1 && p2 && p3, "split3_weight_t: null pointer");
    PTTS_REQUIRE(KW > 0 && C > 0 && N > 0, "split3_weight_t: bad dims");
    PTTS_REQUIRE(Cp >= C && Cp % SBK == 0, "split3_weight_t: Cp=%d must be a multiple of %d and >= C=%d", Cp, SBK, C);
    const long long total = (long long)KW * (Cp / 8) * N;
    hipLaunchKernelGGL(split3_weight_t_kernel, dim3(split_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w, (u16*)p1,
                       (u16*)p2, (u16*)p3, KW, C, N, Cp);
    return check_launch("split3_weight_t");
}

extern "C" int ptts_conv1d_bf16x6(const void* a1, const void* a2, const void* a3, const void* bt1, const void* bt2,
                                 const void* bt3, const float* bias, float* y, int B, int T, int KW, int Cp, int N,
                                 void* stream) {
    PTTS_REQUIRE(a1 && a2 && a3 && bt1 && bt2 && bt3 && y, "conv1d_bf16x6: null pointer");
    PTTS_REQUIRE(B > 0 && T > 0 && KW > 0 && Cp > 0 && N > 0, "conv1d_bf16x6: bad dims");
    PTTS_REQUIRE(N % SBN == 0, "conv1d_bf16x6: N=%d must be a multiple of %d", N, SBN);
    PTTS_REQUIRE(Cp % SBK == 0, "conv1d_bf16x6: Cp=%d must be a multiple of %d", Cp, SBK);
    // the LDS image of a tile's frames: 128 rows + the halo of the taps + the pad rows of one utterance boundary
    PTTS_REQUIRE((T >= SBM || B == 1) && SBM - 1 + 2 * (KW - 1) < S_AROWS,
                 "conv1d_bf16x6: T=%d, KW=%d do not fit the %d-row frame image (needs T >= %d, KW <= %d)", T, KW, S_AROWS,
                 SBM, (S_AROWS - SBM) / 2 + 1);
    PTTS_REQUIRE((long long)B * T < (1LL << 31) && (long long)B * (T + KW - 1) < (1LL << 31), "conv1d_bf16x6: too many frames");
    PTTS_REQUIRE((((size_t)a1 | (size_t)a2 | (size_t)a3 | (size_t)bt1 | (size_t)bt2 | (size_t)bt3) & 15) == 0,
                 "conv1d_bf16x6: planes must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x6_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x1_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES);
        if (e != hipSuccess) { set_error("conv1d_bf16x6: cannot reserve %d B of LDS: %s", S_LDS_BYTES, hipGetErrorString(e)); return PTTS_ELAUNCH; }
        attr_set = true;
    }
    // bf16 products (ptts_set_bf16_products): one product per position; the three-taps-per-step kernel wants KW % 3 == 0, KW >= 6
    const bool one = ptts::bf16_products() && KW % 3 == 0 && KW >= 6 && 3 * (KW / 3 - 1) + 2 + SBM - 1 + (KW - 1) < S_AROWS + KW;
    SplitGemmArgs g;
    g.A[0] = (const u16*)a1; g.A[1] = (const u16*)a2; g.A[2] = (const u16*)a3;
    g.Bt[0] = (const u16*)bt1; g.Bt[1] = (const u16*)bt2; g.Bt[2] = (const u16*)bt3;
    g.bias = bias; g.C = y; g.M = B * T; g.N = N;
    g.rows_per_seg = T; g.seg_rows = T + KW - 1; g.a_rows = (long long)B * (T + KW - 1);
    g.tiles_n = N / SBN; g.taps = KW; g.cblocks = Cp / SBK;
    const int tiles = ((g.M + SBM - 1) / SBM) * g.tiles_n;
    g.iters_total = (long long)tiles * g.cblocks;
    static int workers_cfg = -1;
    if (workers_cfg < 0) { const char* e = getenv("PTTS_SPLIT_WORKERS"); workers_cfg = e ? atoi(e) : 256; }
    long long workers = workers_cfg;                    // one workgroup per CU (138 KB of LDS each)
    if (workers > g.iters_total) workers = g.iters_total;
    g.workers = (int)workers;
    // partial tiles are combined with atomics into a zeroed y
    if (g.iters_total % workers != 0 || (g.iters_total / workers) % g.cblocks != 0) {
        if (zero_f32(y, (size_t)g.M * N, st) != PTTS_OK) return PTTS_ELAUNCH;
    }
    if (one) hipLaunchKernelGGL(gemm_bf16x1_kernel, dim3(g.workers), dim3(S_THREADS), S_LDS_BYTES, st, g);
    else hipLaunchKernelGGL(gemm_bf16x6_kernel, dim3(g.workers), dim3(S_THREADS), S_LDS_BYTES, st, g);
    return check_launch("conv1d_bf16x6");
}

namespace ptts {

// ================================================================================================
// Weight gradient of the context Conv1D as a bf16x6 split product.
//   dW[j][c][n] = sum_{b,t} Xp[b][t + j][c] dY[b][t][n]  =  sum_q Xt[c][q + j] dYt[n][q],   q = b (T + KW - 1) + t
// with FRAME-MAJOR planes (the reduction index q is the contiguous one, as the MFMA operands want it):
//   Xt [Crows][Pp]: the zero-padded frames, transposed;  dYt [N][Pp]: the incoming gradient at the frame positions of
//   the padded buffer, zero in its pad rows.  Both are written by split3_frames_t_kernel.
// A workgroup (8 waves) owns 64 channels x 32 outputs x ALL KW taps over a slice of q; wave (cf, nf) owns 16 channels x
// 16 outputs: KW accumulator blocks.  Per q-step of 32 a lane reads its 64 bytes of Xt (32 consecutive frames from
// 8 (lane>>4)) ONCE and forms the operand of every tap from them in registers: tap j starts j elements further, i.e. at
// dword j/2 (even j: a register choice) or across two dwords (odd j: one v_alignbyte_b32 per dword).  The B operand (dYt)
// is the same for all taps.  LDS reads per MFMA: 15 ds_read_b128 per 6 KW MFMAs -- the loop is MFMA-paced.
// ================================================================================================
constexpr int W_THREADS = 512;                          // 8 waves
constexpr int W_CB = 64, W_NB = 32, W_QS = 32;           // channels, outputs per workgroup; frames per step
constexpr int W_AROW = 64;                               // frames of an Xt row in LDS: 32 + taps (<= 24) + alignment
constexpr int W_ATILE = W_CB * W_AROW;                   // elements of one plane of the Xt tile: 8 KB
constexpr int W_BTILE = W_NB * W_QS;                     // dYt tile: 2 KB
constexpr int W_STAGE = 3 * W_ATILE + 3 * W_BTILE;       // 30 KB
constexpr int W_STAGES = 3;
constexpr int W_LDS_BYTES = W_STAGES * W_STAGE * 2;      // 90 KB

struct WgradSplitArgs {
    const u16* Xt[3]; const u16* Yt[3]; float* dW;
    int C, N, Crows;
    long long Pp;                 // row length of the planes (elements)
    int qsteps, nsplit, steps_per_split;
    int tiles_n;
};

// fp32 [B][T][C] -> three bf16 planes [Crows][Pp], element (c, b Tp + pad_left + t); everything else zero.
__global__ __launch_bounds__(256) void split3_frames_t_kernel(const float* __restrict__ x, u16* __restrict__ p1,
                                                              u16* __restrict__ p2, u16* __restrict__ p3, int B, int T,
                                                              int C, int pad_left, int Tp, int Crows, long long Pp) {
    __shared__ u16 s[3][32][64 + 2];
    const int tid = threadIdx.x;
    const long long q0 = (long long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 32;
    {
        const int tx = tid & 31, ty = tid >> 5;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ql = ty + 8 * i;
            const long long q = q0 + ql;
            const int b = (int)(q / Tp), t = (int)(q - (long long)b * Tp) - pad_left;
            float v = 0.f;
            if (b < B && t >= 0 && t < T && c0 + tx < C) v = x[((long long)b * T + t) * C + c0 + tx];
            u16 h1, h2, h3;
            split3(v, h1, h2, h3);
            s[0][tx][ql] = h1; s[1][tx][ql] = h2; s[2][tx][ql] = h3;
        }
    }
    __syncthreads();
    {
        const int c = tid >> 3, g8 = tid & 7;
        if (c0 + c < Crows && q0 + 8 * g8 + 7 < Pp) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                U16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o.v[e] = s[p][c][8 * g8 + e];
                u16* dst = (p == 0 ? p1 : (p == 1 ? p2 : p3)) + (long long)(c0 + c) * Pp + q0 + 8 * g8;
                *reinterpret_cast<U16x8*>(dst) = o;
            }
        }
    }
}

// NPL = 3: the six products of the fp32 split.  NPL = 1 (ptts_set_bf16_products, BASELINE configs[2]): plane 1 of each operand
// alone (its bf16 rounding), ONE product per tap, fp32 accumulation.
template <int KW, int NPL>
__global__ __launch_bounds__(W_THREADS) void wgrad_bf16x6_kernel(WgradSplitArgs g) {
    extern __shared__ __attribute__((aligned(16))) u16 s_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cf = wave >> 1, nf = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int tile = blockIdx.x / g.nsplit, split = blockIdx.x - tile * g.nsplit;
    const int c0 = (tile / g.tiles_n) * W_CB, n0 = (tile % g.tiles_n) * W_NB;
    const int s_begin = split * g.steps_per_split;
    const int s_end = min(g.qsteps, s_begin + g.steps_per_split);
    const int nsteps = s_end - s_begin;
    if (nsteps <= 0) return;

    f32x4 acc[KW];
#pragma unroll
    for (int j = 0; j < KW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // DMA sources.  Xt tile: plane p, rows 8 wave .. 8 wave + 7 (a wave-instruction = 8 rows x 128 B): lane -> row
    // lane>>3, physical 16-byte unit lane&7 holding logical unit (lane&7) ^ (row&7).  dYt tile (waves 0..5): plane wave>>1,
    // rows 16 (wave&1) .. +15 (16 rows x 64 B): lane -> row lane>>2, slot lane&3 holding quad (lane&3) ^ ((row>>1)&2).
    long long asrc;
    {
        const int r = 8 * wave + (lane >> 3);
        const int u = (lane & 7) ^ (r & 7);
        asrc = (long long)(c0 + r) * g.Pp + 8 * u;
    }
    long long bsrc = 0;
    const int bplane = wave >> 1;
    if (wave < 2 * NPL) {
        const int r = 16 * (wave & 1) + (lane >> 2);
        const int qd = (lane & 3) ^ ((r >> 1) & 2);
        bsrc = (long long)(n0 + r) * g.Pp + 8 * qd;
    }
    auto issue = [&](int s, int stage) {
        u16* st = s_lds + stage * W_STAGE;
        const long long q0 = (long long)s * W_QS;
#pragma unroll
        for (int p = 0; p < NPL; ++p) dma16h(g.Xt[p] + asrc + q0, st + p * W_ATILE + 8 * wave * W_AROW);
        if (wave < 2 * NPL) {
            const u16* yp = bplane == 0 ? g.Yt[0] : (bplane == 1 ? g.Yt[1] : g.Yt[2]);
            dma16h(yp + bsrc + q0, st + 3 * W_ATILE + bplane * W_BTILE + 16 * (wave & 1) * W_QS);
        }
    };
    // fragment addresses: Xt row 16 cf + li, logical units lg .. lg + 3;  dYt row 16 nf + li, quad lg
    const int arow = 16 * cf + li;
    const int abase = arow * W_AROW;
    const int asw = arow & 7;
    const int brow = 16 * nf + li;
    const int boff = brow * W_QS + 8 * (lg ^ ((brow >> 1) & 2));

    issue(s_begin, 0);
    if (nsteps > 1) issue(s_begin + 1, 1);
    for (int s = 0; s < nsteps; ++s) {
        // DMAs of this wave issued after those of step s: the ones of step s+1 (4 for waves 0..5, 3 for waves 6, 7)
        if (s + 1 < nsteps) {
            if (NPL == 3) { if (wave < 6) __builtin_amdgcn_s_waitcnt(0xF74); else __builtin_amdgcn_s_waitcnt(0xF73); }
            else { if (wave < 2) __builtin_amdgcn_s_waitcnt(0xF72); else __builtin_amdgcn_s_waitcnt(0xF71); }
        } else __builtin_amdgcn_s_waitcnt(0xF70);
        __syncthreads();
        if (s + 2 < nsteps) issue(s_begin + s + 2, (s + 2) % W_STAGES);
        const u16* st = s_lds + (s % W_STAGES) * W_STAGE;
        // 32 consecutive frames of this lane's channel row, per plane: 16 dwords
        unsigned d[NPL][16];
        bf16x8 bq[NPL];
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint4 v = *reinterpret_cast<const uint4*>(st + p * W_ATILE + abase + 8 * ((lg + u) ^ asw));
                d[p][4 * u] = v.x; d[p][4 * u + 1] = v.y; d[p][4 * u + 2] = v.z; d[p][4 * u + 3] = v.w;
            }
            bq[p] = *reinterpret_cast<const bf16x8*>(st + 3 * W_ATILE + p * W_BTILE + boff);
        }
#pragma unroll
        for (int j = 0; j < KW; ++j) {
            bf16x8 a[NPL];
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                unsigned w4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    w4[k] = (j & 1) ? __builtin_amdgcn_alignbyte(d[p][j / 2 + k + 1], d[p][j / 2 + k], 2) : d[p][j / 2 + k];
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 t = {w4[0], w4[1], w4[2], w4[3]};
                a[p] = __builtin_bit_cast(bf16x8, t);
            }
            if (NPL == 3) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[NPL - 1], bq[0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[NPL / 2], bq[NPL / 2], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bq[NPL - 1], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[NPL / 2], bq[0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bq[NPL / 2], acc[j], 0, 0, 0);
            }
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bq[0], acc[j], 0, 0, 0);
        }
    }
    // C/D layout: col = lane&15 -> n, row = 4 (lane>>4) + reg -> c
    const int n = n0 + 16 * nf + li;
#pragma unroll
    for (int j = 0; j < KW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = c0 + 16 * cf + 4 * lg + r;
            if (c < g.C && n < g.N) atomicAdd(g.dW + ((long long)j * g.C + c) * g.N + n, acc[j][r]);
        }
}

}  // namespace ptts

extern "C" int ptts_split3_frames_t(const float* x, void* p1, void* p2, void* p3, int B, int T, int C, int pad_left,
                                    int Tp, int Crows, long long Pp, void* stream) {
    PTTS_REQUIRE(x && p1 && p2 && p3, "split3_frames_t: null pointer");
    PTTS_REQUIRE(B > 0 && T > 0 && C > 0 && pad_left >= 0 && Tp >= T + pad_left, "split3_frames_t: bad dims");
    PTTS_REQUIRE(Crows >= C && Crows % 32 == 0 && Pp % 64 == 0 && Pp >= (long long)B * Tp,
                 "split3_frames_t: Crows=%d must be a multiple of 32 >= C, Pp=%lld a multiple of 64 >= B*Tp", Crows, Pp);
    dim3 grid((unsigned)(Pp / 64), (unsigned)(Crows / 32));
    hipLaunchKernelGGL(split3_frames_t_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (u16*)p1, (u16*)p2, (u16*)p3, B, T, C,
                       pad_left, Tp, Crows, Pp);
    return check_launch("split3_frames_t");
}

extern "C" int ptts_conv1d_wgrad_bf16x6(const void* xt1, const void* xt2, const void* xt3, const void* yt1, const void* yt2,
                                        const void* yt3, float* dw, int B, int T, int KW, int C, int N, int Crows,
                                        long long Pp, void* stream) {
    PTTS_REQUIRE(xt1 && xt2 && xt3 && yt1 && yt2 && yt3 && dw, "conv1d_wgrad_bf16x6: null pointer");
    PTTS_REQUIRE(B > 0 && T > 0 && C > 0 && N > 0, "conv1d_wgrad_bf16x6: bad dims");
    PTTS_REQUIRE(KW == 21 || KW == 5 || KW == 3, "conv1d_wgrad_bf16x6: KW=%d is not instantiated (3, 5, 21)", KW);
    PTTS_REQUIRE(N % W_NB == 0 && Crows % W_CB == 0 && Crows >= C, "conv1d_wgrad_bf16x6: N %% 32 == 0, Crows %% 64 == 0 >= C");
    const long long frames = (long long)B * (T + KW - 1);
    const int qsteps = (int)((frames + W_QS - 1) / W_QS);
    PTTS_REQUIRE(Pp % 64 == 0 && Pp >= (long long)qsteps * W_QS + W_AROW, "conv1d_wgrad_bf16x6: Pp=%lld too short (needs %lld)", Pp,
                 (long long)qsteps * W_QS + W_AROW);
    PTTS_REQUIRE((((size_t)xt1 | (size_t)xt2 | (size_t)xt3 | (size_t)yt1 | (size_t)yt2 | (size_t)yt3) & 15) == 0,
                 "conv1d_wgrad_bf16x6: planes must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    WgradSplitArgs g;
    g.Xt[0] = (const u16*)xt1; g.Xt[1] = (const u16*)xt2; g.Xt[2] = (const u16*)xt3;
    g.Yt[0] = (const u16*)yt1; g.Yt[1] = (const u16*)yt2; g.Yt[2] = (const u16*)yt3;
    g.dW = dw; g.C = C; g.N = N; g.Crows = Crows; g.Pp = Pp; g.qsteps = qsteps;
    g.tiles_n = N / W_NB;
    const int tiles = (Crows / W_CB) * g.tiles_n;
    int nsplit = 256 / tiles;                              // about one workgroup per CU
    if (nsplit < 1) nsplit = 1;
    if (nsplit > qsteps) nsplit = qsteps;
    g.nsplit = nsplit; g.steps_per_split = (qsteps + nsplit - 1) / nsplit;
    if (zero_f32(dw, (size_t)KW * C * N, st) != PTTS_OK) return PTTS_ELAUNCH;
    const bool one = ptts::bf16_products();
#define PTTS_WG_LAUNCH(KWV)                                                                                               \
    {                                                                                                                     \
        static bool attr = false;                                                                                         \
        if (!attr) {                                                                                                      \
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_bf16x6_kernel<KWV, 3>),                \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS_BYTES);                 \
            if (e2 == hipSuccess) e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_bf16x6_kernel<KWV, 1>),     \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS_BYTES);                 \
            if (e2 != hipSuccess) { set_error("conv1d_wgrad_bf16x6: LDS attribute: %s", hipGetErrorString(e2)); return PTTS_ELAUNCH; } \
            attr = true;                                                                                                  \
        }                                                                                                                 \
        if (one) hipLaunchKernelGGL((wgrad_bf16x6_kernel<KWV, 1>), dim3(tiles * nsplit), dim3(W_THREADS), W_LDS_BYTES, st, g); \
        else hipLaunchKernelGGL((wgrad_bf16x6_kernel<KWV, 3>), dim3(tiles * nsplit), dim3(W_THREADS), W_LDS_BYTES, st, g); \
    }
    if (KW == 21) PTTS_WG_LAUNCH(21) else if (KW == 5) PTTS_WG_LAUNCH(5) else PTTS_WG_LAUNCH(3)
#undef PTTS_WG_LAUNCH
    return check_launch("conv1d_wgrad_bf16x6");
}

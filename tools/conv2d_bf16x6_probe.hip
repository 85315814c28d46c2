// Prototype (measurement only, not in the library): the critic's 4 -> 4 channel 5x5 'same' NHWC convolution FORWARD as a
// bf16x6 split product on the bf16 matrix cores, to price the next step of DESIGN.md section 7.
//   y[b,t,f,co] = bias[co] + sum_{kt,kf,ci} lrelu(x[b,t+kt-2,f+kf-2,ci]) w[kt,kf,ci,co]
// MFMA mapping (v_mfma_f32_16x16x32_bf16): M = 16 time rows, N = 16 = 4 output bins x 4 co, K = 32 = 8 input bins x 4 ci
// per kernel row kt: the 5x4 taps of a kernel row form a banded (Toeplitz) 32x16 block with 20 of 32 rows used per
// column (62.5 % of the MFMA flops are useful), held in registers for the whole kernel.  The A operand of a lane is 8
// consecutive bf16 of the NHWC row (2 bins x 4 channels) -- no im2col, no shuffles.  The tile's input is activated and
// split into its three bf16 planes ONCE, on the way into LDS.
// build: hipcc -O3 --offload-arch=gfx950 [-DTR_=16|32 -DTHREADS_=256|512 -DNO_STAGE -DNO_MFMA] tools/conv2d_bf16x6_probe.hip -o tools/conv2d_bf16x6_probe
// Measured on MI355X, [64,400,65,4], against 30 us for the library's packed-FMA kernel (all bit-for-bit fp32-level: max error 1.7e-6 of
// mean |y| against fp64): this unpipelined version 23.5 us (TR 32 or 16, 512 threads); its staging phase
// alone (load + activation + split + LDS write + the output stores, -DNO_MFMA) 8.0 us = 6.7 TB/s; its MFMA phase alone 6.1 us without
// (-DNO_STAGE -DNO_STORE) and 11.3 us with the output stores (-DNO_STAGE); staging + MFMA without stores (-DNO_STORE) 19.4 us.  The phases
// run one after the other inside a workgroup there.  conv2d_bf16x6_fwd_pipe below (persistent, 9 MFMA + 7 staging waves, double-buffered
// stage) is bit-identical and measured 21.8 us: one workgroup per CU makes both roles slower (-DPIPE_NO_STAGE 19.3 us, -DPIPE_NO_MFMA 14.0 us).
// conv2d_bf16x6_fwd_v3 (persistent, 8 waves doing both jobs, loads two tiles ahead in registers, two workgroups per CU): 20.3 us.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

constexpr int F = 65, C = 4, KT = 5, KF = 5;
#ifndef TR_
#define TR_ 32
#endif
#ifndef THREADS_
#define THREADS_ 512
#endif
constexpr int TR = TR_;                   // time rows per workgroup
constexpr int ROWS = TR + KT - 1;         // staged rows
constexpr int FG = 17;                    // groups of 4 output bins
constexpr int BINS = 72;                  // staged bins: -2 .. 69
constexpr int RSTRIDE = BINS * C + 8;     // elements per staged row (592 B: 5 sixteen-byte units mod 16 -> rows spread over the banks)
constexpr int PLANE = ROWS * RSTRIDE;
constexpr int THREADS = THREADS_;
constexpr int NWAVES = THREADS / 64;
constexpr int RP = THREADS / BINS;         // staged rows per pass

__device__ __forceinline__ u16 bf16_rn(float f) {
    const unsigned u = __float_as_uint(f);
    return (u16)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_f(u16 h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void split3(float x, u16& h1, u16& h2, u16& h3) {
    h1 = bf16_rn(x); const float r1 = x - bf16_f(h1);
    h2 = bf16_rn(r1); const float r2 = r1 - bf16_f(h2);
    h3 = bf16_rn(r2);
}

// the banded (Toeplitz) kernel blocks as the MFMA's B operand, three bf16 planes: [KT][3][64 lanes][8], made once per weight update
__global__ void toeplitz_kernel(const float* __restrict__ w, u16* __restrict__ tab) {
    const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4, so = li >> 2, co = li & 3;
    for (int kt = 0; kt < KT; ++kt)
        for (int e = 0; e < 8; ++e) {
            const int j = 2 * lg + (e >> 2), ci = e & 3, kf = j - so;
            const float v = (kf >= 0 && kf < KF) ? w[((kt * KF + kf) * C + ci) * C + co] : 0.f;
            u16 h1, h2, h3;
            split3(v, h1, h2, h3);
            tab[((kt * 3 + 0) * 64 + lane) * 8 + e] = h1; tab[((kt * 3 + 1) * 64 + lane) * 8 + e] = h2; tab[((kt * 3 + 2) * 64 + lane) * 8 + e] = h3;
        }
}

__global__ __launch_bounds__(THREADS) void conv2d_bf16x6_fwd(const float* __restrict__ x, const u16* __restrict__ wtab,
                                                             const float* __restrict__ bias, float* __restrict__ y, int B,
                                                             int T, float alpha) {
    __shared__ __attribute__((aligned(16))) u16 s[3 * PLANE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int tiles_t = (T + TR - 1) / TR;
    const int b = blockIdx.x / tiles_t, t0 = (blockIdx.x % tiles_t) * TR;

    // Toeplitz blocks of the kernel (B operand): lane (n = li -> (s = li>>2, co = li&3), lg) holds k = 8 lg + e -> input bin
    // j = 2 lg + e/4, ci = e&3:  B[k][n] = w[kt][j - s][ci][co] when 0 <= j - s < 5
    bf16x8 wb[KT][3];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int p = 0; p < 3; ++p) wb[kt][p] = *reinterpret_cast<const bf16x8*>(wtab + ((kt * 3 + p) * 64 + lane) * 8);
    // stage: rows t0-2 .. t0+TR+1, bins -2 .. 69 (zero outside the image), activated and split once.  The hardware's packed
    // conversion (v_cvt_pk_bf16_f32, round to nearest even) does the rounding: 11 vector instructions per pair of values
#ifndef NO_STAGE
    {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        const int bin = tid % BINS, r0 = tid / BINS;                 // RP rows of 72 bins per pass
        const int f = bin - 2;
        // (issuing all of a thread's loads before the arithmetic measured slower: 34 vs 23.5 us -- register pressure)
        if (r0 < RP)
            for (int r = r0; r < ROWS; r += RP) {
                const int t = t0 + r - 2;
                f32x4v v = {0.f, 0.f, 0.f, 0.f};
                if (t >= 0 && t < T && f >= 0 && f < F) {
                    v = *reinterpret_cast<const f32x4v*>(x + (((long long)b * T + t) * F + f) * C);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : alpha * v[e];
                }
                const bf16x4 h1 = __builtin_convertvector(v, bf16x4);
                const f32x4v r1 = v - __builtin_convertvector(h1, f32x4v);
                const bf16x4 h2 = __builtin_convertvector(r1, bf16x4);
                const f32x4v r2 = r1 - __builtin_convertvector(h2, f32x4v);
                const bf16x4 h3 = __builtin_convertvector(r2, bf16x4);
                u16* dst = s + r * RSTRIDE + bin * C;
                *reinterpret_cast<bf16x4*>(dst) = h1;
                *reinterpret_cast<bf16x4*>(dst + PLANE) = h2;
                *reinterpret_cast<bf16x4*>(dst + 2 * PLANE) = h3;
            }
    }
#endif
    __syncthreads();
    const float bv = bias[li & 3];
    // wave-tiles: (time block of 16 rows, bin group g): 2 x 17 = 34 over 8 waves
    for (int wt = wave; wt < (TR / 16) * FG; wt += 2 * NWAVES) {
        // two wave-tiles at a time: two independent accumulator chains
        const int wt2 = wt + NWAVES;
        const bool two = wt2 < (TR / 16) * FG;
        const int tb = wt / FG, g = wt - tb * FG;
        const int tb2 = two ? wt2 / FG : tb, g2 = two ? wt2 - tb2 * FG : g;
        f32x4 acc = {bv, bv, bv, bv}, acc2 = {bv, bv, bv, bv};
        const int base = (16 * tb + li) * RSTRIDE + (4 * g + 2 * lg) * C;       // row li of the block, staged bins 4g+2lg, +1
        const int base2 = (16 * tb2 + li) * RSTRIDE + (4 * g2 + 2 * lg) * C;
#ifndef NO_MFMA
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            bf16x8 a[3], a2[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[p] = *reinterpret_cast<const bf16x8*>(s + p * PLANE + base + kt * RSTRIDE);
                a2[p] = *reinterpret_cast<const bf16x8*>(s + p * PLANE + base2 + kt * RSTRIDE);
            }
#define P6(PA, PB)                                                                          \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA], wb[kt][PB], acc, 0, 0, 0);     \
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[PA], wb[kt][PB], acc2, 0, 0, 0);
            P6(2, 0) P6(1, 1) P6(0, 2) P6(1, 0) P6(0, 1) P6(0, 0)
#undef P6
        }
#endif
        // D[row = 4 lg + r -> time row][col = li -> (bin 4g + li/4, co li&3)]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 4 * g + (li >> 2), t = t0 + 16 * tb + 4 * lg + r;
#ifdef NO_STORE
            if (acc[r] == 12345.678f || acc2[r] == 12345.678f) y[0] = acc[r] + acc2[r];
#else
            if (t < T && f < F) y[(((long long)b * T + t) * F + f) * C + (li & 3)] = acc[r];
            const int f2 = 4 * g2 + (li >> 2), t2 = t0 + 16 * tb2 + 4 * lg + r;
            if (two && t2 < T && f2 < F) y[(((long long)b * T + t2) * F + f2) * C + (li & 3)] = acc2[r];
#endif
        }
    }
}

// ---- pipelined version: persistent workgroups of 16 waves, 9 MFMA waves + 7 staging waves, double-buffered LDS stage ----
constexpr int PT = 16;                        // time rows per tile (400 = 25 tiles: no tile straddles two utterances)
constexpr int PROWS = PT + KT - 1;            // 20 staged rows
constexpr int PPLANE = PROWS * RSTRIDE;
constexpr int PBUF = 3 * PPLANE;              // one stage: 35.5 KB
constexpr int PTHREADS = 1024;
constexpr int MFMA_WAVES = 9;                 // 17 bin groups: waves 0..7 take two (g, g + 9), wave 8 one

__global__ __launch_bounds__(PTHREADS) void conv2d_bf16x6_fwd_pipe(const float* __restrict__ x, const u16* __restrict__ wtab,
                                                                  const float* __restrict__ bias, float* __restrict__ y, int B,
                                                                  int T, float alpha) {
    extern __shared__ __attribute__((aligned(16))) u16 sp[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int tiles_t = T / PT, ntiles = B * tiles_t;
    const bool mfma_role = wave < MFMA_WAVES;

    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    // staging waves: the loads of a tile are issued one whole iteration before they are converted (registers), so their
    // latency is covered by the MFMA waves' work on the tile in between
    const int stid = tid - MFMA_WAVES * 64;                      // 0 .. 447
    const int sbin = stid % BINS, sr0 = stid / BINS;             // 6 rows of 72 bins per pass, 4 passes
    f32x4v pre[4];
    auto load = [&](int tile) {
        const int b = tile / tiles_t, t0 = (tile - b * tiles_t) * PT;
        const int f = sbin - 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = sr0 + 6 * q, t = t0 + r - 2;
            pre[q] = f32x4v{0.f, 0.f, 0.f, 0.f};
            if (sr0 < 6 && r < PROWS && t >= 0 && t < T && f >= 0 && f < F)
                pre[q] = *reinterpret_cast<const f32x4v*>(x + (((long long)b * T + t) * F + f) * C);
        }
    };
    auto convert = [&](u16* s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = sr0 + 6 * q;
            if (sr0 < 6 && r < PROWS) {
                f32x4v v = pre[q];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : alpha * v[e];
                const bf16x4 h1 = __builtin_convertvector(v, bf16x4);
                const f32x4v r1 = v - __builtin_convertvector(h1, f32x4v);
                const bf16x4 h2 = __builtin_convertvector(r1, bf16x4);
                const f32x4v r2 = r1 - __builtin_convertvector(h2, f32x4v);
                const bf16x4 h3 = __builtin_convertvector(r2, bf16x4);
                u16* dst = s + r * RSTRIDE + sbin * C;
                *reinterpret_cast<bf16x4*>(dst) = h1;
                *reinterpret_cast<bf16x4*>(dst + PPLANE) = h2;
                *reinterpret_cast<bf16x4*>(dst + 2 * PPLANE) = h3;
            }
        }
    };

    bf16x8 wb[KT][3];
    float bv = 0.f;
    if (mfma_role) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int p = 0; p < 3; ++p) wb[kt][p] = *reinterpret_cast<const bf16x8*>(wtab + ((kt * 3 + p) * 64 + lane) * 8);
        bv = bias[li & 3];
    }
    int tile = blockIdx.x, i = 0;
    if (!mfma_role && tile < ntiles) {
        load(tile); convert(sp);
        if (tile + (int)gridDim.x < ntiles) load(tile + gridDim.x);
    }
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x, ++i) {
        if (!mfma_role) {
            const int nxt = tile + gridDim.x;
#ifndef PIPE_NO_STAGE
            if (nxt < ntiles) {
                convert(sp + ((i + 1) & 1) * PBUF);              // the loads of tile `nxt`, issued an iteration ago
                if (nxt + (int)gridDim.x < ntiles) load(nxt + gridDim.x);
            }
#endif
        } else {
            const u16* s = sp + (i & 1) * PBUF;
            const int b = tile / tiles_t, t0 = (tile - b * tiles_t) * PT;
            const int g = wave, g2 = wave + MFMA_WAVES;
            const bool two = g2 < FG;
            f32x4 acc = {bv, bv, bv, bv}, acc2 = {bv, bv, bv, bv};
            const int base = li * RSTRIDE + (4 * g + 2 * lg) * C;
            const int base2 = li * RSTRIDE + (4 * (two ? g2 : g) + 2 * lg) * C;
#ifndef PIPE_NO_MFMA
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                bf16x8 a[3], a2[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[p] = *reinterpret_cast<const bf16x8*>(s + p * PPLANE + base + kt * RSTRIDE);
                    a2[p] = *reinterpret_cast<const bf16x8*>(s + p * PPLANE + base2 + kt * RSTRIDE);
                }
#define P6(PA, PB)                                                                              \
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA], wb[kt][PB], acc, 0, 0, 0);     \
                if (two) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[PA], wb[kt][PB], acc2, 0, 0, 0);
                P6(2, 0) P6(1, 1) P6(0, 2) P6(1, 0) P6(0, 1) P6(0, 0)
#undef P6
            }
#endif
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = t0 + 4 * lg + r;
                const int f = 4 * g + (li >> 2), f2 = 4 * g2 + (li >> 2);
                if (f < F) y[(((long long)b * T + t) * F + f) * C + (li & 3)] = acc[r];
                if (two && f2 < F) y[(((long long)b * T + t) * F + f2) * C + (li & 3)] = acc2[r];
            }
        }
        __syncthreads();
    }
}

// ---- third version: persistent workgroups of 8 waves, every wave does both jobs, software-pipelined over the tiles: the
// global loads of tile i+2 are in flight (registers) while tile i is multiplied and tile i+1 is converted into the other
// stage; 71 KB of LDS -> two workgroups per CU. ----
constexpr int V3_THREADS = 512;
__global__ __launch_bounds__(V3_THREADS) void conv2d_bf16x6_fwd_v3(const float* __restrict__ x, const u16* __restrict__ wtab,
                                                                 const float* __restrict__ bias, float* __restrict__ y, int B,
                                                                 int T, float alpha) {
    extern __shared__ __attribute__((aligned(16))) u16 sp[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int tiles_t = T / PT, ntiles = B * tiles_t;
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const int sbin = tid % BINS, sr0 = tid / BINS;               // 7 rows of 72 bins per pass (504 threads), 3 passes
    f32x4v pre[3];
    auto load = [&](int tile) {
        const int b = tile / tiles_t, t0 = (tile - b * tiles_t) * PT;
        const int f = sbin - 2;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int r = sr0 + 7 * q, t = t0 + r - 2;
            pre[q] = f32x4v{0.f, 0.f, 0.f, 0.f};
            if (sr0 < 7 && r < PROWS && t >= 0 && t < T && f >= 0 && f < F)
                pre[q] = *reinterpret_cast<const f32x4v*>(x + (((long long)b * T + t) * F + f) * C);
        }
    };
    auto convert = [&](u16* s) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int r = sr0 + 7 * q;
            if (sr0 < 7 && r < PROWS) {
                f32x4v v = pre[q];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : alpha * v[e];
                const bf16x4 h1 = __builtin_convertvector(v, bf16x4);
                const f32x4v r1 = v - __builtin_convertvector(h1, f32x4v);
                const bf16x4 h2 = __builtin_convertvector(r1, bf16x4);
                const f32x4v r2 = r1 - __builtin_convertvector(h2, f32x4v);
                const bf16x4 h3 = __builtin_convertvector(r2, bf16x4);
                u16* dst = s + r * RSTRIDE + sbin * C;
                *reinterpret_cast<bf16x4*>(dst) = h1;
                *reinterpret_cast<bf16x4*>(dst + PPLANE) = h2;
                *reinterpret_cast<bf16x4*>(dst + 2 * PPLANE) = h3;
            }
        }
    };
    bf16x8 wb[KT][3];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int p = 0; p < 3; ++p) wb[kt][p] = *reinterpret_cast<const bf16x8*>(wtab + ((kt * 3 + p) * 64 + lane) * 8);
    const float bv = bias[li & 3];

    int tile = blockIdx.x, i = 0;
    const int step = gridDim.x;
    if (tile < ntiles) { load(tile); convert(sp); }
    if (tile + step < ntiles) load(tile + step);
    __syncthreads();
    for (; tile < ntiles; tile += step, ++i) {
        const u16* s = sp + (i & 1) * PBUF;
        const int b = tile / tiles_t, t0 = (tile - b * tiles_t) * PT;
        // bin groups of this wave: g = wave, wave + 8, (wave 0 only) 16
        for (int g = wave; g < FG; g += 16) {
            const int g2 = g + 8;
            const bool two = g2 < FG;
            f32x4 acc = {bv, bv, bv, bv}, acc2 = {bv, bv, bv, bv};
            const int base = li * RSTRIDE + (4 * g + 2 * lg) * C;
            const int base2 = li * RSTRIDE + (4 * (two ? g2 : g) + 2 * lg) * C;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                bf16x8 a[3], a2[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[p] = *reinterpret_cast<const bf16x8*>(s + p * PPLANE + base + kt * RSTRIDE);
                    a2[p] = *reinterpret_cast<const bf16x8*>(s + p * PPLANE + base2 + kt * RSTRIDE);
                }
#define P6(PA, PB)                                                                              \
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA], wb[kt][PB], acc, 0, 0, 0);     \
                if (two) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[PA], wb[kt][PB], acc2, 0, 0, 0);
                P6(2, 0) P6(1, 1) P6(0, 2) P6(1, 0) P6(0, 1) P6(0, 0)
#undef P6
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = t0 + 4 * lg + r;
                const int f = 4 * g + (li >> 2), f2 = 4 * g2 + (li >> 2);
                if (f < F) y[(((long long)b * T + t) * F + f) * C + (li & 3)] = acc[r];
                if (two && f2 < F) y[(((long long)b * T + t) * F + f2) * C + (li & 3)] = acc2[r];
            }
        }
        // the next tile's registers (loaded an iteration ago) into the other stage, then the loads of the one after
        if (tile + step < ntiles) {
            convert(sp + ((i + 1) & 1) * PBUF);
            if (tile + 2 * step < ntiles) load(tile + 2 * step);
        }
        __syncthreads();
    }
}

int main() {
    const int B = 64, T = 400;
    const size_t n = (size_t)B * T * F * C;
    std::vector<float> hx(n), hw(KT * KF * C * C), hb(C), hy(n);
    unsigned seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((int)(seed >> 9) - (1 << 22)) / (float)(1 << 22); };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.2f;
    for (auto& v : hb) v = rnd();
    float *dx, *dw, *db, *dy;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dw, hw.size() * 4); hipMalloc(&db, 16);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), 16, hipMemcpyHostToDevice);
    u16* dtab; hipMalloc(&dtab, KT * 3 * 64 * 8 * 2);
    hipLaunchKernelGGL(toeplitz_kernel, dim3(1), dim3(64), 0, 0, dw, dtab);
    const int grid = B * ((T + TR - 1) / TR);
    const float alpha = 0.3f;
    hipLaunchKernelGGL(conv2d_bf16x6_fwd, dim3(grid), dim3(THREADS), 0, 0, dx, dtab, db, dy, B, T, alpha);
    hipDeviceSynchronize();
    hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost);
    // fp64 reference on a sample of pixels (corners and interior)
    double emax = 0, ssum = 0; int cnt = 0;
    for (int trial = 0; trial < 4000; ++trial) {
        int b = trial % B, t = (trial * 37) % T, f = (trial * 11) % F;
        if (trial < 8) { b = trial & 1 ? B - 1 : 0; t = trial & 2 ? T - 1 : 0; f = trial & 4 ? F - 1 : 0; }
        for (int co = 0; co < C; ++co) {
            double a = hb[co];
            for (int kt = 0; kt < KT; ++kt)
                for (int kf = 0; kf < KF; ++kf) {
                    const int tt = t + kt - 2, ff = f + kf - 2;
                    if (tt < 0 || tt >= T || ff < 0 || ff >= F) continue;
                    for (int ci = 0; ci < C; ++ci) {
                        float v = hx[(((size_t)b * T + tt) * F + ff) * C + ci];
                        v = v > 0.f ? v : alpha * v;
                        a += (double)v * hw[((kt * KF + kf) * C + ci) * C + co];
                    }
                }
            const double e = std::fabs(a - hy[(((size_t)b * T + t) * F + f) * C + co]);
            emax = e > emax ? e : emax; ssum += std::fabs(a); ++cnt;
        }
    }
    printf("max |err| / mean |ref| = %.3e over %d outputs\n", emax / (ssum / cnt), cnt);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 50;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(conv2d_bf16x6_fwd, dim3(grid), dim3(THREADS), 0, 0, dx, dtab, db, dy, B, T, alpha);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("conv2d 4->4 5x5 forward, [64,400,65,4]: %.1f us per launch; algorithmic %.2f TFLOP/s; %.2f TB/s of the 53.2 MB in + out\n",
           us, 2.0 * B * T * F * C * C * KT * KF / us / 1e6, 2.0 * n * 4 / us / 1e6);
    // pipelined version: same check (against the first kernel's verified output), then timing
    {
        float* dy2; hipMalloc(&dy2, n * 4); hipMemset(dy2, 0, n * 4);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_bf16x6_fwd_pipe), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF * 2);
        int ngrid = 256; if (const char* e = getenv("PIPE_GRID")) ngrid = atoi(e);
        hipLaunchKernelGGL(conv2d_bf16x6_fwd_pipe, dim3(ngrid), dim3(PTHREADS), 2 * PBUF * 2, 0, dx, dtab, db, dy2, B, T, alpha);
        hipError_t err = hipDeviceSynchronize();
        std::vector<float> hy2(n);
        hipMemcpy(hy2.data(), dy2, n * 4, hipMemcpyDeviceToHost);
        double dmax = 0;
        for (size_t k = 0; k < n; ++k) { const double d = std::fabs((double)hy2[k] - hy[k]); dmax = d > dmax ? d : dmax; }
        printf("pipelined kernel: %s, max |y_pipe - y| = %.3e\n", hipGetErrorString(err), dmax);
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(conv2d_bf16x6_fwd_pipe, dim3(ngrid), dim3(PTHREADS), 2 * PBUF * 2, 0, dx, dtab, db, dy2, B, T, alpha);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double us2 = ms * 1e3 / reps;
        printf("pipelined (persistent, 9 MFMA + 7 staging waves, double-buffered): %.1f us per launch; %.2f TB/s of the 53.2 MB = %.2f of 8 TB/s\n",
               us2, 2.0 * n * 4 / us2 / 1e6, 2.0 * n * 4 / us2 / 1e6 / 8.0);
    }
    {
        float* dy3; hipMalloc(&dy3, n * 4); hipMemset(dy3, 0, n * 4);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_bf16x6_fwd_v3), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF * 2);
        int ngrid = 512; if (const char* e = getenv("V3_GRID")) ngrid = atoi(e);
        hipLaunchKernelGGL(conv2d_bf16x6_fwd_v3, dim3(ngrid), dim3(V3_THREADS), 2 * PBUF * 2, 0, dx, dtab, db, dy3, B, T, alpha);
        hipError_t err = hipDeviceSynchronize();
        std::vector<float> hy3(n);
        hipMemcpy(hy3.data(), dy3, n * 4, hipMemcpyDeviceToHost);
        double dmax = 0;
        for (size_t k = 0; k < n; ++k) { const double d = std::fabs((double)hy3[k] - hy[k]); dmax = d > dmax ? d : dmax; }
        printf("v3 kernel: %s, max |y_v3 - y| = %.3e\n", hipGetErrorString(err), dmax);
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(conv2d_bf16x6_fwd_v3, dim3(ngrid), dim3(V3_THREADS), 2 * PBUF * 2, 0, dx, dtab, db, dy3, B, T, alpha);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double us3 = ms * 1e3 / reps;
        printf("v3 (persistent, 8 waves doing both jobs, loads two tiles ahead, grid %d): %.1f us per launch; %.2f TB/s = %.2f of 8 TB/s\n",
               ngrid, us3, 2.0 * n * 4 / us3 / 1e6, 2.0 * n * 4 / us3 / 1e6 / 8.0);
    }
    return 0;
}

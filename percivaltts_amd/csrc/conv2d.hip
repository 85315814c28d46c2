// 2D convolution over (time x frequency) for gfx950: forward and fused backward.
//
// Role on the hot path: the 8-deep 5x5 Conv2D stacks of the critic (reference
// networks_critic.py:66-68) and of the generator's spectral branch (modeltts_common.py:97-100,
// networktts.py:122-126).  Since round 2 the 4 -> 4 channel layers of those stacks run on the bf16 matrix cores
// (conv2d_mfma.hip); this file keeps the 1 -> 4, 4 -> 1 and gated 4 -> 8 layers, every other shape, the backward of
// BatchNorm-fused layers, and stays selectable for the 4 -> 4 layers as the fp32 A/B partner (PTTS_CONV2D_MFMA=0).
// Channel counts are tiny (C = 1..4) so this is a vector-ALU stencil,
// not a GEMM: one lane owns a strip of KF consecutive frequency bins x all output channels,
// the input tile (full frequency width + time halo) is staged once in LDS with the
// BatchNorm-affine/LeakyReLU of the PREVIOUS layer applied on load, weights are wave-uniform
// (scalar loads), HBM is touched exactly once per element per pass.
//
// Backward is one kernel per layer: it reads dy (with halo) and x once, produces dx with the
// LeakyReLU mask of the previous layer already applied, and per-workgroup partial sums of
// dw / dbias / dscale / dshift that a second tiny kernel reduces in a fixed order (ptts_conv2d_bwd:
// deterministic, no float atomics).  Inside the optimiser's steps the partial rows of all layers are added into the
// gradient buffers by ptts_conv2d_reduce_grouped instead: one launch per 16 passes, with fp32 atomics between passes
// that share a buffer -- one descriptor per launch (a fixed order again) when ptts_set_deterministic(1) is in force.
#include "common.h"
#include <cstdlib>

namespace ptts {

template <int C> struct VecIO;
template <> struct VecIO<1> {
    static __device__ __forceinline__ void ld(const float* p, float* v) { v[0] = p[0]; }
    static __device__ __forceinline__ void st(float* p, const float* v) { p[0] = v[0]; }
};
template <> struct VecIO<2> {
    static __device__ __forceinline__ void ld(const float* p, float* v) {
        float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
    static __device__ __forceinline__ void st(float* p, const float* v) {
        *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); }
};
template <> struct VecIO<4> {
    static __device__ __forceinline__ void ld(const float* p, float* v) {
        float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    static __device__ __forceinline__ void st(float* p, const float* v) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};

constexpr int CONV_THREADS = 256;

// v_pk_fma_f32: two fp32 FMAs per lane per issue slot -- the only way to the fp32 vector peak on gfx950.  The
// accumulators are kept as float2 over adjacent OUTPUT channels, the activation is a broadcast (op_sel) and the
// weight pair comes straight from an SGPR pair, so the inner loops compile to bare v_pk_fma_f32 streams.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pkfma(float a, f2 w, f2 c) { return __builtin_elementwise_fma((f2){a, a}, w, c); }

// a = transform(x) for one pixel (CIN channels)
template <int CIN>
__device__ __forceinline__ void in_transform(float* a, const float* __restrict__ in_scale,
                                             const float* __restrict__ in_shift,
                                             const float* __restrict__ mask_src, long long off,
                                             int in_mode, float alpha) {
    if (in_mode == PTTS_IN_LRELU) {
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
            float p = a[c];
            if (in_scale) p = p * in_scale[c] + in_shift[c];
            a[c] = lrelu(p, alpha);
        }
    } else if (in_mode == PTTS_IN_MASKMUL) {
        float m[CIN];
        VecIO<CIN>::ld(mask_src + off, m);
#pragma unroll
        for (int c = 0; c < CIN; ++c) a[c] *= lrelu_d(m[c], alpha);
    }
}


// Stage a [rows x cols] window of an NHWC tensor into LDS (zero outside the image), NB loads in flight per lane
// before the first one is consumed (a load-wait-store loop would expose the HBM latency once per element).
// dst[idx*C..] = transform(src[pixel]);  raw (optional) receives the untransformed values.
template <int C, int NB>
__device__ __forceinline__ void stage_window(const float* __restrict__ src, const float* __restrict__ mask_src,
                                             float* __restrict__ dst, float* __restrict__ raw, long long img,
                                             int T, int F, int rows, int cols, int t_org, int f_org,
                                             const float* __restrict__ in_scale,
                                             const float* __restrict__ in_shift, int in_mode, float alpha) {
    const int total = rows * cols;
    for (int base = threadIdx.x; base < total; base += CONV_THREADS * NB) {
        float v[NB][C], m[NB][C];
        bool ok[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = base + u * CONV_THREADS;
            const int r = idx / cols, c = idx - r * cols;
            const int t = t_org + r, f = f_org + c;
            ok[u] = idx < total && t >= 0 && t < T && f >= 0 && f < F;
#pragma unroll
            for (int c2 = 0; c2 < C; ++c2) { v[u][c2] = 0.f; m[u][c2] = 0.f; }
            if (ok[u]) {
                const long long off = (img + (long long)t * F + f) * C;
                VecIO<C>::ld(src + off, v[u]);
                if (in_mode == PTTS_IN_MASKMUL) VecIO<C>::ld(mask_src + off, m[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = base + u * CONV_THREADS;
            if (idx >= total) continue;
            if (raw) VecIO<C>::st(raw + (size_t)idx * C, v[u]);
            if (ok[u]) {
                if (in_mode == PTTS_IN_LRELU) {
#pragma unroll
                    for (int c2 = 0; c2 < C; ++c2) {
                        float p = v[u][c2];
                        if (in_scale) p = p * in_scale[c2] + in_shift[c2];
                        v[u][c2] = lrelu(p, alpha);
                    }
                } else if (in_mode == PTTS_IN_MASKMUL) {
#pragma unroll
                    for (int c2 = 0; c2 < C; ++c2) v[u][c2] *= lrelu_d(m[u][c2], alpha);
                }
            }
            VecIO<C>::st(dst + (size_t)idx * C, v[u]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// forward
// grid (ceil(T/TT), B), 256 threads.  LDS: [TT + (KT-1)*dil_t][Fp + KF-1][CIN] floats, Fp = ceil(F/KF)*KF
// ------------------------------------------------------------------------------------------
// Persistent, software-pipelined: a workgroup walks tiles blockIdx.x, +gridDim.x, ...; the global loads of tile i+1
// are issued (into registers) before the FMA phase of tile i and land in LDS after it, and the coalesced stores of
// tile i drain while tile i+1 is computed -- without this every workgroup of the (single) wave of workgroups loads,
// computes and stores in lock-step and HBM idles during the FMA phase.
constexpr int NPF = 8;   // prefetch registers: up to 8 pixels per lane per tile (tile rows*cols <= 2048)

template <int CIN>
struct Prefetch {
    float v[NPF][CIN];
    float m[NPF][CIN];
    unsigned okmask;
};

template <int CIN, bool MASK>
__device__ __forceinline__ void prefetch_tile(Prefetch<CIN>& pf, const float* __restrict__ x,
                                              const float* __restrict__ mask_src, long long img, int T, int F,
                                              int rows, int cols, int t_org, int f_org) {
    const int total = rows * cols;
    pf.okmask = 0u;
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
        const int idx = threadIdx.x + u * CONV_THREADS;
        const int r = idx / cols, c = idx - r * cols;
        const int t = t_org + r, f = f_org + c;
        const bool ok = idx < total && t >= 0 && t < T && f >= 0 && f < F;
#pragma unroll
        for (int c2 = 0; c2 < CIN; ++c2) { pf.v[u][c2] = 0.f; if (MASK) pf.m[u][c2] = 0.f; }
        if (ok) {
            const long long off = (img + (long long)t * F + f) * CIN;
            VecIO<CIN>::ld(x + off, pf.v[u]);
            if (MASK) VecIO<CIN>::ld(mask_src + off, pf.m[u]);
            pf.okmask |= 1u << u;
        }
    }
}

template <int CIN, bool MASK>
__device__ __forceinline__ void commit_tile(const Prefetch<CIN>& pf, float* __restrict__ dst, int total,
                                            const float* __restrict__ in_scale,
                                            const float* __restrict__ in_shift, int in_mode, float alpha) {
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
        const int idx = threadIdx.x + u * CONV_THREADS;
        if (idx >= total) continue;
        float a[CIN];
#pragma unroll
        for (int c2 = 0; c2 < CIN; ++c2) a[c2] = pf.v[u][c2];
        if ((pf.okmask >> u) & 1u) {
            if (MASK) {
#pragma unroll
                for (int c2 = 0; c2 < CIN; ++c2) a[c2] *= lrelu_d(pf.m[u][c2], alpha);
            } else if (in_mode == PTTS_IN_LRELU) {
#pragma unroll
                for (int c2 = 0; c2 < CIN; ++c2) {
                    float p = a[c2];
                    if (in_scale) p = p * in_scale[c2] + in_shift[c2];
                    a[c2] = lrelu(p, alpha);
                }
            }
        }
        VecIO<CIN>::st(dst + (size_t)idx * CIN, a);
    }
}

// LDS: in-tile [TT + (KT-1)*dil_t][cols][CIN] | out-tile [TT][F][COUT]   (separate regions: the stores of tile i
// read the out region while the in region already receives tile i+1)
template <int CIN, int COUT, int KT, int KF, int NR, bool MASK>
__global__ __launch_bounds__(CONV_THREADS) void conv2d_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    const float* __restrict__ mask_src, float* __restrict__ y,
    int T, int F, int TT, int ntiles_t, int ntiles, int dil_t, int pad_t, int in_mode, float alpha) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int P = KF;                 // outputs per lane along frequency
    constexpr int PF = (KF - 1) / 2;      // 'same' padding, low side (TF: total//2)
    const int nspr = (F + P - 1) / P;     // strips per row
    const int cols = nspr * P + KF - 1;
    const int rows = TT + (KT - 1) * dil_t;
    float* smem_out = smem + (((size_t)rows * cols * CIN + 3) & ~(size_t)3);
    const int nstrips = TT * nspr;
    constexpr bool PK = (COUT % 2 == 0);
    constexpr int HC = PK ? COUT / 2 : 1;

    int tile = blockIdx.x;
    Prefetch<CIN> pf;
    if (tile < ntiles) {
        const int b = tile / ntiles_t, t0 = (tile - b * ntiles_t) * TT;
        prefetch_tile<CIN, MASK>(pf, x, mask_src, (long long)b * T * F, T, F, rows, cols, t0 - pad_t, -PF);
    }
    while (tile < ntiles) {
        const int b = tile / ntiles_t, t0 = (tile - b * ntiles_t) * TT;
        const long long img = (long long)b * T * F;
        commit_tile<CIN, MASK>(pf, smem, rows * cols, in_scale, in_shift, in_mode, alpha);
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < ntiles) {
            const int nb = next / ntiles_t, nt0 = (next - nb * ntiles_t) * TT;
            prefetch_tile<CIN, MASK>(pf, x, mask_src, (long long)nb * T * F, T, F, rows, cols, nt0 - pad_t, -PF);
        }

        // ---- compute: NR strips per lane, advanced together so that a kernel row's weights (SGPRs) are fetched once
        float acc[NR][P][COUT];
        f2 acc2[NR][P][HC];
        int sr[NR], sf[NR];
        bool act[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int s = threadIdx.x + q * CONV_THREADS;
            const int r = s / nspr;
            act[q] = s < nstrips && t0 + r < T;
            sr[q] = act[q] ? r : 0;
            sf[q] = act[q] ? (s - r * nspr) * P : 0;
#pragma unroll
            for (int p = 0; p < P; ++p) {
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[q][p][co] = bias ? bias[co] : 0.f;
                if (PK) {
#pragma unroll
                    for (int h = 0; h < HC; ++h) acc2[q][p][h] = (f2){acc[q][p][2 * h], acc[q][p][2 * h + (PK ? 1 : 0)]};
                }
            }
        }
#pragma unroll 1   // keep one kernel row (KF*CIN*COUT weights) in SGPRs at a time; full unroll spills SGPRs
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const float* row = smem + ((size_t)(sr[q] + kt * dil_t) * cols + sf[q]) * CIN;
#pragma unroll
                for (int j = 0; j < P + KF - 1; ++j) {
                    float a[CIN];
                    VecIO<CIN>::ld(row + j * CIN, a);
#pragma unroll
                    for (int kf = 0; kf < KF; ++kf) {
                        const int p = j - kf;
                        if (p < 0 || p >= P) continue;
                        const float* wk = w + ((kt * KF + kf) * CIN) * COUT;
                        if (PK) {
                            const f2* wk2 = reinterpret_cast<const f2*>(wk);
#pragma unroll
                            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                                for (int h = 0; h < HC; ++h)
                                    acc2[q][p][h] = pkfma(a[ci], wk2[ci * HC + h], acc2[q][p][h]);
                        } else {
#pragma unroll
                            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                                for (int co = 0; co < COUT; ++co)
                                    acc[q][p][co] = fmaf(a[ci], wk[ci * COUT + co], acc[q][p][co]);
                        }
                    }
                }
            }
        }
        // ---- outputs go through LDS so that the global stores are whole contiguous lines (the tile's rows are one
        //      contiguous range of y); a lane's own strip is 5 x 16 B at an 80-B stride otherwise.
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            if (!act[q]) continue;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (sf[q] + p >= F) continue;
                float o[COUT];
                if (PK) {
#pragma unroll
                    for (int h = 0; h < HC; ++h) { o[2 * h] = acc2[q][p][h].x; o[2 * h + (PK ? 1 : 0)] = acc2[q][p][h].y; }
                } else {
#pragma unroll
                    for (int co = 0; co < COUT; ++co) o[co] = acc[q][p][co];
                }
                VecIO<COUT>::st(smem_out + ((size_t)sr[q] * F + sf[q] + p) * COUT, o);
            }
        }
        __syncthreads();   // in-tile reads done (next commit may overwrite it); out-tile complete
        const int nrows = min(TT, T - t0);
        const int nout = nrows * F * COUT;
        float* yo = y + (img + (long long)t0 * F) * COUT;
        if ((F * COUT) % 4 == 0) {
            for (int i = threadIdx.x * 4; i < nout; i += CONV_THREADS * 4)
                *reinterpret_cast<float4*>(yo + i) = *reinterpret_cast<const float4*>(smem_out + i);
        } else {
            for (int i = threadIdx.x; i < nout; i += CONV_THREADS) yo[i] = smem_out[i];
        }
        tile = next;
        // the next iteration's barrier (after commit) orders these out-tile reads before the next out-tile writes
    }
}

// Same persistent pipeline, FMAs on the matrix pipe: v_mfma_f32_4x4x1_16B_f32 computes, for 16 blocks at once,
// D[4 pixels x 4 out-channels] += A[4 pixels x 1] . B[1 x 4 out-channels] -- one (tap, in-channel) pair per
// instruction, 64 pixels per wave.  The weights sit in VGPRs for the whole launch (lane j%4 holds w[tap][ci][j]),
// the activation vector of a pixel's tap comes from LDS with one ds_read_b128 per 4 MFMAs, the VALU only computes
// addresses.  Full fp32 rate (64 flop/clk/SIMD) without the SGPR-operand issue limits of the v_pk_fma_f32 form.
// Layout checked on hardware (tools/mfma4_test.hip): lane 4b+j, register r  <-  sum_k A(lane 4b+r) * B(lane 4b+j).
typedef float f32x4m __attribute__((ext_vector_type(4)));

template <int CIN, int COUT, int KT, int KF, bool MASK, int NGI>
__global__ __launch_bounds__(CONV_THREADS) void conv2d_fwd_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    const float* __restrict__ mask_src, float* __restrict__ y,
    int T, int F, int TT, int ntiles_t, int ntiles, int dil_t, int pad_t, int in_mode, float alpha) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int P = KF;
    constexpr int PF = (KF - 1) / 2;
    const int nspr = (F + P - 1) / P;
    const int cols = nspr * P + KF - 1;
    const int rows = TT + (KT - 1) * dil_t;
    float* smem_out = smem + (((size_t)rows * cols * CIN + 3) & ~(size_t)3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j4 = lane & 3;

    // weights: one VGPR per (tap, ci), lane j holds output channel j (0 beyond COUT)
    float wreg[KT * KF][CIN];
#pragma unroll
    for (int tap = 0; tap < KT * KF; ++tap)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) wreg[tap][ci] = j4 < COUT ? w[(tap * CIN + ci) * COUT + j4] : 0.f;
    const float bj = (bias && j4 < COUT) ? bias[j4] : 0.f;

    int tile = blockIdx.x;
    Prefetch<CIN> pf;
    if (tile < ntiles) {
        const int b = tile / ntiles_t, t0 = (tile - b * ntiles_t) * TT;
        prefetch_tile<CIN, MASK>(pf, x, mask_src, (long long)b * T * F, T, F, rows, cols, t0 - pad_t, -PF);
    }
    while (tile < ntiles) {
        const int b = tile / ntiles_t, t0 = (tile - b * ntiles_t) * TT;
        const long long img = (long long)b * T * F;
        commit_tile<CIN, MASK>(pf, smem, rows * cols, in_scale, in_shift, in_mode, alpha);
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < ntiles) {
            const int nb = next / ntiles_t, nt0 = (next - nb * ntiles_t) * TT;
            prefetch_tile<CIN, MASK>(pf, x, mask_src, (long long)nb * T * F, T, F, rows, cols, nt0 - pad_t, -PF);
        }
        const int nrows = min(TT, T - t0);
        const int npix = nrows * F;
        const int ngroups = (npix + 63) >> 6;
        // NGI pixel groups of 64 advance together: independent accumulators between dependent MFMAs
        for (int g0 = wave * NGI; g0 < ngroups; g0 += 4 * NGI) {
            f32x4m acc[NGI][CIN];    // one chain per (group, in-channel): consecutive MFMAs never share an accumulator
            const float* base[NGI];
            int pixn[NGI];
#pragma unroll
            for (int u = 0; u < NGI; ++u) {
                const int n = (g0 + u) * 64 + lane;
                pixn[u] = n;
                const int nn = n < npix ? n : 0;
                const int r = nn / F, f = nn - r * F;
                base[u] = smem + ((size_t)r * cols + f) * CIN;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) acc[u][ci] = ci == 0 ? (f32x4m){bj, bj, bj, bj} : (f32x4m){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                for (int kf = 0; kf < KF; ++kf) {
                    float a[NGI][CIN];
#pragma unroll
                    for (int u = 0; u < NGI; ++u) VecIO<CIN>::ld(base[u] + ((size_t)kt * dil_t * cols + kf) * CIN, a[u]);
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                        for (int u = 0; u < NGI; ++u)
                            acc[u][ci] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u][ci], wreg[kt * KF + kf][ci], acc[u][ci], 0, 0, 0);
                }
            }
            // lane 4b+j, register r = pixel 4b+r of the group, channel j
#pragma unroll
            for (int u = 0; u < NGI; ++u) {
                if ((g0 + u) >= ngroups || j4 >= COUT) continue;
                const int pb = (g0 + u) * 64 + (lane & ~3);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[u][0][r];
#pragma unroll
                    for (int ci = 1; ci < CIN; ++ci) v += acc[u][ci][r];
                    if (pb + r < npix) smem_out[(size_t)(pb + r) * COUT + j4] = v;
                }
            }
        }
        __syncthreads();
        const int nout = npix * COUT;
        float* yo = y + (img + (long long)t0 * F) * COUT;
        if ((F * COUT) % 4 == 0) {
            for (int i = threadIdx.x * 4; i < nout; i += CONV_THREADS * 4)
                *reinterpret_cast<float4*>(yo + i) = *reinterpret_cast<const float4*>(smem_out + i);
        } else {
            for (int i = threadIdx.x; i < nout; i += CONV_THREADS) yo[i] = smem_out[i];
        }
        tile = next;
    }
}

// ------------------------------------------------------------------------------------------
// fused backward
// LDS: dyt [TT + (KT-1)*dil_t][Fp + KF-1][COUT] | at [TT][Fp][CIN] | xt [TT][Fp][CIN] (affine only)
// partials[block][NW + COUT + 2*CIN]
// ------------------------------------------------------------------------------------------
template <int CIN, int COUT, int KT, int KF>
__global__ __launch_bounds__(CONV_THREADS) void conv2d_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ w,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    const float* __restrict__ mask_src, float* __restrict__ dx, float* __restrict__ partials,
    const float* __restrict__ wt /*unused: head of the workspace, kept for the layout of the partials behind it*/,
    int want_dx, int want_dw, int want_affine,
    int T, int F, int TT, int dil_t, int pad_t, int in_mode, float alpha) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int P = KF;
    constexpr int PF = (KF - 1) / 2;
    constexpr int LO_F = (KF - 1) - PF;   // low-side halo of dy along frequency
    constexpr int NW = KT * KF * CIN * COUT;
    constexpr int NPART = NW + COUT + 2 * CIN;
    const int nspr = (F + P - 1) / P;
    const int Fp = nspr * P;
    const int cols = Fp + KF - 1;
    const int halo_t = (KT - 1) * dil_t;
    const int lo_t = halo_t - pad_t;      // dyt row 0 <-> t = t0 - lo_t
    const int rows = TT + halo_t;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * TT;
    const long long img = (long long)b * T * F;

    float* dyt = smem;
    float* at = dyt + (size_t)rows * cols * COUT;
    float* xt = at + (size_t)TT * Fp * CIN;
    const int tid = threadIdx.x;

    stage_window<COUT, 6>(dy, nullptr, dyt, nullptr, img, T, F, rows, cols, t0 - lo_t, -LO_F, nullptr, nullptr, PTTS_IN_NONE, alpha);
    stage_window<CIN, 4>(x, mask_src, at, want_affine ? xt : nullptr, img, T, F, TT, Fp, t0, 0, in_scale, in_shift, in_mode, alpha);
    __syncthreads();

    // ---- phase 1: dx = conv^T(dy, w) * d(a)/d(x), plus dscale/dshift sums --------------------
    float s_scale[CIN], s_shift[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) { s_scale[c] = 0.f; s_shift[c] = 0.f; }
    if (want_dx || want_affine) {
        const int nstrips = TT * nspr;
        for (int s = tid; s < nstrips; s += CONV_THREADS) {
            const int r = s / nspr;
            const int fs = (s - r * nspr) * P;
            const int t = t0 + r;
            if (t >= T) continue;
            // packed over PAIRS OF co: da[p][ci] keeps the partial sums of the even and of the odd output channels in the two
            // halves of one register pair; g's pairs are the natural halves of the loaded pixel and the weight pairs
            // (w[tap][ci][co], w[tap][ci][co+1]) are adjacent in the kernel as stored -- no transposed copy of w is needed
            constexpr bool PKI = (COUT % 2 == 0);
            constexpr int HO2 = PKI ? COUT / 2 : 1;
            float da[P][CIN];
            f2 da2[P][CIN];
#pragma unroll
            for (int p = 0; p < P; ++p) {
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) { da[p][ci] = 0.f; da2[p][ci] = (f2){0.f, 0.f}; }
            }
#pragma unroll 1
            for (int kt = 0; kt < KT; ++kt) {
                const float* row = dyt + ((size_t)(r + (KT - 1 - kt) * dil_t) * cols + fs) * COUT;
#pragma unroll
                for (int j = 0; j < P + KF - 1; ++j) {
                    float g[COUT];
                    VecIO<COUT>::ld(row + j * COUT, g);
#pragma unroll
                    for (int kf = 0; kf < KF; ++kf) {
                        const int p = j - (KF - 1 - kf);
                        if (p < 0 || p >= P) continue;
                        const float* wk = w + ((kt * KF + kf) * CIN) * COUT;
                        if (PKI) {
                            const f2* wk2 = reinterpret_cast<const f2*>(wk);
#pragma unroll
                            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                                for (int h = 0; h < HO2; ++h)
                                    da2[p][ci] = __builtin_elementwise_fma((f2){g[2 * h], g[2 * h + (PKI ? 1 : 0)]}, wk2[ci * HO2 + h], da2[p][ci]);
                        } else {
#pragma unroll
                            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                                for (int co = 0; co < COUT; ++co)
                                    da[p][ci] = fmaf(g[co], wk[ci * COUT + co], da[p][ci]);
                        }
                    }
                }
            }
            if (PKI) {
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) da[p][ci] = da2[p][ci].x + da2[p][ci].y;
            }
            const float* arow = at + ((size_t)r * Fp + fs) * CIN;
            const float* xrow = xt + ((size_t)r * Fp + fs) * CIN;
            float* dxo = dx + (img + (long long)t * F + fs) * CIN;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (fs + p >= F) continue;
                float o[CIN];
                if (in_mode == PTTS_IN_LRELU) {
                    float a[CIN];
                    VecIO<CIN>::ld(arow + p * CIN, a);
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) {
                        const float gd = da[p][ci] * (a[ci] > 0.f ? 1.f : alpha);
                        o[ci] = in_scale ? gd * in_scale[ci] : gd;
                        da[p][ci] = gd;
                    }
                    if (want_affine) {
                        float xr[CIN];
                        VecIO<CIN>::ld(xrow + p * CIN, xr);
#pragma unroll
                        for (int ci = 0; ci < CIN; ++ci) {
                            s_shift[ci] += da[p][ci];
                            s_scale[ci] += da[p][ci] * xr[ci];
                        }
                    }
                } else {
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) o[ci] = da[p][ci];
                }
                if (want_dx) VecIO<CIN>::st(dxo + p * CIN, o);
            }
        }
    }

    // ---- phase 2: dw partials.  lane role = (kt, ci); it owns acc[kf][co] and marches along f ----
    constexpr int NROLE = KT * CIN;
    constexpr int NGRP = CONV_THREADS / NROLE;
    constexpr bool PKO = (COUT % 2 == 0);
    constexpr int HO = PKO ? COUT / 2 : 1;
    float accw[KF][COUT];
    f2 accw2[KF][HO];
#pragma unroll
    for (int kf = 0; kf < KF; ++kf) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) accw[kf][co] = 0.f;
#pragma unroll
        for (int h = 0; h < HO; ++h) accw2[kf][h] = (f2){0.f, 0.f};
    }
    const int grp = tid / NROLE;
    const int role = tid - grp * NROLE;
    const int rkt = role / CIN, rci = role - rkt * CIN;
    if (want_dw && grp < NGRP) {
        for (int r = grp; r < TT; r += NGRP) {
            if (t0 + r >= T) break;
            const float* drow = dyt + (size_t)(r + (KT - 1 - rkt) * dil_t) * cols * COUT;
            const float* arow = at + (size_t)r * Fp * CIN + rci;
            float win[KF][COUT];
#pragma unroll
            for (int k = 0; k < KF - 1; ++k) VecIO<COUT>::ld(drow + k * COUT, win[k]);
            for (int f0 = 0; f0 < Fp; f0 += KF) {
#pragma unroll
                for (int u = 0; u < KF; ++u) {
                    const int fq = f0 + u;
                    // newest column fq+KF-1 goes to slot (u+KF-1)%KF
                    VecIO<COUT>::ld(drow + (size_t)(fq + KF - 1) * COUT, win[(u + KF - 1) % KF]);
                    const float av = arow[(size_t)fq * CIN];
#pragma unroll
                    for (int kf = 0; kf < KF; ++kf) {
                        // dy column fq + (KF-1-kf) lives in slot (u + KF-1-kf) % KF
                        const int slot = (u + KF - 1 - kf) % KF;
                        if (PKO) {
#pragma unroll
                            for (int h = 0; h < HO; ++h)
                                accw2[kf][h] = pkfma(av, (f2){win[slot][2 * h], win[slot][2 * h + (PKO ? 1 : 0)]}, accw2[kf][h]);
                        } else {
#pragma unroll
                            for (int co = 0; co < COUT; ++co)
                                accw[kf][co] = fmaf(av, win[slot][co], accw[kf][co]);
                        }
                    }
                }
            }
        }
    }
    if (PKO) {
#pragma unroll
        for (int kf = 0; kf < KF; ++kf)
#pragma unroll
            for (int h = 0; h < HO; ++h) {
                accw[kf][2 * h] = accw2[kf][h].x;
                accw[kf][2 * h + (PKO ? 1 : 0)] = accw2[kf][h].y;
            }
    }
    // ---- phase 3: dbias partial (sum of dy over the pixels this block owns) -------------------
    float s_b[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) s_b[co] = 0.f;
    if (want_dw) {
        for (int idx = tid; idx < TT * F; idx += CONV_THREADS) {
            const int r = idx / F, f = idx - r * F;
            if (t0 + r >= T) break;
            float g[COUT];
            VecIO<COUT>::ld(dyt + ((size_t)(r + lo_t) * cols + f + LO_F) * COUT, g);
#pragma unroll
            for (int co = 0; co < COUT; ++co) s_b[co] += g[co];
        }
    }
    __syncthreads();   // everyone is done with dyt/at: reuse LDS as reduction scratch

    float* red = smem;                        // [NGRP][NW]
    float* red2 = smem + (size_t)NGRP * NW;   // [4 waves][COUT + 2*CIN]
    if (want_dw && grp < NGRP) {
#pragma unroll
        for (int kf = 0; kf < KF; ++kf)
#pragma unroll
            for (int co = 0; co < COUT; ++co)
                red[(size_t)grp * NW + ((rkt * KF + kf) * CIN + rci) * COUT + co] = accw[kf][co];
    }
    {
        const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float v = wave_sum(s_b[co]);
            if (lane == 0) red2[wave * (COUT + 2 * CIN) + co] = v;
        }
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            const float v1 = wave_sum(s_scale[ci]);
            const float v2 = wave_sum(s_shift[ci]);
            if (lane == 0) {
                red2[wave * (COUT + 2 * CIN) + COUT + ci] = v1;
                red2[wave * (COUT + 2 * CIN) + COUT + CIN + ci] = v2;
            }
        }
    }
    __syncthreads();
    float* out = partials + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NPART;
    if (want_dw) {
        for (int j = tid; j < NW; j += CONV_THREADS) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < NGRP; ++g) s += red[(size_t)g * NW + j];
            out[j] = s;
        }
    }
    if (tid < COUT + 2 * CIN) {
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < CONV_THREADS / 64; ++wv) s += red2[wv * (COUT + 2 * CIN) + tid];
        out[NW + tid] = s;
    }
}

// out[j] = sum over blocks of partials[blk][j], fixed order -> deterministic. grid = npart, 256 threads.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partials,
                                                              int nblocks, int npart,
                                                              float* __restrict__ dw, int nw,
                                                              float* __restrict__ dbias, int cout,
                                                              float* __restrict__ dscale,
                                                              float* __restrict__ dshift, int cin) {
    __shared__ double sh[4];
    const int j = blockIdx.x;
    double s = 0.0;
    for (int blk = threadIdx.x; blk < nblocks; blk += 256) s += (double)partials[(size_t)blk * npart + j];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
        if (j < nw) { if (dw) dw[j] = v; }
        else if (j < nw + cout) { if (dbias) dbias[j - nw] = v; }
        else if (j < nw + cout + cin) { if (dscale) dscale[j - nw - cout] = v; }
        else { if (dshift) dshift[j - nw - cout - cin] = v; }
    }
}

// ------------------------------------------------------------------------------------------
// generic fallbacks (any Cin/Cout/KT/KF): correct, unoptimised, global-memory only.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float gen_in(const float* __restrict__ x, const float* __restrict__ in_scale,
                                        const float* __restrict__ in_shift,
                                        const float* __restrict__ mask_src, long long off, int c,
                                        int in_mode, float alpha) {
    float v = x[off];
    if (in_mode == PTTS_IN_LRELU) {
        if (in_scale) v = v * in_scale[c] + in_shift[c];
        v = lrelu(v, alpha);
    } else if (in_mode == PTTS_IN_MASKMUL) {
        v *= lrelu_d(mask_src[off], alpha);
    }
    return v;
}

__global__ void conv2d_fwd_generic(const float* __restrict__ x, const float* __restrict__ w,
                                   const float* __restrict__ bias, const float* __restrict__ in_scale,
                                   const float* __restrict__ in_shift, const float* __restrict__ mask_src,
                                   float* __restrict__ y, long long total, int T, int F, int Cin, int Cout,
                                   int KT, int KF, int dil_t, int pad_t, int in_mode, float alpha) {
    const int PF = (KF - 1) / 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        long long pix = i / Cout;
        const int f = (int)(pix % F);
        pix /= F;
        const int t = (int)(pix % T);
        const long long b = pix / T;
        float acc = bias ? bias[co] : 0.f;
        for (int kt = 0; kt < KT; ++kt) {
            const int tt = t + kt * dil_t - pad_t;
            if (tt < 0 || tt >= T) continue;
            for (int kf = 0; kf < KF; ++kf) {
                const int ff = f + kf - PF;
                if (ff < 0 || ff >= F) continue;
                const long long base = ((b * T + tt) * F + ff) * Cin;
                for (int ci = 0; ci < Cin; ++ci)
                    acc = fmaf(gen_in(x, in_scale, in_shift, mask_src, base + ci, ci, in_mode, alpha),
                               w[((kt * KF + kf) * Cin + ci) * Cout + co], acc);
            }
        }
        y[i] = acc;
    }
}

// dx and (per-element, non-reduced) affine terms; one thread per (pixel, ci)
__global__ void conv2d_bwd_dx_generic(const float* __restrict__ dy, const float* __restrict__ x,
                                      const float* __restrict__ w, const float* __restrict__ in_scale,
                                      const float* __restrict__ in_shift, float* __restrict__ dx,
                                      long long total, int T, int F, int Cin, int Cout, int KT, int KF,
                                      int dil_t, int pad_t, int in_mode, float alpha) {
    const int PF = (KF - 1) / 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        long long pix = i / Cin;
        const int f = (int)(pix % F);
        pix /= F;
        const int t = (int)(pix % T);
        const long long b = pix / T;
        float da = 0.f;
        for (int kt = 0; kt < KT; ++kt) {
            const int tt = t - kt * dil_t + pad_t;
            if (tt < 0 || tt >= T) continue;
            for (int kf = 0; kf < KF; ++kf) {
                const int ff = f - kf + PF;
                if (ff < 0 || ff >= F) continue;
                const long long base = ((b * T + tt) * F + ff) * Cout;
                for (int co = 0; co < Cout; ++co)
                    da = fmaf(dy[base + co], w[((kt * KF + kf) * Cin + ci) * Cout + co], da);
            }
        }
        if (in_mode == PTTS_IN_LRELU) {
            float p = x[i];
            if (in_scale) p = p * in_scale[ci] + in_shift[ci];
            da *= lrelu_d(p, alpha);
            if (in_scale) da *= in_scale[ci];
        }
        dx[i] = da;
    }
}

// one block per weight element (or per bias / dscale / dshift channel); fixed-order reduction.
__global__ __launch_bounds__(256) void conv2d_bwd_w_generic(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ w,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    const float* __restrict__ mask_src, float* __restrict__ dw, float* __restrict__ dbias,
    float* __restrict__ dscale, float* __restrict__ dshift, long long npix, int T, int F, int Cin,
    int Cout, int KT, int KF, int dil_t, int pad_t, int in_mode, float alpha) {
    __shared__ double sh[4];
    const int PF = (KF - 1) / 2;
    const int NW = KT * KF * Cin * Cout;
    const int j = blockIdx.x;
    double s = 0.0;
    if (j < NW) {
        const int co = j % Cout, ci = (j / Cout) % Cin, kf = (j / (Cout * Cin)) % KF, kt = j / (Cout * Cin * KF);
        for (long long pix = threadIdx.x; pix < npix; pix += 256) {
            const int f = (int)(pix % F);
            const int t = (int)((pix / F) % T);
            const long long b = pix / ((long long)F * T);
            const int tt = t + kt * dil_t - pad_t, ff = f + kf - PF;
            if (tt < 0 || tt >= T || ff < 0 || ff >= F) continue;
            const long long off = ((b * T + tt) * F + ff) * Cin + ci;
            s += (double)(gen_in(x, in_scale, in_shift, mask_src, off, ci, in_mode, alpha) * dy[pix * Cout + co]);
        }
    } else if (j < NW + Cout) {
        const int co = j - NW;
        for (long long pix = threadIdx.x; pix < npix; pix += 256) s += (double)dy[pix * Cout + co];
    } else {
        // dscale (first Cin) / dshift (next Cin): recompute da at each pixel
        const int which = (j - NW - Cout) / Cin, ci = (j - NW - Cout) % Cin;
        for (long long pix = threadIdx.x; pix < npix; pix += 256) {
            const int f = (int)(pix % F);
            const int t = (int)((pix / F) % T);
            const long long b = pix / ((long long)F * T);
            float da = 0.f;
            for (int kt = 0; kt < KT; ++kt) {
                const int tt = t - kt * dil_t + pad_t;
                if (tt < 0 || tt >= T) continue;
                for (int kf = 0; kf < KF; ++kf) {
                    const int ff = f - kf + PF;
                    if (ff < 0 || ff >= F) continue;
                    const long long base = ((b * T + tt) * F + ff) * Cout;
                    for (int co = 0; co < Cout; ++co)
                        da = fmaf(dy[base + co], w[((kt * KF + kf) * Cin + ci) * Cout + co], da);
                }
            }
            const float xr = x[pix * Cin + ci];
            const float p = xr * in_scale[ci] + in_shift[ci];
            da *= lrelu_d(p, alpha);
            s += (double)(which == 0 ? da * xr : da);
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
        if (j < NW) { if (dw) dw[j] = v; }
        else if (j < NW + Cout) { if (dbias) dbias[j - NW] = v; }
        else if (j < NW + Cout + Cin) { if (dscale) dscale[j - NW - Cout] = v; }
        else { if (dshift) dshift[j - NW - Cout - Cin] = v; }
    }
}

// ------------------------------------------------------------------------------------------
// host side: tile choice + dispatch
// ------------------------------------------------------------------------------------------
static bool conv_use_mfma() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PTTS_CONV_MFMA"); v = e ? atoi(e) : 0; }
    return v != 0;
}

// LDS per workgroup bounds the tile height; PTTS_CONV_LDS_KB overrides it (tuning knob).
static size_t lds_budget() {
    static size_t v = 0;
    if (!v) {
        const char* e = getenv("PTTS_CONV_LDS_KB");
        v = (size_t)(e ? atoi(e) : 48) * 1024;
        if (v < 8 * 1024) v = 8 * 1024;
        if (v > 150 * 1024) v = 150 * 1024;
    }
    return v;
}

struct Tile { int TT; int ntiles; };

// Pick the time-tile height: maximise (useful rows / launched rows) x (busy lanes / launched lanes)
// under the LDS budget.  Strips per tile = TT * ceil(F/KF); 256 lanes take them round-robin.
static Tile pick_tile(int T, int F, int KT, int KF, int dil_t, size_t bytes_per_row_fixed,
                      size_t bytes_per_halo_row) {
    const int nspr = (F + KF - 1) / KF;
    const int halo = (KT - 1) * dil_t;
    Tile best{0, 0};
    double best_score = -1.0;
    for (int TT = 4; TT <= 128; ++TT) {
        const size_t lds = (size_t)TT * bytes_per_row_fixed + (size_t)(TT + halo) * bytes_per_halo_row;
        if (lds > lds_budget() && best.TT > 0) break;
        const int ntiles = (T + TT - 1) / TT;
        const int strips = TT * nspr;
        const int rounds = (strips + CONV_THREADS - 1) / CONV_THREADS;
        const double lane_eff = (double)strips / (rounds * CONV_THREADS);
        const double row_eff = (double)T / ((double)ntiles * TT);
        const double halo_eff = (double)TT / (TT + halo);
        const double score = lane_eff * row_eff * (0.75 + 0.25 * halo_eff);
        if (score > best_score + 1e-9) { best_score = score; best = Tile{TT, ntiles}; }
    }
    return best;
}

// Forward tile: rows*cols must fit the prefetch registers (NPF pixels per lane); maximise lane use x row use.
static Tile pick_tile_fwd(int T, int F, int KT, int KF, int dil_t, int cols) {
    const int nspr = (F + KF - 1) / KF;
    const int halo = (KT - 1) * dil_t;
    if (const char* e = getenv("PTTS_CONV_TT")) {     // tuning knob
        const int TT = atoi(e);
        if (TT > 0 && (TT + halo) * cols <= NPF * CONV_THREADS) return Tile{TT, (T + TT - 1) / TT};
    }
    Tile best{0, 0};
    double best_score = -1.0;
    for (int TT = 1; TT <= 128; ++TT) {
        if ((TT + halo) * cols > NPF * CONV_THREADS) break;
        const int strips = TT * nspr;
        const int rounds = (strips + CONV_THREADS - 1) / CONV_THREADS;
        if (rounds > 2) break;
        const int ntiles = (T + TT - 1) / TT;
        const double lane_eff = (double)strips / (rounds * CONV_THREADS);
        const double row_eff = (double)T / ((double)ntiles * TT);
        const double halo_eff = (double)TT / (TT + halo);
        const double score = lane_eff * row_eff * (0.6 + 0.4 * halo_eff);
        if (score > best_score + 1e-9) { best_score = score; best = Tile{TT, ntiles}; }
    }
    return best;
}

struct ConvGeom {
    int pad_t;
    bool ok;
};
static ConvGeom geom(int KT, int dil_t, int pad_mode) {
    ConvGeom g;
    const int span = (KT - 1) * dil_t;
    g.pad_t = pad_mode == PTTS_PAD_CAUSAL ? span : span / 2;
    g.ok = true;
    return g;
}

#define PTTS_CONV_CASES(M)                                                                    \
    M(1, 1, 3, 3) M(1, 2, 3, 3) M(1, 4, 3, 3) M(2, 1, 3, 3) M(2, 2, 3, 3) M(2, 4, 3, 3)       \
    M(4, 1, 3, 3) M(4, 2, 3, 3) M(4, 4, 3, 3)                                                 \
    M(1, 1, 5, 5) M(1, 2, 5, 5) M(1, 4, 5, 5) M(2, 1, 5, 5) M(2, 2, 5, 5) M(2, 4, 5, 5)       \
    M(4, 1, 5, 5) M(4, 2, 5, 5) M(4, 4, 5, 5)

}  // namespace ptts

using namespace ptts;

extern "C" int ptts_conv2d_fwd(const float* x, const float* w, const float* bias, const float* in_scale,
                               const float* in_shift, const float* mask_src, float* y, int B, int T, int F,
                               int Cin, int Cout, int KT, int KF, int dil_t, int pad_mode, int in_mode,
                               float alpha, void* stream) {
    PTTS_REQUIRE(x && w && y, "conv2d_fwd: null tensor");
    PTTS_REQUIRE(B > 0 && T > 0 && F > 0 && Cin > 0 && Cout > 0 && KT > 0 && KF > 0 && dil_t > 0,
                 "conv2d_fwd: bad dims B=%d T=%d F=%d Cin=%d Cout=%d KT=%d KF=%d dil=%d", B, T, F, Cin, Cout, KT, KF, dil_t);
    PTTS_REQUIRE(in_mode >= 0 && in_mode <= 2, "conv2d_fwd: bad in_mode %d", in_mode);
    PTTS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv2d_fwd: scale/shift must come together");
    PTTS_REQUIRE(in_mode != PTTS_IN_MASKMUL || mask_src, "conv2d_fwd: MASKMUL needs mask_src");
    PTTS_REQUIRE(B <= 65535, "conv2d_fwd: B too large");
    hipStream_t st = (hipStream_t)stream;
    const ConvGeom g = geom(KT, dil_t, pad_mode);
#define FWD_LAUNCH(CI, CO, KTT, KFF, NRR, MK)                                                           \
    hipLaunchKernelGGL((conv2d_fwd_kernel<CI, CO, KTT, KFF, NRR, MK>), dim3(grid), dim3(CONV_THREADS), lds, \
                       st, x, w, bias, in_scale, in_shift, mask_src, y, T, F, tl.TT, tl.ntiles,             \
                       tl.ntiles * B, dil_t, g.pad_t, in_mode, alpha)
#define FWD_CASE(CI, CO, KTT, KFF)                                                                       \
    if (Cin == CI && Cout == CO && KT == KTT && KF == KFF) {                                             \
        const int nspr = (F + KFF - 1) / KFF;                                                            \
        const int cols = nspr * KFF + KFF - 1;                                                           \
        const Tile tl = pick_tile_fwd(T, F, KTT, KFF, dil_t, cols);                                      \
        const int rows = tl.TT + (KTT - 1) * dil_t;                                                      \
        const size_t in_fl = ((size_t)rows * cols * CI + 3) & ~(size_t)3;                                \
        const size_t lds = (in_fl + (size_t)tl.TT * F * CO) * sizeof(float);                             \
        const int nr = (tl.TT * nspr + CONV_THREADS - 1) / CONV_THREADS;                                 \
        if (tl.TT > 0 && lds <= 150 * 1024 && nr <= 2 && rows * cols <= NPF * CONV_THREADS) {            \
            int per_cu = (int)((150 * 1024) / lds);                                                      \
            if (per_cu > 4) per_cu = 4;                                                                  \
            long long grid = (long long)tl.ntiles * B;                                                   \
            if (grid > 256LL * per_cu) grid = 256LL * per_cu;                                            \
            const bool mk = in_mode == PTTS_IN_MASKMUL;                                                  \
            if (conv_use_mfma() && dil_t >= 1) {                                                         \
                if (mk) hipLaunchKernelGGL((conv2d_fwd_mfma_kernel<CI, CO, KTT, KFF, true, 2>), dim3(grid), dim3(CONV_THREADS), lds, st, x, w, bias, in_scale, in_shift, mask_src, y, T, F, tl.TT, tl.ntiles, tl.ntiles * B, dil_t, g.pad_t, in_mode, alpha); \
                else hipLaunchKernelGGL((conv2d_fwd_mfma_kernel<CI, CO, KTT, KFF, false, 2>), dim3(grid), dim3(CONV_THREADS), lds, st, x, w, bias, in_scale, in_shift, mask_src, y, T, F, tl.TT, tl.ntiles, tl.ntiles * B, dil_t, g.pad_t, in_mode, alpha); \
                return check_launch("conv2d_fwd_mfma");                                                  \
            }                                                                                            \
            if (nr == 1) { if (mk) FWD_LAUNCH(CI, CO, KTT, KFF, 1, true); else FWD_LAUNCH(CI, CO, KTT, KFF, 1, false); } \
            else { if (mk) FWD_LAUNCH(CI, CO, KTT, KFF, 2, true); else FWD_LAUNCH(CI, CO, KTT, KFF, 2, false); }         \
            return check_launch("conv2d_fwd");                                                           \
        }                                                                                                \
    }
    PTTS_CONV_CASES(FWD_CASE)
#undef FWD_CASE
#undef FWD_LAUNCH
    const long long total = (long long)B * T * F * Cout;
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(conv2d_fwd_generic, dim3(blocks), dim3(256), 0, st, x, w, bias, in_scale, in_shift,
                       mask_src, y, total, T, F, Cin, Cout, KT, KF, dil_t, g.pad_t, in_mode, alpha);
    return check_launch("conv2d_fwd_generic");
}

namespace {
constexpr size_t WT_BYTES = 4096;   // head of the workspace: transposed weights for the packed dx loop
struct BwdPlan { bool fast; Tile tl; size_t lds; int npart; };
template <int CI, int CO, int KTT, int KFF>
BwdPlan plan_bwd(int T, int F, int dil_t) {
    BwdPlan p;
    const int nspr = (F + KFF - 1) / KFF;
    const int Fp = nspr * KFF;
    const size_t halo_row = (size_t)(Fp + KFF - 1) * CO * sizeof(float);
    const size_t fixed_row = (size_t)Fp * CI * sizeof(float) * 2;   // at + xt
    p.tl = pick_tile(T, F, KTT, KFF, dil_t, fixed_row, halo_row);
    p.lds = (size_t)p.tl.TT * fixed_row + (size_t)(p.tl.TT + (KTT - 1) * dil_t) * halo_row;
    constexpr int NW = KTT * KFF * CI * CO;
    constexpr int NGRP = CONV_THREADS / (KTT * CI);
    const size_t red = ((size_t)NGRP * NW + 4 * (CO + 2 * CI)) * sizeof(float);
    if (p.lds < red) p.lds = red;
    p.npart = NW + CO + 2 * CI;
    p.fast = p.lds <= 160 * 1024 - 256;
    return p;
}
}  // namespace

extern "C" size_t ptts_conv2d_bwd_workspace_bytes(int B, int T, int F, int Cin, int Cout, int KT, int KF,
                                                  int dil_t) {
#define WS_CASE(CI, CO, KTT, KFF)                                                      \
    if (Cin == CI && Cout == CO && KT == KTT && KF == KFF) {                           \
        const BwdPlan p = plan_bwd<CI, CO, KTT, KFF>(T, F, dil_t);                     \
        if (p.fast) return WT_BYTES + (size_t)B * p.tl.ntiles * p.npart * sizeof(float); \
    }
    PTTS_CONV_CASES(WS_CASE)
#undef WS_CASE
    return 16;
}

// partials_only: the dw / dbias sums stay as per-workgroup rows in the caller's workspace ([WT_BYTES head][nblocks][npart],
// nblocks returned through *nblocks_out) for ptts_conv2d_reduce_grouped; dw / dbias are not written.
static int conv2d_bwd_impl(const float* dy, const float* x, const float* w, const float* in_scale,
                           const float* in_shift, const float* mask_src, float* dx, float* dw,
                           float* dbias, float* dscale, float* dshift, void* workspace,
                           size_t workspace_bytes, int B, int T, int F, int Cin, int Cout, int KT, int KF,
                           int dil_t, int pad_mode, int in_mode, float alpha, void* stream, int partials_only,
                           int* nblocks_out) {
    PTTS_REQUIRE(dy && x && w, "conv2d_bwd: null tensor");
    PTTS_REQUIRE(B > 0 && T > 0 && F > 0 && Cin > 0 && Cout > 0 && KT > 0 && KF > 0 && dil_t > 0,
                 "conv2d_bwd: bad dims");
    PTTS_REQUIRE(in_mode >= 0 && in_mode <= 2, "conv2d_bwd: bad in_mode %d", in_mode);
    PTTS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv2d_bwd: scale/shift must come together");
    PTTS_REQUIRE(in_mode != PTTS_IN_MASKMUL || mask_src, "conv2d_bwd: MASKMUL needs mask_src");
    PTTS_REQUIRE(!(in_mode == PTTS_IN_MASKMUL && dx), "conv2d_bwd: dx is not defined for MASKMUL (weight-only sweep)");
    PTTS_REQUIRE((dscale == nullptr) == (dshift == nullptr), "conv2d_bwd: dscale/dshift must come together");
    PTTS_REQUIRE(!dscale || (in_mode == PTTS_IN_LRELU && in_scale), "conv2d_bwd: dscale needs LRELU with scale/shift");
    PTTS_REQUIRE(B <= 65535, "conv2d_bwd: B too large");
    hipStream_t st = (hipStream_t)stream;
    const ConvGeom g = geom(KT, dil_t, pad_mode);
    const int want_dx = dx != nullptr, want_dw = (dw != nullptr || dbias != nullptr || partials_only), want_aff = dscale != nullptr;
    if (!want_dx && !want_dw && !want_aff) return PTTS_OK;
#define BWD_CASE(CI, CO, KTT, KFF)                                                                         \
    if (Cin == CI && Cout == CO && KT == KTT && KF == KFF) {                                               \
        const BwdPlan p = plan_bwd<CI, CO, KTT, KFF>(T, F, dil_t);                                         \
        if (p.fast) {                                                                                      \
            const int nblocks = B * p.tl.ntiles;                                                           \
            const size_t need = WT_BYTES + (size_t)nblocks * p.npart * sizeof(float);                      \
            if (!workspace || workspace_bytes < need) {                                                    \
                set_error("conv2d_bwd: workspace %zu < %zu", workspace_bytes, need);                       \
                return PTTS_EWORKSPACE;                                                                    \
            }                                                                                              \
            float* wt = (float*)workspace;                                                                 \
            float* parts = (float*)((char*)workspace + WT_BYTES);                                          \
            hipLaunchKernelGGL((conv2d_bwd_kernel<CI, CO, KTT, KFF>), dim3(p.tl.ntiles, B),                \
                               dim3(CONV_THREADS), p.lds, st, dy, x, w, in_scale, in_shift, mask_src, dx,  \
                               parts, (const float*)wt, want_dx, want_dw, want_aff, T, F, p.tl.TT, dil_t,  \
                               g.pad_t, in_mode, alpha);                                                   \
            int rc = check_launch("conv2d_bwd");                                                           \
            if (rc) return rc;                                                                             \
            if (nblocks_out) *nblocks_out = nblocks;                                                       \
            if ((want_dw || want_aff) && !partials_only) {                                                 \
                hipLaunchKernelGGL(reduce_partials_kernel, dim3(p.npart), dim3(256), 0, st,                \
                                   (const float*)parts, nblocks, p.npart, want_dw ? dw : nullptr,          \
                                   KTT * KFF * CI * CO, want_dw ? dbias : nullptr, CO, dscale, dshift, CI);\
                rc = check_launch("conv2d_bwd_reduce");                                                    \
            }                                                                                              \
            return rc;                                                                                     \
        }                                                                                                  \
    }
    PTTS_CONV_CASES(BWD_CASE)
#undef BWD_CASE
    PTTS_REQUIRE(!partials_only, "conv2d_bwd_partials: this shape has no tiled kernel (workspace_bytes == 16): use ptts_conv2d_bwd");
    const long long npix = (long long)B * T * F;
    if (want_dx) {
        const long long total = npix * Cin;
        const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
        hipLaunchKernelGGL(conv2d_bwd_dx_generic, dim3(blocks), dim3(256), 0, st, dy, x, w, in_scale, in_shift,
                           dx, total, T, F, Cin, Cout, KT, KF, dil_t, g.pad_t, in_mode, alpha);
        int rc = check_launch("conv2d_bwd_dx_generic");
        if (rc) return rc;
    }
    if (want_dw || want_aff) {
        const int nw = KT * KF * Cin * Cout;
        const int nblk = nw + Cout + (want_aff ? 2 * Cin : 0);
        hipLaunchKernelGGL(conv2d_bwd_w_generic, dim3(nblk), dim3(256), 0, st, dy, x, w, in_scale, in_shift,
                           mask_src, want_dw ? dw : nullptr, want_dw ? dbias : nullptr, dscale, dshift, npix, T,
                           F, Cin, Cout, KT, KF, dil_t, g.pad_t, in_mode, alpha);
        return check_launch("conv2d_bwd_w_generic");
    }
    return PTTS_OK;
}

extern "C" int ptts_conv2d_bwd(const float* dy, const float* x, const float* w, const float* in_scale,
                               const float* in_shift, const float* mask_src, float* dx, float* dw,
                               float* dbias, float* dscale, float* dshift, void* workspace,
                               size_t workspace_bytes, int B, int T, int F, int Cin, int Cout, int KT, int KF,
                               int dil_t, int pad_mode, int in_mode, float alpha, void* stream) {
    return conv2d_bwd_impl(dy, x, w, in_scale, in_shift, mask_src, dx, dw, dbias, dscale, dshift, workspace, workspace_bytes,
                           B, T, F, Cin, Cout, KT, KF, dil_t, pad_mode, in_mode, alpha, stream, 0, nullptr);
}

extern "C" int ptts_conv2d_bwd_partials(const float* dy, const float* x, const float* w, const float* mask_src,
                                        float* dx, void* workspace, size_t workspace_bytes, int* nblocks_out,
                                        int B, int T, int F, int Cin, int Cout, int KT, int KF,
                                        int dil_t, int pad_mode, int in_mode, float alpha, void* stream) {
    PTTS_REQUIRE(nblocks_out, "conv2d_bwd_partials: nblocks_out is NULL");
    return conv2d_bwd_impl(dy, x, w, nullptr, nullptr, mask_src, dx, nullptr, nullptr, nullptr, nullptr, workspace,
                           workspace_bytes, B, T, F, Cin, Cout, KT, KF, dil_t, pad_mode, in_mode, alpha, stream, 1, nblocks_out);
}

// dw[j] += sum_blk partials[blk][j] (j < nw), dbias[j - nw] += ... (nw <= j < nw + cout) for up to RG_MAX queued passes in
// one launch: block -> (pass, column); fp32 atomics because several passes may add into the same gradient buffer.
constexpr int RG_MAX = 16;
struct ReduceGroupArgs {
    int n;
    int col_begin[RG_MAX + 1];
    const float* partials[RG_MAX]; int nblocks[RG_MAX], npart[RG_MAX], nw[RG_MAX], ncout[RG_MAX];
    float* dw[RG_MAX]; float* dbias[RG_MAX];
};

// block -> (pass, chunk of RG_COLS consecutive columns, slab of RG_SLAB rows of partial sums); thread (column cx, row phase ry): the rows
// are read as whole 256-byte runs (one workgroup per COLUMN, as in rounds 1-3, fetched a 64-byte sector per 4-byte value: 20.8 us for
// 6.7 MB at the critic's 16 queued passes).  The slabs bound a thread's run of dependent iterations whatever the number of rows
// (256 for the persistent matrix-core kernels, one per tile -- thousands -- for the 1 -> 4 layer's).
constexpr int RG_COLS = 64, RG_ROWS = 4, RG_SLAB = 128;
__global__ __launch_bounds__(RG_COLS * RG_ROWS) void conv2d_reduce_grouped_kernel(ReduceGroupArgs a) {
    __shared__ double sh[RG_ROWS][RG_COLS];
    int gi = 0;
    while ((int)blockIdx.x >= a.col_begin[gi + 1]) ++gi;            // (col_begin counts (chunk, slab) items here)
    const int cx = threadIdx.x & (RG_COLS - 1), ry = threadIdx.x / RG_COLS;
    const int npart = a.npart[gi], nblocks = a.nblocks[gi], ncols = a.nw[gi] + a.ncout[gi];
    const int nchunks = (ncols + RG_COLS - 1) / RG_COLS, r = blockIdx.x - a.col_begin[gi];
    const int slab = r / nchunks, j = (r - slab * nchunks) * RG_COLS + cx;
    const int row_end = min(nblocks, (slab + 1) * RG_SLAB);
    const float* p = a.partials[gi] + j;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (j < ncols) {
        int blk = slab * RG_SLAB + ry;
        for (; blk + 3 * RG_ROWS < row_end; blk += 4 * RG_ROWS) {
            const float v0 = p[(size_t)blk * npart], v1 = p[(size_t)(blk + RG_ROWS) * npart];
            const float v2 = p[(size_t)(blk + 2 * RG_ROWS) * npart], v3 = p[(size_t)(blk + 3 * RG_ROWS) * npart];
            s0 += (double)v0; s1 += (double)v1; s2 += (double)v2; s3 += (double)v3;
        }
        for (; blk < row_end; blk += RG_ROWS) s0 += (double)p[(size_t)blk * npart];
    }
    sh[ry][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ry == 0 && j < ncols) {
        double t = sh[0][cx];
#pragma unroll
        for (int q = 1; q < RG_ROWS; ++q) t += sh[q][cx];
        const float v = (float)t;
        if (j < a.nw[gi]) { if (a.dw[gi]) atomicAdd(a.dw[gi] + j, v); }
        else if (a.dbias[gi]) atomicAdd(a.dbias[gi] + (j - a.nw[gi]), v);
    }
}

extern "C" int ptts_conv2d_reduce_grouped(const ptts_conv2d_reduce_desc* descs, int n, void* stream) {
    PTTS_REQUIRE(descs && n > 0, "conv2d_reduce_grouped: nothing to reduce");
    for (int base = 0; base < n; base += RG_MAX) {
        ReduceGroupArgs a;
        a.n = n - base < RG_MAX ? n - base : RG_MAX;
        int cols = 0;
        for (int i = 0; i < a.n; ++i) {
            const ptts_conv2d_reduce_desc& d = descs[base + i];
            PTTS_REQUIRE(d.partials && d.nblocks > 0 && d.nw > 0 && d.cout > 0 && d.npart >= d.nw + d.cout,
                         "conv2d_reduce_grouped: bad pass %d", base + i);
            a.col_begin[i] = cols;
            cols += ((d.nw + d.cout + RG_COLS - 1) / RG_COLS) * ((d.nblocks + RG_SLAB - 1) / RG_SLAB);
            a.partials[i] = d.partials; a.nblocks[i] = d.nblocks; a.npart[i] = d.npart; a.nw[i] = d.nw; a.ncout[i] = d.cout;
            a.dw[i] = d.dw; a.dbias[i] = d.dbias;
        }
        a.col_begin[a.n] = cols;
        hipLaunchKernelGGL(conv2d_reduce_grouped_kernel, dim3(cols), dim3(RG_COLS * RG_ROWS), 0, (hipStream_t)stream, a);
        int rc = check_launch("conv2d_reduce_grouped");
        if (rc) return rc;
    }
    return PTTS_OK;
}

// Channel-last reductions, BatchNorm statistics, activations, WGAN-GP pieces, losses, Adam, clip.
// All of these are HBM-bound single-pass kernels over [rows, C] tensors (C innermost).
#include "common.h"

namespace ptts {

constexpr int EW_THREADS = 256;

static inline int ew_blocks(long long n, int per_thread = 4) {
    long long b = (n + (long long)EW_THREADS * per_thread - 1) / ((long long)EW_THREADS * per_thread);
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;   // 256 CUs x 8: grid-stride the rest
    return (int)b;
}

// ---------------------------------------------------------------------------------------------
// two-quantity column reduction over [rows, C]: deterministic two-stage, fp64 accumulation
// ---------------------------------------------------------------------------------------------
constexpr int MAXC_SLOTS = 8;   // C <= 2048

template <class Op>
__global__ __launch_bounds__(EW_THREADS) void colreduce2_kernel(Op op, long long rows, int C,
                                                                double* __restrict__ partials) {
    __shared__ double sh[2][EW_THREADS];
    const int tid = threadIdx.x;
    double* out = partials + (size_t)blockIdx.x * 2 * C;
    if (C <= EW_THREADS) {
        const int R = EW_THREADS / C;
        const int r = tid / C, c = tid - r * C;
        double q1 = 0.0, q2 = 0.0;
        if (r < R) {
#pragma unroll 4
            for (long long row = (long long)blockIdx.x * R + r; row < rows; row += (long long)gridDim.x * R) {
                float a, b;
                op(row * C + c, c, a, b);
                q1 += (double)a;
                q2 += (double)b;
            }
        }
        sh[0][tid] = q1;
        sh[1][tid] = q2;
        __syncthreads();
        if (tid < C) {
            double s1 = 0.0, s2 = 0.0;
            for (int rr = 0; rr < R; ++rr) { s1 += sh[0][rr * C + tid]; s2 += sh[1][rr * C + tid]; }
            out[tid] = s1;
            out[C + tid] = s2;
        }
    } else {
        double q1[MAXC_SLOTS], q2[MAXC_SLOTS];
#pragma unroll
        for (int s = 0; s < MAXC_SLOTS; ++s) { q1[s] = 0.0; q2[s] = 0.0; }
        for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
#pragma unroll
            for (int s = 0; s < MAXC_SLOTS; ++s) {
                const int c = tid + s * EW_THREADS;
                if (c < C) {
                    float a, b;
                    op(row * C + c, c, a, b);
                    q1[s] += (double)a;
                    q2[s] += (double)b;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < MAXC_SLOTS; ++s) {
            const int c = tid + s * EW_THREADS;
            if (c < C) { out[c] = q1[s]; out[C + c] = q2[s]; }
        }
    }
}

// Vectorised form of the two-quantity column reduction for C % 4 == 0, C <= 1024 and 16-byte aligned operands: every
// lane owns four adjacent channels and streams float4.  The loads of VU row sweeps are issued before any of them is
// consumed (a lone 16-byte load per lane per iteration leaves the kernel latency-bound at ~1 TB/s); fp32 partials over
// 16 rows are folded into fp64.  Op4 provides  load(off, regs)  and  apply(off, c, regs, a[4], b[4]).
constexpr int VU = 4;

template <class Op4>
__global__ __launch_bounds__(EW_THREADS) void colreduce2_vec4_kernel(Op4 op, long long rows, int C,
                                                                     double* __restrict__ partials) {
    __shared__ double sh[8][EW_THREADS];
    const int tid = threadIdx.x;
    const int C4 = C / 4;
    const int R = EW_THREADS / C4;              // rows per sweep
    const int r = tid / C4, c4 = tid - r * C4;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (r < R) {
        float fs[4] = {0, 0, 0, 0}, fq[4] = {0, 0, 0, 0};
        int n = 0;
        const long long step = (long long)gridDim.x * R * VU;
        for (long long base = (long long)blockIdx.x * R * VU + r; base < rows; base += step) {
            typename Op4::Regs regs[VU];
#pragma unroll
            for (int u = 0; u < VU; ++u) {
                const long long row = base + (long long)u * R;
                if (row < rows) op.load(row * C + c4 * 4, regs[u]);
            }
#pragma unroll
            for (int u = 0; u < VU; ++u) {
                const long long row = base + (long long)u * R;
                if (row < rows) {
                    float a[4], b[4];
                    op.apply(row * C + c4 * 4, c4 * 4, regs[u], a, b);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { fs[e] += a[e]; fq[e] += b[e]; }
                }
            }
            if (++n == 16 / VU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { s[e] += fs[e]; q[e] += fq[e]; fs[e] = 0.f; fq[e] = 0.f; }
                n = 0;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += fs[e]; q[e] += fq[e]; }
    }
    double* out = partials + (size_t)blockIdx.x * 2 * C;
    if (C4 <= 32 && (C4 & (C4 - 1)) == 0 && R * C4 == EW_THREADS) {
        // few channels (the C = 4 conv stacks: C4 = 1): the lanes of a wave that own the same channels differ in the lane
        // bits >= log2(C4); butterfly over those, then one LDS slot per wave instead of a serial sum over R rows
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            for (int m = 32; m >= C4; m >>= 1) {
                s[e] += __shfl_xor(s[e], m, 64);
                q[e] += __shfl_xor(q[e], m, 64);
            }
        }
        const int lane = tid & 63, wave = tid >> 6;
        if (lane < C4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { sh[e][wave * C4 + lane] = s[e]; sh[4 + e][wave * C4 + lane] = q[e]; }
        }
        __syncthreads();
        if (tid < C4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double a = 0.0, b = 0.0;
                for (int wv = 0; wv < EW_THREADS / 64; ++wv) { a += sh[e][wv * C4 + tid]; b += sh[4 + e][wv * C4 + tid]; }
                out[tid * 4 + e] = a;
                out[C + tid * 4 + e] = b;
            }
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { sh[e][tid] = s[e]; sh[4 + e][tid] = q[e]; }
    __syncthreads();
    if (tid < C4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double a = 0.0, b = 0.0;
            for (int rr = 0; rr < R; ++rr) { a += sh[e][rr * C4 + tid]; b += sh[4 + e][rr * C4 + tid]; }
            out[tid * 4 + e] = a;
            out[C + tid * 4 + e] = b;
        }
    }
}

// out[j] = sum_b partials[b][j] in a fixed order.  grid = ceil(n2c/8), 256 lanes = 8 columns x 32 row groups, so
// the serial chain per lane is nblocks/32 L2-resident loads.
__global__ __launch_bounds__(256) void colreduce_final_kernel(const double* __restrict__ partials, int nblocks,
                                                              int n2c, double* __restrict__ out) {
    __shared__ double sh[32][8];
    const int jj = threadIdx.x & 7, g = threadIdx.x >> 3;
    const int j = blockIdx.x * 8 + jj;
    double s = 0.0;
    if (j < n2c) {
#pragma unroll 4
        for (int b = g; b < nblocks; b += 32) s += partials[(size_t)b * n2c + j];
    }
    sh[g][jj] = s;
    __syncthreads();
    if (g == 0 && j < n2c) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += sh[k][jj];
        out[j] = t;
    }
}

static int colreduce_blocks(long long rows, int C) {
    const int R = C <= EW_THREADS ? EW_THREADS / C : 1;
    long long b = (rows + (long long)R * 4 - 1) / ((long long)R * 4);
    if (b < 1) b = 1;
    if (b > 512) b = 512;
    return (int)b;
}

struct StatsOp {
    const float* x; int in_mode; const float* in_scale; const float* in_shift; const float* mask_src; float alpha;
    __device__ __forceinline__ void operator()(long long i, int c, float& a, float& b) const {
        float v = x[i];
        if (in_mode == PTTS_IN_LRELU) {
            if (in_scale) v = v * in_scale[c] + in_shift[c];
            v = lrelu(v, alpha);
        } else if (in_mode == PTTS_IN_MASKMUL) {
            v *= lrelu_d(mask_src[i], alpha);
        }
        a = v;
        b = v * v;
    }
};

__device__ __forceinline__ float act_fwd(float p, int act, float alpha) {
    switch (act) {
        case PTTS_ACT_LRELU: return lrelu(p, alpha);
        case PTTS_ACT_SIGMOID: return 1.f / (1.f + __expf(-p));
        case PTTS_ACT_TANH: return tanhf(p);
        default: return p;
    }
}

struct ActBwdOp {
    const float* dy; const float* x; const float* y; const float* scale; const float* shift; float* dx;
    int act; float alpha;
    __device__ __forceinline__ void operator()(long long i, int c, float& a, float& b) const {
        const float xv = x[i];
        const float sc = scale ? scale[c] : 1.f;
        float d;
        if (act == PTTS_ACT_LRELU) {
            const float p = scale ? xv * sc + shift[c] : xv;
            d = lrelu_d(p, alpha);
        } else if (act == PTTS_ACT_SIGMOID) {
            const float yv = y[i];
            d = yv * (1.f - yv);
        } else if (act == PTTS_ACT_TANH) {
            const float yv = y[i];
            d = 1.f - yv * yv;
        } else {
            d = 1.f;
        }
        const float gd = dy[i] * d;
        if (dx) dx[i] = gd * sc;
        a = gd * xv;   // dscale
        b = gd;        // dshift
    }
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

struct StatsOp4 {
    const float* x; int in_mode; const float* in_scale; const float* in_shift; const float* mask_src; float alpha;
    struct Regs { float4 v, m; };
    __device__ __forceinline__ void load(long long off, Regs& r) const {
        r.v = ld4(x + off);
        if (in_mode == PTTS_IN_MASKMUL) r.m = ld4(mask_src + off);
    }
    __device__ __forceinline__ void apply(long long off, int c, const Regs& r, float* a, float* b) const {
        float v[4] = {r.v.x, r.v.y, r.v.z, r.v.w};
        const float m[4] = {r.m.x, r.m.y, r.m.z, r.m.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (in_mode == PTTS_IN_LRELU) {
                if (in_scale) v[e] = v[e] * in_scale[c + e] + in_shift[c + e];
                v[e] = lrelu(v[e], alpha);
            } else if (in_mode == PTTS_IN_MASKMUL) {
                v[e] *= lrelu_d(m[e], alpha);
            }
            a[e] = v[e];
            b[e] = v[e] * v[e];
        }
    }
};

struct ActBwdOp4 {
    const float* dy; const float* x; const float* y; const float* scale; const float* shift; float* dx;
    int act; float alpha;
    struct Regs { float4 d, x, y; };
    __device__ __forceinline__ void load(long long off, Regs& r) const {
        r.d = ld4(dy + off);
        r.x = ld4(x + off);
        if (act == PTTS_ACT_SIGMOID || act == PTTS_ACT_TANH) r.y = ld4(y + off);
    }
    __device__ __forceinline__ void apply(long long off, int c, const Regs& r, float* a, float* b) const {
        const float dv[4] = {r.d.x, r.d.y, r.d.z, r.d.w}, xv[4] = {r.x.x, r.x.y, r.x.z, r.x.w};
        const float yv[4] = {r.y.x, r.y.y, r.y.z, r.y.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sc = scale ? scale[c + e] : 1.f;
            float d;
            if (act == PTTS_ACT_LRELU) d = lrelu_d(scale ? xv[e] * sc + shift[c + e] : xv[e], alpha);
            else if (act == PTTS_ACT_SIGMOID) d = yv[e] * (1.f - yv[e]);
            else if (act == PTTS_ACT_TANH) d = 1.f - yv[e] * yv[e];
            else d = 1.f;
            const float gd = dv[e] * d;
            o[e] = gd * sc;
            a[e] = gd * xv[e];   // dscale
            b[e] = gd;           // dshift
        }
        if (dx) *reinterpret_cast<float4*>(dx + off) = make_float4(o[0], o[1], o[2], o[3]);
    }
};

static inline bool al16(const void* p) { return p == nullptr || ((uintptr_t)p % 16) == 0; }

// workgroups of the vectorised reduction: whole passes of R*VU rows each, at most 1024
static int colreduce_vec4_blocks(long long rows, int C) {
    const int R = EW_THREADS / (C / 4);
    const long long chunks = (rows + (long long)R * VU - 1) / ((long long)R * VU);
    const long long passes = (chunks + 1023) / 1024;
    long long b = (chunks + passes - 1) / passes;
    if (b < 1) b = 1;
    return (int)b;
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, long long count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ moving_mean, float* __restrict__ moving_var, float eps,
                                   float momentum, int training, int update_moving, int unbiased_moving, int C,
                                   float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        double mean, var;
        if (training) {
            mean = sums[c] / (double)count;
            var = sums[C + c] / (double)count - mean * mean;
            if (var < 0.0) var = 0.0;
            if (update_moving) {
                const double vm = (unbiased_moving && count > 1) ? var * (double)count / (double)(count - 1) : var;
                moving_mean[c] = (float)(moving_mean[c] * (double)momentum + mean * (1.0 - (double)momentum));
                moving_var[c] = (float)(moving_var[c] * (double)momentum + vm * (1.0 - (double)momentum));
            }
        } else {
            mean = moving_mean[c];
            var = moving_var[c];
        }
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const double g = gamma ? (double)gamma[c] : 1.0;
        const double bt = beta ? (double)beta[c] : 0.0;
        scale[c] = (float)(g * rstd);
        shift[c] = (float)(bt - mean * g * rstd);
        if (mean_out) mean_out[c] = (float)mean;
        if (rstd_out) rstd_out[c] = (float)rstd;
    }
}

// Batch statistics AND the BatchNorm affine of a few-channel map (C = 4: the conv stacks) in ONE launch: the vectorised partial-sum
// stage of colreduce2_vec4_kernel, then the workgroup that finishes last (agent-scope counter) adds the partial rows in a fixed
// order -- the result does not depend on which workgroup that is -- and does bn_finalize's arithmetic.  Three launches of
// 10 + 5 + 5 us per BatchNorm layer otherwise, 8 such layers in the generator's stack.  `counter` is a zeroed int the caller keeps per
// stream; the finishing workgroup puts it back to zero.
__global__ __launch_bounds__(EW_THREADS) void bn_stats_fused_kernel(const float* __restrict__ x, long long rows, int C,
                                                                   double* __restrict__ partials, int* __restrict__ counter,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                                                   float eps, float momentum, int update_moving, int unbiased_moving,
                                                                   float* __restrict__ scale, float* __restrict__ shift,
                                                                   float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    __shared__ double sh[8][EW_THREADS];
    __shared__ int is_last;
    const int tid = threadIdx.x;
    const int C4 = C / 4;
    const int R = EW_THREADS / C4;              // rows per sweep (C4 is a power of two <= 4 here: R * C4 == EW_THREADS)
    const int r = tid / C4, c4 = tid - r * C4;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    {
        float fs[4] = {0, 0, 0, 0}, fq[4] = {0, 0, 0, 0};
        int n = 0;
        const long long step = (long long)gridDim.x * R * VU;
        for (long long base = (long long)blockIdx.x * R * VU + r; base < rows; base += step) {
            float4 v[VU];
#pragma unroll
            for (int u = 0; u < VU; ++u) {
                const long long row = base + (long long)u * R;
                v[u] = row < rows ? ld4(x + row * C + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < VU; ++u) {
                fs[0] += v[u].x; fs[1] += v[u].y; fs[2] += v[u].z; fs[3] += v[u].w;
                fq[0] += v[u].x * v[u].x; fq[1] += v[u].y * v[u].y; fq[2] += v[u].z * v[u].z; fq[3] += v[u].w * v[u].w;
            }
            if (++n == 16 / VU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { s[e] += fs[e]; q[e] += fq[e]; fs[e] = 0.f; fq[e] = 0.f; }
                n = 0;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += fs[e]; q[e] += fq[e]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        for (int m = 32; m >= C4; m >>= 1) {
            s[e] += __shfl_xor(s[e], m, 64);
            q[e] += __shfl_xor(q[e], m, 64);
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
    if (lane < C4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { sh[e][wave * C4 + lane] = s[e]; sh[4 + e][wave * C4 + lane] = q[e]; }
    }
    __syncthreads();
    double* out = partials + (size_t)blockIdx.x * 2 * C;
    if (tid < C4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double a = 0.0, b = 0.0;
            for (int wv = 0; wv < EW_THREADS / 64; ++wv) { a += sh[e][wv * C4 + tid]; b += sh[4 + e][wv * C4 + tid]; }
            out[tid * 4 + e] = a;
            out[C + tid * 4 + e] = b;
        }
        __threadfence();                                       // this workgroup's row is visible before its tick
    }
    __syncthreads();
    if (tid == 0) is_last = atomicAdd(counter, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    // the finishing workgroup: column j = tid % n2c, row group g = tid / n2c; fixed order whatever the arrival order was
    const int n2c = 2 * C, G = EW_THREADS / n2c;
    const int j = tid % n2c, g = tid / n2c;
    double t = 0.0;
    if (g < G) {
        // eight loads in flight per lane (one at a time -- a load, an add, the next load -- made this tail 15 us of a 32-us kernel);
        // the order of the additions is fixed by the indices alone
        const int nbk = (int)gridDim.x;
        int b = g;
        for (; b + 7 * G < nbk; b += 8 * G) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partials[(size_t)(b + u * G) * n2c + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        for (; b < nbk; b += G) t += partials[(size_t)b * n2c + j];
    }
    __syncthreads();                                           // sh is free again
    if (g < G) sh[0][g * n2c + j] = t;
    __syncthreads();
    if (tid < n2c) {
        double a = 0.0;
        for (int k = 0; k < G; ++k) a += sh[0][k * n2c + tid];
        sh[1][tid] = a;
    }
    __syncthreads();
    if (tid < C) {
        const int c = tid;
        const double count = (double)rows;
        const double mean = sh[1][c] / count;
        double var = sh[1][C + c] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        if (update_moving) {
            const double vm = (unbiased_moving && rows > 1) ? var * count / (count - 1.0) : var;
            moving_mean[c] = (float)(moving_mean[c] * (double)momentum + mean * (1.0 - (double)momentum));
            moving_var[c] = (float)(moving_var[c] * (double)momentum + vm * (1.0 - (double)momentum));
        }
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const double gm = gamma ? (double)gamma[c] : 1.0;
        const double bt = beta ? (double)beta[c] : 0.0;
        scale[c] = (float)(gm * rstd);
        shift[c] = (float)(bt - mean * gm * rstd);
        if (mean_out) mean_out[c] = (float)mean;
        if (rstd_out) rstd_out[c] = (float)rstd;
    }
    if (tid == 0) *counter = 0;
}

// The same finish for per-workgroup sums that ANOTHER kernel left behind (c2m::fwd_ws_kernel<.., STATS>: the convolution in front of the
// BatchNormalization layer adds up what it stores): partials[nrows][2 C] doubles -- C sums, C sums of squares -- added in index order by
// one workgroup, then bn_finalize's arithmetic.  C <= 16.
__global__ __launch_bounds__(256) void bn_finalize_partials_kernel(const double* __restrict__ partials, int nrows, long long rows, int C,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                                                   float eps, float momentum, int update_moving, int unbiased_moving,
                                                                   float* __restrict__ scale, float* __restrict__ shift,
                                                                   float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    __shared__ double sh[256];
    __shared__ double tot[32];
    const int tid = threadIdx.x, n2c = 2 * C, G = 256 / n2c;
    const int j = tid % n2c, g = tid / n2c;
    double t = 0.0;
    if (g < G) {
        // eight rows in flight per lane (one load, one add, the next load made this 1-workgroup kernel 5 us: eight dependent round trips)
        int b = g;
        for (; b + 7 * G < nrows; b += 8 * G) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partials[(size_t)(b + u * G) * n2c + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        for (; b < nrows; b += G) t += partials[(size_t)b * n2c + j];
    }
    sh[tid] = (g < G) ? t : 0.0;
    __syncthreads();
    if (tid < n2c) {
        double a = 0.0;
        for (int k = 0; k < G; ++k) a += sh[k * n2c + tid];
        tot[tid] = a;
    }
    __syncthreads();
    if (tid < C) {
        const int c = tid;
        const double count = (double)rows;
        const double mean = tot[c] / count;
        double var = tot[C + c] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        if (update_moving) {
            const double vm = (unbiased_moving && rows > 1) ? var * count / (count - 1.0) : var;
            moving_mean[c] = (float)(moving_mean[c] * (double)momentum + mean * (1.0 - (double)momentum));
            moving_var[c] = (float)(moving_var[c] * (double)momentum + vm * (1.0 - (double)momentum));
        }
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const double gm = gamma ? (double)gamma[c] : 1.0;
        const double bt = beta ? (double)beta[c] : 0.0;
        scale[c] = (float)(gm * rstd);
        shift[c] = (float)(bt - mean * gm * rstd);
        if (mean_out) mean_out[c] = (float)mean;
        if (rstd_out) rstd_out[c] = (float)rstd;
    }
}

// The same for wide layers (the Dense layers' C = 256 ...): a workgroup per 8 channels, 256 lanes = 16 columns (8 sums, 8 sums of squares)
// x 16 row groups; partials[nrows][2 C].
__global__ __launch_bounds__(256) void bn_finalize_partials_wide_kernel(const double* __restrict__ partials, int nrows, long long rows, int C,
                                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                        float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                                                        float eps, float momentum, int update_moving, int unbiased_moving,
                                                                        float* __restrict__ scale, float* __restrict__ shift,
                                                                        float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    __shared__ double sh[16][16];
    __shared__ double tot[16];
    const int tid = threadIdx.x, jj = tid & 15, g = tid >> 4;
    const int c = blockIdx.x * 8 + (jj & 7);
    const int col = (jj < 8) ? c : C + c;
    double t = 0.0;
    if (c < C) {
        int b = g;
        for (; b + 7 * 16 < nrows; b += 8 * 16) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partials[(size_t)(b + u * 16) * 2 * C + col];
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        for (; b < nrows; b += 16) t += partials[(size_t)b * 2 * C + col];
    }
    sh[g][jj] = t;
    __syncthreads();
    if (tid < 16) {
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += sh[k][tid];
        tot[tid] = a;
    }
    __syncthreads();
    if (tid < 8 && blockIdx.x * 8 + tid < C) {
        const int ch = blockIdx.x * 8 + tid;
        const double count = (double)rows;
        const double mean = tot[tid] / count;
        double var = tot[8 + tid] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        if (update_moving) {
            const double vm = (unbiased_moving && rows > 1) ? var * count / (count - 1.0) : var;
            moving_mean[ch] = (float)(moving_mean[ch] * (double)momentum + mean * (1.0 - (double)momentum));
            moving_var[ch] = (float)(moving_var[ch] * (double)momentum + vm * (1.0 - (double)momentum));
        }
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const double gm = gamma ? (double)gamma[ch] : 1.0;
        const double bt = beta ? (double)beta[ch] : 0.0;
        scale[ch] = (float)(gm * rstd);
        shift[ch] = (float)(bt - mean * gm * rstd);
        if (mean_out) mean_out[ch] = (float)mean;
        if (rstd_out) rstd_out[ch] = (float)rstd;
    }
}

__global__ void bn_bwd_coefs_kernel(const float* __restrict__ dscale, const float* __restrict__ dshift,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, long long count, int C,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta,
                                    float* __restrict__ c0, float* __restrict__ c2, int accumulate) {
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        const double g = gamma ? (double)gamma[c] : 1.0;
        const double mu = mean[c], rs = rstd[c];
        const double dsc = (double)dscale[c] - (double)dshift[c] * mu;   // total gradient w.r.t. scale
        const double dmean = -(double)dshift[c] * g * rs;
        const double dvar = -0.5 * dsc * g * rs * rs * rs;
        const double k2 = 2.0 * dvar / (double)count;
        if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)(dsc * rs);      // (one thread per channel: no race)
        if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + dshift[c];
        c2[c] = (float)k2;
        c0[c] = (float)(dmean / (double)count - k2 * mu);
    }
}

__global__ void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                  const float* __restrict__ shift, float* __restrict__ y, long long n, int C,
                                  int act, float alpha) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float p = x[i];
        if (scale) { const int c = (int)(i % C); p = p * scale[c] + shift[c]; }
        y[i] = act_fwd(p, act, alpha);
    }
}

// C % 4 == 0: one float4 per lane
__global__ void affine_act_kernel4(const float4* __restrict__ x, const float* __restrict__ scale,
                                   const float* __restrict__ shift, float4* __restrict__ y, long long n4, int C,
                                   int act, float alpha) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        float4 v = x[i];
        float p[4] = {v.x, v.y, v.z, v.w};
        if (scale) {
            const int c = (int)((i * 4) % C);
#pragma unroll
            for (int e = 0; e < 4; ++e) p[e] = p[e] * scale[c + e] + shift[c + e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = act_fwd(p[e], act, alpha);
        y[i] = make_float4(p[0], p[1], p[2], p[3]);
    }
}

__global__ void axpby_cols_kernel(const float* __restrict__ a, const float* __restrict__ c1,
                                  const float* __restrict__ x, const float* __restrict__ c2,
                                  const float* __restrict__ c0, float* __restrict__ out, long long n, int C) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        float v = c0 ? c0[c] : 0.f;
        if (a) v += a[i] * (c1 ? c1[c] : 1.f);
        if (x) v += x[i] * (c2 ? c2[c] : 1.f);
        out[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// WGAN-GP
// ---------------------------------------------------------------------------------------------
__global__ void gp_interpolate_kernel(const float* __restrict__ real, const float* __restrict__ fake,
                                      const float* __restrict__ alpha_b, float* __restrict__ out,
                                      long long TD) {
    const int b = blockIdx.y;
    const float a = alpha_b[b];
    const long long base = (long long)b * TD;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < TD;
         i += (long long)gridDim.x * blockDim.x)
        out[base + i] = a * real[base + i] + (1.f - a) * fake[base + i];
}

// one workgroup of 1024 lanes per sample: per-lane partial -> wave64 butterfly -> 16 wave sums in LDS
__global__ __launch_bounds__(1024) void gp_sqnorm_kernel(const float* __restrict__ g, float* __restrict__ out,
                                                         long long TD) {
    __shared__ double sh[16];
    const int b = blockIdx.x;
    const float* p = g + (long long)b * TD;
    double s = 0.0;
    for (long long i = threadIdx.x; i < TD; i += 1024) { const float v = p[i]; s += (double)v * (double)v; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += sh[w];
        out[b] = (float)t;
    }
}

__global__ __launch_bounds__(256) void gp_penalty_kernel(const float* __restrict__ sq, float* __restrict__ penalty,
                                                         float* __restrict__ coef, int B) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float n = sqrtf(sq[b]);
        const float d = 1.f - n;
        s += (double)d * (double)d;
        // d/dg (1-n)^2 = 2 (n-1)/n * g ; mean over B.  n == 0 gives inf/nan exactly like K.sqrt's gradient.
        if (coef) coef[b] = 2.f * (n - 1.f) / (n * (float)B);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) penalty[0] = (float)((sh[0] + sh[1] + sh[2] + sh[3]) / (double)B);
}

__global__ void gp_scale_rows_kernel(const float* __restrict__ g, const float* __restrict__ coef,
                                     const float* __restrict__ upstream, float* __restrict__ dg, long long TD) {
    const int b = blockIdx.y;
    const float s = coef[b] * (upstream ? upstream[0] : 1.f);
    const long long base = (long long)b * TD;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < TD;
         i += (long long)gridDim.x * blockDim.x)
        dg[base + i] = s * g[base + i];
}

// ---------------------------------------------------------------------------------------------
// losses: block partials in fp64, combined with one float atomic per workgroup into a zeroed scalar
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mean_scaled_kernel(const float* __restrict__ v, long long n, float scale,
                                                          float* __restrict__ out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += (double)v[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (float)((sh[0] + sh[1] + sh[2] + sh[3]) * (double)scale));
}

__global__ __launch_bounds__(256) void wlse_fwd_kernel(const float* __restrict__ y, const float* __restrict__ yhat,
                                                       const float* __restrict__ w, long long n, int D,
                                                       float scale, float* __restrict__ out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float d = y[i] - yhat[i];
        s += (double)(d * d * (w ? w[i % D] : 1.f));
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (float)((sh[0] + sh[1] + sh[2] + sh[3]) * (double)scale));
}

__global__ void wlse_bwd_kernel(const float* __restrict__ y, const float* __restrict__ yhat,
                                const float* __restrict__ w, const float* __restrict__ upstream,
                                float* __restrict__ dyhat, long long n, int D, float scale) {
    const float up = (upstream ? upstream[0] : 1.f) * scale;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        dyhat[i] = up * 2.f * (yhat[i] - y[i]) * (w ? w[i % D] : 1.f);
}

__global__ void weight_clip_kernel(float* __restrict__ p, long long n, float lo, float hi) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        p[i] = fminf(fmaxf(p[i], lo), hi);
}

__global__ void adam_tick_kernel(int* step) { *step += 1; }

__global__ void adam_keras_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                  float gscale, const int* __restrict__ step) {
    const float t = (float)(*step);
    const float lr_t = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

}  // namespace ptts

using namespace ptts;

extern "C" size_t ptts_colstats_workspace_bytes(long long rows, int C) {
    int nb = colreduce_blocks(rows, C);
    if (C % 4 == 0 && C <= 4 * EW_THREADS) { const int v = colreduce_vec4_blocks(rows, C); if (v > nb) nb = v; }
    return (size_t)nb * 2 * (size_t)C * sizeof(double);
}

template <class Op>
static int run_colreduce(const Op& op, long long rows, int C, double* out, void* workspace,
                         size_t workspace_bytes, hipStream_t st, const char* what) {
    PTTS_REQUIRE(C > 0 && C <= MAXC_SLOTS * EW_THREADS, "%s: C=%d unsupported", what, C);
    PTTS_REQUIRE(rows > 0, "%s: rows=%lld", what, rows);
    const int nb = colreduce_blocks(rows, C);
    const size_t need = (size_t)nb * 2 * C * sizeof(double);
    if (!workspace || workspace_bytes < need) {
        set_error("%s: workspace %zu < %zu", what, workspace_bytes, need);
        return PTTS_EWORKSPACE;
    }
    hipLaunchKernelGGL((colreduce2_kernel<Op>), dim3(nb), dim3(EW_THREADS), 0, st, op, rows, C, (double*)workspace);
    int rc = check_launch(what);
    if (rc) return rc;
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((2 * C + 7) / 8), dim3(256), 0, st,
                       (const double*)workspace, nb, 2 * C, out);
    return check_launch(what);
}

template <class Op4>
static int run_colreduce_vec4(const Op4& op, long long rows, int C, double* out, void* workspace,
                              size_t workspace_bytes, hipStream_t st, const char* what) {
    PTTS_REQUIRE(rows > 0, "%s: rows=%lld", what, rows);
    const int nb = colreduce_vec4_blocks(rows, C);
    const size_t need = (size_t)nb * 2 * C * sizeof(double);
    if (!workspace || workspace_bytes < need) {
        set_error("%s: workspace %zu < %zu", what, workspace_bytes, need);
        return PTTS_EWORKSPACE;
    }
    hipLaunchKernelGGL((colreduce2_vec4_kernel<Op4>), dim3(nb), dim3(EW_THREADS), 0, st, op, rows, C, (double*)workspace);
    int rc = check_launch(what);
    if (rc) return rc;
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((2 * C + 7) / 8), dim3(256), 0, st,
                       (const double*)workspace, nb, 2 * C, out);
    return check_launch(what);
}

extern "C" int ptts_colstats(const float* x, long long rows, int C, int in_mode, const float* in_scale,
                             const float* in_shift, const float* mask_src, float alpha, double* sums,
                             void* workspace, size_t workspace_bytes, void* stream) {
    PTTS_REQUIRE(x && sums, "colstats: null tensor");
    PTTS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "colstats: scale/shift must come together");
    if (C % 4 == 0 && C <= 4 * EW_THREADS && al16(x) && al16(mask_src)) {
        StatsOp4 op4{x, in_mode, in_scale, in_shift, mask_src, alpha};
        return run_colreduce_vec4(op4, rows, C, sums, workspace, workspace_bytes, (hipStream_t)stream, "colstats");
    }
    StatsOp op{x, in_mode, in_scale, in_shift, mask_src, alpha};
    return run_colreduce(op, rows, C, sums, workspace, workspace_bytes, (hipStream_t)stream, "colstats");
}

extern "C" int ptts_bn_finalize(const double* sums, long long count, const float* gamma, const float* beta,
                                float* moving_mean, float* moving_var, float eps, float momentum, int training,
                                int update_moving, int unbiased_moving, int C, float* scale, float* shift,
                                float* mean, float* rstd, void* stream) {
    PTTS_REQUIRE(scale && shift && C > 0, "bn_finalize: null output");
    PTTS_REQUIRE(!training || sums, "bn_finalize: training needs sums");
    PTTS_REQUIRE(training || (moving_mean && moving_var), "bn_finalize: inference needs moving stats");
    PTTS_REQUIRE(!(training && update_moving) || (moving_mean && moving_var), "bn_finalize: update needs moving stats");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, count,
                       gamma, beta, moving_mean, moving_var, eps, momentum, training, update_moving,
                       unbiased_moving, C, scale, shift, mean, rstd);
    return check_launch("bn_finalize");
}

// 1 when ptts_bn_batch_stats takes the shape: few channels (C = 4, 8, 16), 16-byte aligned rows
extern "C" int ptts_bn_batch_stats_supported(long long rows, int C) {
    return (rows > 0 && (C == 4 || C == 8 || C == 16)) ? 1 : 0;
}

extern "C" int ptts_bn_batch_stats(const float* x, long long rows, int C, const float* gamma, const float* beta,
                                   float* moving_mean, float* moving_var, float eps, float momentum, int update_moving,
                                   int unbiased_moving, float* scale, float* shift, float* mean, float* rstd,
                                   void* workspace, size_t workspace_bytes, int* counter, void* stream) {
    PTTS_REQUIRE(x && scale && shift && counter, "bn_batch_stats: null pointer");
    PTTS_REQUIRE(ptts_bn_batch_stats_supported(rows, C) && al16(x), "bn_batch_stats: unsupported shape rows=%lld C=%d (or x not 16-byte aligned)", rows, C);
    PTTS_REQUIRE(!update_moving || (moving_mean && moving_var), "bn_batch_stats: update needs moving stats");
    // at most one workgroup per CU: every workgroup ends with an atomic on ONE counter, and 813 of them (the row count of the
    // two-stage kernel) queue up for 16 us there (29 us a launch; 17 with 256, against 17 + 6 for the two-stage path)
    int nb = colreduce_vec4_blocks(rows, C);
    static int cap = -1;
    if (cap < 0) { const char* e = getenv("PTTS_BN_FUSED_BLOCKS"); cap = e ? atoi(e) : 256; }
    if (nb > cap) nb = cap;
    const size_t need = (size_t)nb * 2 * C * sizeof(double);
    if (!workspace || workspace_bytes < need) { set_error("bn_batch_stats: workspace %zu < %zu", workspace_bytes, need); return PTTS_EWORKSPACE; }
    hipLaunchKernelGGL(bn_stats_fused_kernel, dim3(nb), dim3(EW_THREADS), 0, (hipStream_t)stream, x, rows, C, (double*)workspace, counter,
                       gamma, beta, moving_mean, moving_var, eps, momentum, update_moving, unbiased_moving, scale, shift, mean, rstd);
    return check_launch("bn_batch_stats");
}

// out[j] = sum over the nrows rows of partials[nrows][ncols] (doubles), in index order: the finish of sums another kernel left per
// workgroup (ptts_dense_bf16x6_bwd_affine)
extern "C" int ptts_partial_rows_sum(const double* partials, int nrows, int ncols, double* out, void* stream) {
    PTTS_REQUIRE(partials && out && nrows > 0 && ncols > 0, "partial_rows_sum: bad args");
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((ncols + 7) / 8), dim3(256), 0, (hipStream_t)stream, partials, nrows, ncols, out);
    return check_launch("partial_rows_sum");
}

extern "C" int ptts_bn_finalize_partials(const double* partials, int nrows, long long rows, int C, const float* gamma, const float* beta,
                                         float* moving_mean, float* moving_var, float eps, float momentum, int update_moving,
                                         int unbiased_moving, float* scale, float* shift, float* mean, float* rstd, void* stream) {
    PTTS_REQUIRE(partials && scale && shift && nrows > 0 && rows > 0, "bn_finalize_partials: bad args");
    PTTS_REQUIRE(C >= 1, "bn_finalize_partials: C = %d", C);
    PTTS_REQUIRE(!update_moving || (moving_mean && moving_var), "bn_finalize_partials: update needs moving stats");
    if (C <= 16)
        hipLaunchKernelGGL(bn_finalize_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nrows, rows, C, gamma, beta,
                           moving_mean, moving_var, eps, momentum, update_moving, unbiased_moving, scale, shift, mean, rstd);
    else
        hipLaunchKernelGGL(bn_finalize_partials_wide_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, partials, nrows, rows, C,
                           gamma, beta, moving_mean, moving_var, eps, momentum, update_moving, unbiased_moving, scale, shift, mean, rstd);
    return check_launch("bn_finalize_partials");
}

extern "C" int ptts_bn_bwd_coefs(const float* dscale, const float* dshift, const float* mean, const float* rstd,
                                 const float* gamma, long long count, int C, float* dgamma, float* dbeta,
                                 float* c0, float* c2, void* stream) {
    PTTS_REQUIRE(dscale && dshift && mean && rstd && c0 && c2 && C > 0 && count > 0, "bn_bwd_coefs: bad args");
    hipLaunchKernelGGL(bn_bwd_coefs_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, dscale, dshift,
                       mean, rstd, gamma, count, C, dgamma, dbeta, c0, c2, 0);
    return check_launch("bn_bwd_coefs");
}

// the same with dgamma / dbeta ADDED to what the buffers hold: the parameters' gradient buffers themselves (no per-parameter add launch)
extern "C" int ptts_bn_bwd_coefs_acc(const float* dscale, const float* dshift, const float* mean, const float* rstd,
                                     const float* gamma, long long count, int C, float* dgamma, float* dbeta,
                                     float* c0, float* c2, void* stream) {
    PTTS_REQUIRE(dscale && dshift && mean && rstd && c0 && c2 && C > 0 && count > 0, "bn_bwd_coefs_acc: bad args");
    hipLaunchKernelGGL(bn_bwd_coefs_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, dscale, dshift,
                       mean, rstd, gamma, count, C, dgamma, dbeta, c0, c2, 1);
    return check_launch("bn_bwd_coefs_acc");
}

extern "C" int ptts_affine_act(const float* x, const float* scale, const float* shift, float* y, long long rows,
                               int C, int act, float alpha, void* stream) {
    PTTS_REQUIRE(x && y && rows > 0 && C > 0, "affine_act: bad args");
    PTTS_REQUIRE((scale == nullptr) == (shift == nullptr), "affine_act: scale/shift must come together");
    const long long n = rows * C;
    hipStream_t st = (hipStream_t)stream;
    if (C % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0)) {
        hipLaunchKernelGGL(affine_act_kernel4, dim3(ew_blocks(n / 4, 2)), dim3(EW_THREADS), 0, st, (const float4*)x,
                           scale, shift, (float4*)y, n / 4, C, act, alpha);
    } else {
        hipLaunchKernelGGL(affine_act_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, st, x, scale, shift, y, n, C,
                           act, alpha);
    }
    return check_launch("affine_act");
}

// ---- gated product of a gated convolution: y = a * sigmoid(b) (reference networktts.py:128-134, pGCNN2D: the second
// Conv2D carries activation=sigmoid and kl.Multiply joins the two).  Both pre-activations are read once; neither the
// sigmoid nor the product's operands are written.  HBM-bound: 12 B per element forward, 20 B backward.
__global__ __launch_bounds__(EW_THREADS) void gated_mul_fwd_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                                 float4* __restrict__ y, long long n4, const float* __restrict__ at,
                                                                 const float* __restrict__ bt, float* __restrict__ yt, int tail) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 av = a[i], bv = b[i];
        float4 o;
        o.x = av.x / (1.f + __expf(-bv.x)); o.y = av.y / (1.f + __expf(-bv.y));
        o.z = av.z / (1.f + __expf(-bv.z)); o.w = av.w / (1.f + __expf(-bv.w));
        y[i] = o;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) yt[threadIdx.x] = at[threadIdx.x] / (1.f + __expf(-bt[threadIdx.x]));
}

__device__ __forceinline__ void gated_bwd1(float dy, float a, float b, float& da, float& db) {
    const float s = 1.f / (1.f + __expf(-b));
    da = dy * s;
    db = dy * a * s * (1.f - s);
}

__global__ __launch_bounds__(EW_THREADS) void gated_mul_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ a,
                                                                 const float4* __restrict__ b, float4* __restrict__ da,
                                                                 float4* __restrict__ db, long long n4,
                                                                 const float* __restrict__ dyt, const float* __restrict__ at,
                                                                 const float* __restrict__ bt, float* __restrict__ dat,
                                                                 float* __restrict__ dbt, int tail) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 g = dy[i], av = a[i], bv = b[i];
        float4 oa, ob;
        gated_bwd1(g.x, av.x, bv.x, oa.x, ob.x); gated_bwd1(g.y, av.y, bv.y, oa.y, ob.y);
        gated_bwd1(g.z, av.z, bv.z, oa.z, ob.z); gated_bwd1(g.w, av.w, bv.w, oa.w, ob.w);
        da[i] = oa; db[i] = ob;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) gated_bwd1(dyt[threadIdx.x], at[threadIdx.x], bt[threadIdx.x], dat[threadIdx.x], dbt[threadIdx.x]);
}

// ---- frequency-domain context Conv1D (ops._C1FFT): the second half of the per-frequency left operand -----------------------------
// Ap [NB][2][B][2 Kh] holds [Xr | .] in the rows of part 0 and [Xi | .] in the rows of part 1 (columns < Cin, written by the DFT
// product); the complex product (Xr + i Xi)(Wr + i Wi) as ONE real product against [Wr; Wi] needs the rows [Xr | -Xi] and [Xi | Xr].
__global__ __launch_bounds__(256) void dft_mirror_kernel(float* __restrict__ Ap, long long total, int B, int Cin, int Kh) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % Cin);
        const long long fb = i / Cin;
        const int b = (int)(fb % B);
        const long long f = fb / B;
        float* r0 = Ap + ((f * 2 + 0) * B + b) * (2LL * Kh);
        float* r1 = Ap + ((f * 2 + 1) * B + b) * (2LL * Kh);
        const float xr = r0[c], xi = r1[c];
        r0[Kh + c] = -xi;
        r1[Kh + c] = xr;
    }
}

// dst[z][c][r] = src[z][r][c] for nb matrices of rows x cols floats (32 x 32 tiles through the LDS, both sides coalesced)
__global__ __launch_bounds__(256) void transpose_batched_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
    __shared__ float tile[32][33];
    const long long zoff = (long long)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 8 * i][tx] = src[zoff + (long long)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < rows && c < cols) dst[zoff + (long long)c * rows + r] = tile[tx][ty + 8 * i];
    }
}

extern "C" int ptts_transpose_batched(const float* src, float* dst, int nb, int rows, int cols, void* stream) {
    PTTS_REQUIRE(src && dst && nb > 0 && nb <= 65535 && rows > 0 && cols > 0, "transpose_batched: bad arguments");
    hipLaunchKernelGGL(transpose_batched_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, nb), dim3(256), 0, (hipStream_t)stream, src, dst, rows, cols);
    return check_launch("transpose_batched");
}

// frequency-domain context Conv1D, weight gradient: dW[k][c][n] = out2[k][n 2Kh + c] + out2[KW + k][n 2Kh + Kh + c]
// (out2 = the inverse-transform product over the frequencies, rows: cosine / sine coefficients of tap k)
__global__ __launch_bounds__(256) void conv1d_freq_wgrad_combine_kernel(const float* __restrict__ out2, float* __restrict__ dW, int KW,
                                                                       int Cin, int N, int Kh) {
    // tile: 32 n x 32 c of one tap; reads contiguous along c, writes contiguous along n
    __shared__ float tile[32][33];
    const int k = blockIdx.z, c0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long long row = (long long)N * 2 * Kh;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + 8 * i, c = c0 + tx;
        if (n < N && c < Cin) {
            const long long j = (long long)n * 2 * Kh + c;
            tile[ty + 8 * i][tx] = out2[(long long)k * row + j] + out2[(long long)(KW + k) * row + j + Kh];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, n = n0 + tx;
        if (n < N && c < Cin) dW[((long long)k * Cin + c) * N + n] = tile[tx][ty + 8 * i];
    }
}

// The inverse transform of the per-frequency correlations and the combination above in ONE pass over Gt [NB][N][2 Kh]:
//     dW[k][c][n] = sum_f  t2[k][f] Gt[f][n][c] + t2[KW + k][f] Gt[f][n][Kh + c]
// A lane owns four consecutive c of one n and all KW taps (4 KW accumulators); per frequency two 16-byte loads, coalesced along c;
// FU frequencies' loads are in flight together.  (As a GEMM with M = 2 KW = 42 rows the matrix kernels reached 1.2 TB/s on the
// 263 MB of Gt: 223 us + 12 for the combination.)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KW, int FU>
__global__ __launch_bounds__(256) void conv1d_freq_wgrad_inverse_kernel(const float* __restrict__ Gt, const float* __restrict__ t2,
                                                                       float* __restrict__ dW, int NB, int NBp, int Cin, int N, int Kh) {
    // workgroup tile: 16 n x 16 quads of c (64 c); lanes along c for the loads (256-byte runs), along n for the stores (below)
    __shared__ float tile[64][17];
    const int q4 = Kh / 4, qt = (q4 + 15) / 16;
    const int nt = blockIdx.x / qt, ct = blockIdx.x - nt * qt;
    const int tn = threadIdx.x >> 4, tq = threadIdx.x & 15;
    const int n = nt * 16 + tn, cq = ct * 16 + tq;
    const bool live = n < N && cq < q4;
    const int c = min(cq, q4 - 1) * 4;
    const float* __restrict__ g0 = Gt + (long long)min(n, N - 1) * 2 * Kh + c;
    const long long fs = (long long)N * 2 * Kh;
    f32x4 acc[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the frequencies are shared out over gridDim.y workgroup rows (608 waves for all of them left three SIMDs in four idle);
    // each writes its partial sums, wgrad_partials_sum_kernel adds them in a fixed order
    const int per = (NB + gridDim.y - 1) / gridDim.y;
    const int fbeg = blockIdx.y * per, fend = min(NB, fbeg + per);
    dW += (long long)blockIdx.y * KW * Cin * N;
    for (int f0 = fbeg; f0 < fend; f0 += FU) {
        f32x4 gr[FU], gi[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int f = min(f0 + u, fend - 1);
            gr[u] = *reinterpret_cast<const f32x4*>(g0 + f * fs);
            gi[u] = *reinterpret_cast<const f32x4*>(g0 + f * fs + Kh);
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            if (f0 + u >= fend) break;                     // wave-uniform
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                // t2 [f][TP]: the 2 KW coefficients of a frequency lie together (uniform address: a few wide scalar loads per f)
                const float tc = t2[(long long)(f0 + u) * NBp + k], ts = t2[(long long)(f0 + u) * NBp + KW + k];
                acc[k] += gr[u] * tc + gi[u] * ts;
            }
        }
    }
    (void)live;
    // dW[k][c][n]: through the LDS, so that a 16-lane group writes 16 consecutive n of one c (scattered 4-byte stores -- 4 KB apart
    // from lane to lane -- cost 0.35 ms here)
    const int wn = threadIdx.x & 15, wc = threadIdx.x >> 4;
#pragma unroll
    for (int k = 0; k < KW; ++k) {          // (unrolled: a run-time index into acc would put it into scratch)
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[tq * 4 + e][tn] = acc[k][e];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cl = wc + 16 * i, cc = ct * 64 + cl, nn = nt * 16 + wn;
            if (cc < Cin && nn < N) dW[((long long)k * Cin + cc) * N + nn] = tile[cl][wn];
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_partials_sum_kernel(const float* __restrict__ part, float* __restrict__ out, long long n, int parts) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float s = part[i];
        for (int p = 1; p < parts; ++p) s += part[(long long)p * n + i];
        out[i] = s;
    }
}

constexpr int WINV_SPLIT = 4;
extern "C" size_t ptts_conv1d_freq_wgrad_inverse_workspace_bytes(int KW, int Cin, int N) {
    return (size_t)WINV_SPLIT * KW * Cin * N * sizeof(float);
}

extern "C" int ptts_conv1d_freq_wgrad_inverse(const float* Gt, const float* t2, float* dW, void* workspace, size_t workspace_bytes,
                                              int NB, int NBp, int KW, int Cin, int N, int Kh, void* stream) {
    PTTS_REQUIRE(Gt && t2 && dW && NB > 0 && NBp >= 2 * KW && Cin > 0 && N > 0 && Kh >= Cin && Kh % 4 == 0, "conv1d_freq_wgrad_inverse: bad arguments");
    PTTS_REQUIRE(((uintptr_t)Gt & 15) == 0, "conv1d_freq_wgrad_inverse: Gt must be 16-byte aligned");
    const size_t need = ptts_conv1d_freq_wgrad_inverse_workspace_bytes(KW, Cin, N);
    if (!workspace || workspace_bytes < need) { set_error("conv1d_freq_wgrad_inverse: workspace %zu < %zu", workspace_bytes, need); return PTTS_EWORKSPACE; }
    const dim3 grid((unsigned)(((N + 15) / 16) * ((Kh / 4 + 15) / 16)), WINV_SPLIT);
    float* part = (float*)workspace;
#define WINV(KWv) hipLaunchKernelGGL((conv1d_freq_wgrad_inverse_kernel<KWv, 8>), grid, dim3(256), 0, (hipStream_t)stream, Gt, t2, part, NB, NBp, Cin, N, Kh)
    switch (KW) {
        case 3: WINV(3); break;
        case 5: WINV(5); break;
        case 7: WINV(7); break;
        case 9: WINV(9); break;
        case 11: WINV(11); break;
        case 21: WINV(21); break;
        default: set_error("conv1d_freq_wgrad_inverse: KW=%d not instantiated (3, 5, 7, 9, 11, 21)", KW); return PTTS_EINVAL;
    }
#undef WINV
    const long long nw = (long long)KW * Cin * N;
    long long blocks = (nw + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_partials_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, part, dW, nw, WINV_SPLIT);
    return check_launch("conv1d_freq_wgrad_inverse");
}

extern "C" int ptts_conv1d_freq_wgrad_combine(const float* out2, float* dW, int KW, int Cin, int N, int Kh, void* stream) {
    PTTS_REQUIRE(out2 && dW && KW > 0 && Cin > 0 && N > 0 && Kh >= Cin, "conv1d_freq_wgrad_combine: bad arguments");
    hipLaunchKernelGGL(conv1d_freq_wgrad_combine_kernel, dim3((Cin + 31) / 32, (N + 31) / 32, KW), dim3(256), 0, (hipStream_t)stream,
                       out2, dW, KW, Cin, N, Kh);
    return check_launch("conv1d_freq_wgrad_combine");
}

extern "C" int ptts_dft_mirror(float* Ap, int NB, int B, int Cin, int Kh, void* stream) {
    PTTS_REQUIRE(Ap && NB > 0 && B > 0 && Cin > 0 && Kh >= Cin, "dft_mirror: bad arguments");
    const long long total = (long long)NB * B * Cin;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(dft_mirror_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, Ap, total, B, Cin, Kh);
    return check_launch("dft_mirror");
}

extern "C" int ptts_gated_mul_fwd(const float* a, const float* b, float* y, long long n, void* stream) {
    PTTS_REQUIRE(a && b && y && n > 0, "gated_mul_fwd: bad args");
    PTTS_REQUIRE(al16(a) && al16(b) && al16(y), "gated_mul_fwd: tensors must be 16-byte aligned");
    const long long n4 = n / 4;
    const int tail = (int)(n - 4 * n4);
    hipLaunchKernelGGL(gated_mul_fwd_kernel, dim3(ew_blocks(n4 > 0 ? n4 : 1, 2)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       (const float4*)a, (const float4*)b, (float4*)y, n4, a + 4 * n4, b + 4 * n4, y + 4 * n4, tail);
    return check_launch("gated_mul_fwd");
}

extern "C" int ptts_gated_mul_bwd(const float* dy, const float* a, const float* b, float* da, float* db, long long n,
                                  void* stream) {
    PTTS_REQUIRE(dy && a && b && da && db && n > 0, "gated_mul_bwd: bad args");
    PTTS_REQUIRE(al16(dy) && al16(a) && al16(b) && al16(da) && al16(db), "gated_mul_bwd: tensors must be 16-byte aligned");
    const long long n4 = n / 4;
    const int tail = (int)(n - 4 * n4);
    hipLaunchKernelGGL(gated_mul_bwd_kernel, dim3(ew_blocks(n4 > 0 ? n4 : 1, 2)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       (const float4*)dy, (const float4*)a, (const float4*)b, (float4*)da, (float4*)db, n4, dy + 4 * n4,
                       a + 4 * n4, b + 4 * n4, da + 4 * n4, db + 4 * n4, tail);
    return check_launch("gated_mul_bwd");
}

extern "C" int ptts_affine_act_bwd(const float* dy, const float* x, const float* y, const float* scale,
                                   const float* shift, float* dx, double* dsums, void* workspace,
                                   size_t workspace_bytes, long long rows, int C, int act, float alpha,
                                   void* stream) {
    PTTS_REQUIRE(dy && x && rows > 0 && C > 0, "affine_act_bwd: bad args");
    PTTS_REQUIRE((scale == nullptr) == (shift == nullptr), "affine_act_bwd: scale/shift must come together");
    PTTS_REQUIRE((act != PTTS_ACT_SIGMOID && act != PTTS_ACT_TANH) || y, "affine_act_bwd: sigmoid/tanh need y");
    PTTS_REQUIRE(dsums, "affine_act_bwd: dsums required (the reduction is fused with the dx pass)");
    if (C % 4 == 0 && C <= 4 * EW_THREADS && al16(dy) && al16(x) && al16(y) && al16(dx)) {
        ActBwdOp4 op4{dy, x, y, scale, shift, dx, act, alpha};
        return run_colreduce_vec4(op4, rows, C, dsums, workspace, workspace_bytes, (hipStream_t)stream, "affine_act_bwd");
    }
    ActBwdOp op{dy, x, y, scale, shift, dx, act, alpha};
    return run_colreduce(op, rows, C, dsums, workspace, workspace_bytes, (hipStream_t)stream, "affine_act_bwd");
}

extern "C" int ptts_axpby_cols(const float* a, const float* c1, const float* x, const float* c2, const float* c0,
                               float* out, long long rows, int C, void* stream) {
    PTTS_REQUIRE(out && rows > 0 && C > 0, "axpby_cols: bad args");
    const long long n = rows * C;
    hipLaunchKernelGGL(axpby_cols_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, a, c1, x, c2,
                       c0, out, n, C);
    return check_launch("axpby_cols");
}

extern "C" int ptts_gp_interpolate(const float* real, const float* fake, const float* alpha_b, float* out, int B,
                                   long long TD, void* stream) {
    PTTS_REQUIRE(real && fake && alpha_b && out && B > 0 && B <= 65535 && TD > 0, "gp_interpolate: bad args");
    int bx = (int)((TD + 1023) / 1024);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(gp_interpolate_kernel, dim3(bx, B), dim3(EW_THREADS), 0, (hipStream_t)stream, real, fake,
                       alpha_b, out, TD);
    return check_launch("gp_interpolate");
}

extern "C" int ptts_gp_sqnorm(const float* g, float* out, int B, long long TD, void* stream) {
    PTTS_REQUIRE(g && out && B > 0 && TD > 0, "gp_sqnorm: bad args");
    hipLaunchKernelGGL(gp_sqnorm_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, g, out, TD);
    return check_launch("gp_sqnorm");
}

extern "C" int ptts_gp_penalty(const float* sq, float* penalty, float* coef, int B, void* stream) {
    PTTS_REQUIRE(sq && penalty && B > 0, "gp_penalty: bad args");
    hipLaunchKernelGGL(gp_penalty_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sq, penalty, coef, B);
    return check_launch("gp_penalty");
}

extern "C" int ptts_gp_scale_rows(const float* g, const float* coef, const float* upstream, float* dg, int B,
                                  long long TD, void* stream) {
    PTTS_REQUIRE(g && coef && dg && B > 0 && B <= 65535 && TD > 0, "gp_scale_rows: bad args");
    int bx = (int)((TD + 1023) / 1024);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(gp_scale_rows_kernel, dim3(bx, B), dim3(EW_THREADS), 0, (hipStream_t)stream, g, coef,
                       upstream, dg, TD);
    return check_launch("gp_scale_rows");
}

extern "C" int ptts_mean_scaled(const float* v, long long n, float sign, float* out, void* stream) {
    PTTS_REQUIRE(v && out && n > 0, "mean_scaled: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (zero_f32(out, 1, st) != PTTS_OK) return PTTS_ELAUNCH;
    int nb = (int)((n + 4095) / 4096);
    if (nb > 256) nb = 256;
    if (deterministic()) nb = 1;      // one workgroup: one float add into the zeroed scalar
    hipLaunchKernelGGL(mean_scaled_kernel, dim3(nb), dim3(256), 0, st, v, n, sign / (float)n, out);
    return check_launch("mean_scaled");
}

extern "C" int ptts_wlse_fwd(const float* y, const float* yhat, const float* w, float* out, long long rows, int D,
                             void* stream) {
    PTTS_REQUIRE(y && yhat && out && rows > 0 && D > 0, "wlse_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (zero_f32(out, 1, st) != PTTS_OK) return PTTS_ELAUNCH;
    const long long n = rows * D;
    int nb = (int)((n + 4095) / 4096);
    if (nb > 512) nb = 512;
    if (deterministic()) nb = 1;
    hipLaunchKernelGGL(wlse_fwd_kernel, dim3(nb), dim3(256), 0, st, y, yhat, w, n, D, 1.f / (float)n, out);
    return check_launch("wlse_fwd");
}

extern "C" int ptts_wlse_bwd(const float* y, const float* yhat, const float* w, const float* upstream,
                             float* dyhat, long long rows, int D, void* stream) {
    PTTS_REQUIRE(y && yhat && dyhat && rows > 0 && D > 0, "wlse_bwd: bad args");
    const long long n = rows * D;
    hipLaunchKernelGGL(wlse_bwd_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, y, yhat, w,
                       upstream, dyhat, n, D, 1.f / (float)n);
    return check_launch("wlse_bwd");
}

extern "C" int ptts_weight_clip(float* p, long long n, float lo, float hi, void* stream) {
    PTTS_REQUIRE(p && n > 0 && lo <= hi, "weight_clip: bad args");
    hipLaunchKernelGGL(weight_clip_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, p, n, lo, hi);
    return check_launch("weight_clip");
}

extern "C" int ptts_adam_keras_step(float* p, const float* g, float* m, float* v, long long n, float lr, float b1,
                                    float b2, float eps, float gscale, int* step, void* stream) {
    PTTS_REQUIRE(p && g && m && v && step && n > 0, "adam: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, step);
    int rc = check_launch("adam_tick");
    if (rc) return rc;
    hipLaunchKernelGGL(adam_keras_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, st, p, g, m, v, n, lr, b1, b2,
                       eps, gscale, (const int*)step);
    return check_launch("adam");
}

// Keras LSTM recurrence (gates i,f,c,o; tanh / sigmoid), both directions of a Bidirectional
// wrapper advanced by the same launches.  Reference: networktts.py:72-96 (pLSTM/pBLSTM), used by
// the generator's f0 branch (modeltts_common.py:82-84).
//
// The input projection x.W+b and every weight gradient are large GEMMs done by ptts_gemm; what
// is left here is the T-sequential part: one small launch per time step that adds h_{t-1}.U,
// applies the gate non-linearities and advances (c, h).  Each lane owns one (sample, unit)
// pair and all four of its gates, h_{t-1} of the workgroup's samples is staged in LDS, U is
// read coalesced along the unit index (it stays L2-resident: 1 MiB per direction at H = 256).
// Round-1 form: generic in (B, H); an MFMA 16x16x4 tile version is the planned replacement.
#include "common.h"
#include <cstdlib>
#include <mutex>
#include <vector>

namespace ptts {

constexpr int LX = 64, LY = 4;   // lanes along units, samples per workgroup

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ __launch_bounds__(LX * LY) void lstm_fwd_step_kernel(
    const float* __restrict__ xproj, const float* __restrict__ U, float* __restrict__ h_out,
    float* __restrict__ gates, float* __restrict__ c_out, int B, int T, int H, int ndir, int reverse, int s) {
    extern __shared__ float hs[];   // [LY][H]
    const int d = blockIdx.z;
    const bool rev = ndir == 2 ? d == 1 : reverse != 0;
    const int t = rev ? T - 1 - s : s;
    const int tp = rev ? t + 1 : t - 1;
    const int tid = threadIdx.y * LX + threadIdx.x;
    const long long HH = (long long)ndir * H;
    if (s > 0) {
        for (int idx = tid; idx < LY * H; idx += LX * LY) {
            const int by = idx / H, k = idx - by * H;
            const int b = blockIdx.y * LY + by;
            hs[idx] = b < B ? h_out[((long long)b * T + tp) * HH + (long long)d * H + k] : 0.f;
        }
        __syncthreads();
    }
    const int j = blockIdx.x * LX + threadIdx.x;
    const int b = blockIdx.y * LY + threadIdx.y;
    if (j >= H || b >= B) return;
    const long long G4 = 4LL * H;
    const float* xp = xproj + (((long long)b * T + t) * ndir + d) * G4;
    float a0 = xp[j], a1 = xp[H + j], a2 = xp[2 * H + j], a3 = xp[3 * H + j];
    if (s > 0) {
        const float* Ud = U + (long long)d * H * G4 + j;
        const float* hrow = hs + threadIdx.y * H;
#pragma unroll 4
        for (int k = 0; k < H; ++k) {
            const float hk = hrow[k];
            const float* u = Ud + (long long)k * G4;
            a0 = fmaf(hk, u[0], a0);
            a1 = fmaf(hk, u[H], a1);
            a2 = fmaf(hk, u[2 * H], a2);
            a3 = fmaf(hk, u[3 * H], a3);
        }
    }
    const float gi = sigmoidf_(a0), gf = sigmoidf_(a1), gc = tanhf(a2), go = sigmoidf_(a3);
    const long long so = ((long long)b * T + t) * HH + (long long)d * H + j;
    const float cp = s > 0 ? c_out[((long long)b * T + tp) * HH + (long long)d * H + j] : 0.f;
    const float c = gf * cp + gi * gc;
    c_out[so] = c;
    h_out[so] = go * tanhf(c);
    float* gp = gates + (((long long)b * T + t) * ndir + d) * G4;
    gp[j] = gi; gp[H + j] = gf; gp[2 * H + j] = gc; gp[3 * H + j] = go;
}

// UT[d][n][k] = U[d][k][n]
__global__ void lstm_transpose_kernel(const float* __restrict__ U, float* __restrict__ UT, int H, int ndir) {
    const long long n4 = 4LL * H;
    const long long total = (long long)ndir * H * n4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % H);
        const long long r = i / H;
        const int n = (int)(r % n4);
        const int d = (int)(r / n4);
        UT[i] = U[((long long)d * H + k) * n4 + n];
    }
}

// backward of forward step s (launched for s = T-1 .. 0)
__global__ __launch_bounds__(LX * LY) void lstm_bwd_step_kernel(
    const float* __restrict__ dh_out, const float* __restrict__ UT, const float* __restrict__ gates,
    const float* __restrict__ c_out, float* __restrict__ dgates, float* __restrict__ dc_state, int B, int T,
    int H, int ndir, int reverse, int s) {
    extern __shared__ float das[];   // [LY][4H]: gate-preactivation grads of forward step s+1
    const int d = blockIdx.z;
    const bool rev = ndir == 2 ? d == 1 : reverse != 0;
    const int t = rev ? T - 1 - s : s;
    const int tp = rev ? t + 1 : t - 1;   // time of forward step s-1
    const int tn = rev ? t - 1 : t + 1;   // time of forward step s+1
    const int tid = threadIdx.y * LX + threadIdx.x;
    const long long G4 = 4LL * H;
    const long long HH = (long long)ndir * H;
    const bool has_next = s < T - 1;
    if (has_next) {
        for (int idx = tid; idx < LY * (int)G4; idx += LX * LY) {
            const int by = idx / (int)G4, n = idx - by * (int)G4;
            const int b = blockIdx.y * LY + by;
            das[idx] = b < B ? dgates[(((long long)b * T + tn) * ndir + d) * G4 + n] : 0.f;
        }
        __syncthreads();
    }
    const int j = blockIdx.x * LX + threadIdx.x;
    const int b = blockIdx.y * LY + threadIdx.y;
    if (j >= H || b >= B) return;
    float dh = dh_out[((long long)b * T + t) * HH + (long long)d * H + j];
    if (has_next) {
        const float* ut = UT + (long long)d * G4 * H + j;
        const float* drow = das + threadIdx.y * G4;
        float acc = 0.f;
#pragma unroll 4
        for (int n = 0; n < (int)G4; ++n) acc = fmaf(drow[n], ut[(long long)n * H], acc);
        dh += acc;
    }
    const float* gp = gates + (((long long)b * T + t) * ndir + d) * G4;
    const float gi = gp[j], gf = gp[H + j], gc = gp[2 * H + j], go = gp[3 * H + j];
    const long long so = ((long long)b * T + t) * HH + (long long)d * H + j;
    const float c = c_out[so];
    const float cp = s > 0 ? c_out[((long long)b * T + tp) * HH + (long long)d * H + j] : 0.f;
    const float tc = tanhf(c);
    const long long si = ((long long)d * B + b) * H + j;
    float dc = dh * go * (1.f - tc * tc);
    if (has_next) dc += dc_state[si];
    float* dg = dgates + (((long long)b * T + t) * ndir + d) * G4;
    dg[j] = dc * gc * gi * (1.f - gi);
    dg[H + j] = dc * cp * gf * (1.f - gf);
    dg[2 * H + j] = dc * gi * (1.f - gc * gc);
    dg[3 * H + j] = dh * tc * go * (1.f - go);
    dc_state[si] = dc * gf;
}


// ------------------------------------------------------------------------------------------------
// MFMA step kernels (H % 16 == 0): v_mfma_f32_16x16x4_f32, K split over the 4 waves of a workgroup,
// partial tiles combined through LDS, then one lane per (sample, unit) applies the gate math.
//   forward : tile = 16 samples x 8 units x 4 gates (two MFMA column blocks: gates {i,f} and {c,o}),
//             grid (H/8, ceil(B/16), ndir) = 256 workgroups at H = 256, B = 64 -> every CU busy.
//   backward: tile = 16 samples x 16 units of dh_rec = da_next . U^T, grid (H/16, ceil(B/16), ndir).
// ------------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void lstm_fwd_step_mfma_kernel(
    const float* __restrict__ xproj, const float* __restrict__ U, float* __restrict__ h_out,
    float* __restrict__ gates, float* __restrict__ c_out, int B, int T, int H, int ndir, int reverse, int s) {
    __shared__ float red[4][2][256];
    const int d = blockIdx.z;
    const bool rev = ndir == 2 ? d == 1 : reverse != 0;
    const int t = rev ? T - 1 - s : s;
    const int tp = rev ? t + 1 : t - 1;
    const int j0 = blockIdx.x * 8, b0 = blockIdx.y * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const long long G4 = 4LL * H, HH = (long long)ndir * H;
    if (s > 0) {
        const int ksteps = H / 16;                 // k-steps of 4 per wave
        const int kbase = wave * (H / 4);
        const int b = b0 + r16;
        const float* hp = h_out + ((long long)(b < B ? b : 0) * T + tp) * HH + (long long)d * H + kbase + q;
        const float* Ud = U + (long long)d * H * G4 + (long long)(kbase + q) * G4 + j0 + (r16 & 7);
        const int gx = (r16 >> 3) * H, gy = (2 + (r16 >> 3)) * H;
        f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < ksteps; s0 += 8) {
            float a[8], bx[8], by[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int st = s0 + i;
                if (st < ksteps) {
                    a[i] = b < B ? hp[4 * st] : 0.f;
                    const float* u = Ud + (long long)(4 * st) * G4;
                    bx[i] = u[gx];
                    by[i] = u[gy];
                } else { a[i] = 0.f; bx[i] = 0.f; by[i] = 0.f; }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bx[i], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], by[i], acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[wave][0][(q * 4 + r) * 16 + r16] = acc0[r];
            red[wave][1][(q * 4 + r) * 16 + r16] = acc1[r];
        }
        __syncthreads();
    }
    if (tid >= 128) return;
    const int bb = tid >> 3, jj = tid & 7;
    const int b = b0 + bb, j = j0 + jj;
    if (b >= B) return;
    const float* xp = xproj + (((long long)b * T + t) * ndir + d) * G4;
    float a4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float v = xp[g * H + j];
        if (s > 0) {
            const int idx = bb * 16 + (g & 1) * 8 + jj;
            v += red[0][g >> 1][idx] + red[1][g >> 1][idx] + red[2][g >> 1][idx] + red[3][g >> 1][idx];
        }
        a4[g] = v;
    }
    const float gi = sigmoidf_(a4[0]), gf = sigmoidf_(a4[1]), gc = tanhf(a4[2]), go = sigmoidf_(a4[3]);
    const long long so = ((long long)b * T + t) * HH + (long long)d * H + j;
    const float cp = s > 0 ? c_out[((long long)b * T + tp) * HH + (long long)d * H + j] : 0.f;
    const float c = gf * cp + gi * gc;
    c_out[so] = c;
    h_out[so] = go * tanhf(c);
    float* gp = gates + (((long long)b * T + t) * ndir + d) * G4;
    gp[j] = gi; gp[H + j] = gf; gp[2 * H + j] = gc; gp[3 * H + j] = go;
}

__global__ __launch_bounds__(256) void lstm_bwd_step_mfma_kernel(
    const float* __restrict__ dh_out, const float* __restrict__ UT, const float* __restrict__ gates,
    const float* __restrict__ c_out, float* __restrict__ dgates, float* __restrict__ dc_state, int B, int T,
    int H, int ndir, int reverse, int s) {
    __shared__ float red[4][256];
    const int d = blockIdx.z;
    const bool rev = ndir == 2 ? d == 1 : reverse != 0;
    const int t = rev ? T - 1 - s : s;
    const int tp = rev ? t + 1 : t - 1;
    const int tn = rev ? t - 1 : t + 1;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const long long G4 = 4LL * H, HH = (long long)ndir * H;
    const bool has_next = s < T - 1;
    if (has_next) {
        const int nsteps = H / 4;                  // n-steps of 4 per wave (each wave owns H of the 4H gate columns)
        const int nbase = wave * H;
        const int b = b0 + r16;
        const float* ap = dgates + (((long long)(b < B ? b : 0) * T + tn) * ndir + d) * G4 + nbase + q;
        const float* bp = UT + (long long)d * G4 * H + (long long)(nbase + q) * H + j0 + r16;
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < nsteps; s0 += 16) {
            float a[16], bv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int st = s0 + i;
                if (st < nsteps) {
                    a[i] = b < B ? ap[4 * st] : 0.f;
                    bv[i] = bp[(long long)(4 * st) * H];
                } else { a[i] = 0.f; bv[i] = 0.f; }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bv[i], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(q * 4 + r) * 16 + r16] = acc[r];
        __syncthreads();
    }
    const int bb = tid >> 4, jj = tid & 15;
    const int b = b0 + bb, j = j0 + jj;
    if (b >= B) return;
    float dh = dh_out[((long long)b * T + t) * HH + (long long)d * H + j];
    if (has_next) dh += red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    const float* gp = gates + (((long long)b * T + t) * ndir + d) * G4;
    const float gi = gp[j], gf = gp[H + j], gc = gp[2 * H + j], go = gp[3 * H + j];
    const long long so = ((long long)b * T + t) * HH + (long long)d * H + j;
    const float c = c_out[so];
    const float cp = s > 0 ? c_out[((long long)b * T + tp) * HH + (long long)d * H + j] : 0.f;
    const float tc = tanhf(c);
    const long long si = ((long long)d * B + b) * H + j;
    float dc = dh * go * (1.f - tc * tc);
    if (has_next) dc += dc_state[si];
    float* dg = dgates + (((long long)b * T + t) * ndir + d) * G4;
    dg[j] = dc * gc * gi * (1.f - gi);
    dg[H + j] = dc * cp * gf * (1.f - gf);
    dg[2 * H + j] = dc * gi * (1.f - gc * gc);
    dg[3 * H + j] = dh * tc * go * (1.f - go);
    dc_state[si] = dc * gf;
}


// ------------------------------------------------------------------------------------------------
// Packed-operand MFMA step kernels (H % 64 == 0): the recurrent kernel is re-laid once per call so that every
// lane's MFMA B operands of all k-steps are contiguous (16-byte loads, whole lines per wave), the k index of step
// `st` for lane group q is  k = wave*(K/4) + q*KS + st  so the lane's A operands (h / dgates rows) are contiguous
// too, and every global load of a step (operands AND the gate inputs of the epilogue) is issued before the first
// wait.  The step is latency-bound: this cuts it to one exposed L2 round trip.
// ------------------------------------------------------------------------------------------------
typedef float f32x4a __attribute__((ext_vector_type(4)));

// Upk[d][jt][w][lane][st][2] = { U[d][k][gate(i|f) col], U[d][k][gate(c|o) col] },  k = w*H/4 + q*KS + st
__global__ void lstm_pack_u_fwd_kernel(const float* __restrict__ U, float* __restrict__ Upk, int H, int ndir) {
    const int KS = H / 16, JT = H / 8;
    const long long total = (long long)ndir * JT * 4 * 64 * KS * 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int e = (int)(r % 2); r /= 2;
        const int st = (int)(r % KS); r /= KS;
        const int lane = (int)(r % 64); r /= 64;
        const int w = (int)(r % 4); r /= 4;
        const int jt = (int)(r % JT);
        const int d = (int)(r / JT);
        const int r16 = lane & 15, q = lane >> 4;
        const int k = w * (H / 4) + q * KS + st;
        const int col = ((e == 0 ? 0 : 2) + (r16 >> 3)) * H + jt * 8 + (r16 & 7);
        Upk[i] = U[((long long)d * H + k) * 4 * H + col];
    }
}

// UTpk[d][jt][w][lane][st] = U[d][jt*16 + r16][n],  n = w*H + q*(H/4) + st
__global__ void lstm_pack_u_bwd_kernel(const float* __restrict__ U, float* __restrict__ UTpk, int H, int ndir) {
    const int NS = H / 4, JT = H / 16;
    const long long total = (long long)ndir * JT * 4 * 64 * NS;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int st = (int)(r % NS); r /= NS;
        const int lane = (int)(r % 64); r /= 64;
        const int w = (int)(r % 4); r /= 4;
        const int jt = (int)(r % JT);
        const int d = (int)(r / JT);
        const int r16 = lane & 15, q = lane >> 4;
        const int n = w * H + q * NS + st;
        UTpk[i] = U[((long long)d * H + jt * 16 + r16) * 4 * H + n];
    }
}

template <int CH>   // k-steps per register chunk (4, 8 or 16)
__global__ __launch_bounds__(256) void lstm_fwd_step_pk_kernel(
    const float* __restrict__ xproj, const float* __restrict__ Upk, float* __restrict__ h_out,
    float* __restrict__ gates, float* __restrict__ c_out, int B, int T, int H, int ndir, int reverse, int s) {
    __shared__ float red[4][2][256];
    // a latency chain that shares its CUs with the wide kernels of the other streams: its waves go first at the issue arbiter
    __builtin_amdgcn_s_setprio(3);
    const int d = blockIdx.z;
    const bool rev = ndir == 2 ? d == 1 : reverse != 0;
    const int t = rev ? T - 1 - s : s;
    const int tp = rev ? t + 1 : t - 1;
    const int jt = blockIdx.x, j0 = jt * 8, b0 = blockIdx.y * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const long long G4 = 4LL * H, HH = (long long)ndir * H;
    const int KS = H / 16;
    // epilogue inputs first: their latency overlaps everything else
    const int bb = tid >> 3, jj = tid & 7;
    const int eb = b0 + bb, ej = j0 + jj;
    const bool epi = tid < 128 && eb < B;
    float xv[4] = {0.f, 0.f, 0.f, 0.f}, cp = 0.f;
    if (epi) {
        const float* xp = xproj + (((long long)eb * T + t) * ndir + d) * G4;
#pragma unroll
        for (int g = 0; g < 4; ++g) xv[g] = xp[g * H + ej];
        if (s > 0) cp = c_out[((long long)eb * T + tp) * HH + (long long)d * H + ej];
    }
    if (s > 0) {
        const int b = b0 + r16;
        const float* hp = h_out + ((long long)(b < B ? b : 0) * T + tp) * HH + (long long)d * H + wave * (H / 4) + q * KS;
        const float* up = Upk + ((((long long)d * (H / 8) + jt) * 4 + wave) * 64 + lane) * KS * 2;
        f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < KS; s0 += CH) {
            f32x4a hv[CH / 4], uv[CH / 2];
#pragma unroll
            for (int i = 0; i < CH / 4; ++i) hv[i] = *reinterpret_cast<const f32x4a*>(hp + s0 + 4 * i);
#pragma unroll
            for (int i = 0; i < CH / 2; ++i) uv[i] = *reinterpret_cast<const f32x4a*>(up + 2 * s0 + 4 * i);
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const float a = b < B ? hv[i / 4][i % 4] : 0.f;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, uv[i / 2][(i % 2) * 2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, uv[i / 2][(i % 2) * 2 + 1], acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[wave][0][(q * 4 + r) * 16 + r16] = acc0[r];
            red[wave][1][(q * 4 + r) * 16 + r16] = acc1[r];
        }
        __syncthreads();
    }
    if (!epi) return;
    float a4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float v = xv[g];
        if (s > 0) {
            const int idx = bb * 16 + (g & 1) * 8 + jj;
            v += red[0][g >> 1][idx] + red[1][g >> 1][idx] + red[2][g >> 1][idx] + red[3][g >> 1][idx];
        }
        a4[g] = v;
    }
    const float gi = sigmoidf_(a4[0]), gf = sigmoidf_(a4[1]), gc = tanhf(a4[2]), go = sigmoidf_(a4[3]);
    const long long so = ((long long)eb * T + t) * HH + (long long)d * H + ej;
    const float c = gf * cp + gi * gc;
    c_out[so] = c;
    h_out[so] = go * tanhf(c);
    float* gp = gates + (((long long)eb * T + t) * ndir + d) * G4;
    gp[ej] = gi; gp[H + ej] = gf; gp[2 * H + ej] = gc; gp[3 * H + ej] = go;
}

template <int NW>   // waves per workgroup: 4, or 8 (H = 256: the K = 4H reduction over eight waves -- half the dependent MFMA chain and
                    // half the load rounds per wave of this latency-bound step)
__global__ __launch_bounds__(64 * NW) void lstm_bwd_step_pk_kernel(
    const float* __restrict__ dh_out, const float* __restrict__ UTpk, const float* __restrict__ gates,
    const float* __restrict__ c_out, float* __restrict__ dgates, float* __restrict__ dc_state, int B, int T,
    int H, int ndir, int reverse, int s) {
    __shared__ float red[NW][256];
    __builtin_amdgcn_s_setprio(3);         // as in the forward step: ahead of the co-resident wide kernels' waves
    const int d = blockIdx.z;
    const bool rev = ndir == 2 ? d == 1 : reverse != 0;
    const int t = rev ? T - 1 - s : s;
    const int tp = rev ? t + 1 : t - 1;
    const int tn = rev ? t - 1 : t + 1;
    const int jt = blockIdx.x, j0 = jt * 16, b0 = blockIdx.y * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gate = wave & 3, part = wave >> 2;            // NW == 8: the two halves of a gate's quarter rows
    const int r16 = lane & 15, q = lane >> 4;
    const long long G4 = 4LL * H, HH = (long long)ndir * H;
    const bool has_next = s < T - 1;
    const int NS = H / 4, NSW = NS / (NW / 4);
    // epilogue inputs first
    const int bb = tid >> 4, jj = tid & 15;
    const int eb = b0 + bb, ej = j0 + jj;
    const bool epi = tid < 256 && eb < B;
    float dh = 0.f, gi = 0.f, gf = 0.f, gc = 0.f, go = 0.f, c = 0.f, cp = 0.f, dcs = 0.f;
    const long long si = ((long long)d * B + eb) * H + ej;
    if (epi) {
        dh = dh_out[((long long)eb * T + t) * HH + (long long)d * H + ej];
        const float* gp = gates + (((long long)eb * T + t) * ndir + d) * G4;
        gi = gp[ej]; gf = gp[H + ej]; gc = gp[2 * H + ej]; go = gp[3 * H + ej];
        c = c_out[((long long)eb * T + t) * HH + (long long)d * H + ej];
        if (s > 0) cp = c_out[((long long)eb * T + tp) * HH + (long long)d * H + ej];
        if (has_next) dcs = dc_state[si];
    }
    if (has_next) {
        const int b = b0 + r16;
        const float* ap = dgates + (((long long)(b < B ? b : 0) * T + tn) * ndir + d) * G4 + gate * H + q * NS + part * NSW;
        const float* bp = UTpk + ((((long long)d * (H / 16) + jt) * 4 + gate) * 64 + lane) * NS + part * NSW;
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < NSW; s0 += 16) {
            f32x4a av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                av[i] = *reinterpret_cast<const f32x4a*>(ap + s0 + 4 * i);
                bv[i] = *reinterpret_cast<const f32x4a*>(bp + s0 + 4 * i);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b < B ? av[i / 4][i % 4] : 0.f, bv[i / 4][i % 4], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(q * 4 + r) * 16 + r16] = acc[r];
        __syncthreads();
    }
    if (!epi) return;
    if (has_next) {
#pragma unroll
        for (int w = 0; w < NW; ++w) dh += red[w][tid];
    }
    const float tc = tanhf(c);
    float dc = dh * go * (1.f - tc * tc);
    if (has_next) dc += dcs;
    float* dg = dgates + (((long long)eb * T + t) * ndir + d) * G4;
    dg[ej] = dc * gc * gi * (1.f - gi);
    dg[H + ej] = dc * cp * gf * (1.f - gf);
    dg[2 * H + ej] = dc * gi * (1.f - gc * gc);
    dg[3 * H + ej] = dh * tc * go * (1.f - go);
    dc_state[si] = dc * gf;
}

static inline bool lstm_pk_ok(int H) { return H % 64 == 0 && ((H / 16) <= 16 ? true : (H / 16) % 16 == 0); }


// ------------------------------------------------------------------------------------------------
// Persistent forward recurrence (H = 256): ONE launch for all T steps.
//
// The recurrence of one (direction, 16-sample slice) -- a "group" -- involves 32 workgroups of 8 hidden units each (the
// tiling of lstm_fwd_step_pk_kernel); nothing couples different groups.  A workgroup keeps its slice of U (32 registers per
// lane) and its cell state c in registers for the whole sequence; what crosses workgroups per step is h_t: 512 bytes
// written, the group's 16 KB read.  It travels as DATA-TAGGED GRANULES (MI355X_MICROARCH.md, hand-off rows): every value is
// one naturally aligned 8-byte {h, step tag} written by ONE sc1 store and read by sc1 loads until the tag is the expected
// one -- no flag, no counter, no drain, no ordering to get wrong.  (A first version with sc1 payload + drained stores + a
// barrier + one agent-scope counter add per workgroup + a polled counter measured 7.0 us per step, slower than the 6.3 us of
// the per-step launches it replaces.)  Two granule buffers by step parity: a workgroup can be one step ahead of the slowest
// member of its group, never two (it needs that member's h first).  Group = blockIdx % groups, so with groups = 8 (B = 64,
// both directions) the 32 members of a group sit on one XCD (workgroups go to XCDs round-robin); any other placement is
// slower, not wrong.  A poll that does not match within ~2^21 rounds (seconds: a workgroup of the group never became
// resident) gives up for good: no hang, NaNs in the outputs from that step on AND STATUS_LSTM_HANDOFF in the device status word
// (common.h), so that ptts_lstm_fwd / ptts_device_status return PTTS_EDEVICE from the next call on instead of a silent bad step.
// ------------------------------------------------------------------------------------------------
struct LstmPersistArgs {
    const float* xproj; const float* Upk; float* h_out; float* gates; float* c_out; unsigned long long* xbuf;
    int B, T, ndir, reverse, groups;
    unsigned* status;
};

__global__ __launch_bounds__(256) void lstm_fwd_persistent_kernel(LstmPersistArgs a) {
    constexpr int H = 256, KS = 16;
    __shared__ float red[2][4][2][256];
    __shared__ volatile int give_up;
    const int g = blockIdx.x % a.groups, jt = blockIdx.x / a.groups;        // 32 members per group
    const int nslices = a.groups / a.ndir;
    const int d = g / nslices, b0 = (g - d * nslices) * 16, j0 = jt * 8;
    const bool rev = a.ndir == 2 ? d == 1 : a.reverse != 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int B = a.B, T = a.T, ndir = a.ndir;
    const long long G4 = 4LL * H, HH = (long long)ndir * H;
    if (tid == 0) give_up = 0;
    // this lane's slice of U: k = wave * 64 + q * 16 + st, columns (gate pair, unit) = r16
    f32x4a uv[KS / 2];
    {
        const float* up = a.Upk + ((((long long)d * (H / 8) + jt) * 4 + wave) * 64 + lane) * KS * 2;
#pragma unroll
        for (int i = 0; i < KS / 2; ++i) uv[i] = *reinterpret_cast<const f32x4a*>(up + 4 * i);
    }
    const int bb = tid >> 3, jj = tid & 7;
    const int eb = b0 + bb, ej = j0 + jj;
    const bool epi = tid < 128 && eb < B;
    const bool feeds = b0 + r16 < B;                                         // the sample whose h this lane feeds to the MFMA exists
    // granules of the group: [parity][16 samples][256 units]
    unsigned long long* const xg = a.xbuf + (size_t)g * 16 * H;
    const size_t xpar = (size_t)a.groups * 16 * H;
    const unsigned long long* const xin = xg + (size_t)r16 * H + wave * (H / 4) + q * KS;
    unsigned long long* const xout = xg + (size_t)bb * H + ej;
    float c = 0.f;
    bool gave_up = false;
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int t = rev ? T - 1 - s : s;
        float xv[4] = {0.f, 0.f, 0.f, 0.f};
        if (epi) {
            const float* xp = a.xproj + (((long long)eb * T + t) * ndir + d) * G4;
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) xv[gt] = xp[gt * H + ej];
        }
        float (*rd)[2][256] = red[s & 1];
        if (s > 0) {
            // h of the previous step: 16 granules of this lane's sample, tag = s
            const unsigned long long* xp = xin + ((s - 1) & 1) * xpar;
            unsigned long long gr[KS];
            int rounds = 0;
            gave_up = gave_up || give_up != 0;
            for (;;) {
#pragma unroll
                for (int i = 0; i < KS; ++i) gr[i] = __hip_atomic_load(xp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool ok = true;
#pragma unroll
                for (int i = 0; i < KS; ++i) ok = ok && (unsigned)(gr[i] >> 32) == (unsigned)s;
                if (__all(ok || !feeds) || gave_up) break;
                __builtin_amdgcn_s_sleep(1);
                if (++rounds > (1 << 21)) {
                    gave_up = true; give_up = 1;
                    if (lane == 0) raise_status(a.status, STATUS_SLOT_LSTM, STATUS_LSTM_HANDOFF);
                }
            }
            f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                float av = feeds ? __builtin_bit_cast(float, (unsigned)gr[i]) : 0.f;
                if (gave_up) av = __builtin_nanf("");
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, uv[i / 2][(i % 2) * 2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, uv[i / 2][(i % 2) * 2 + 1], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                rd[wave][0][(q * 4 + r) * 16 + r16] = acc0[r];
                rd[wave][1][(q * 4 + r) * 16 + r16] = acc1[r];
            }
            __syncthreads();
        }
        if (epi) {
            float a4[4];
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) {
                float v = xv[gt];
                if (s > 0) {
                    const int idx = bb * 16 + (gt & 1) * 8 + jj;
                    v += rd[0][gt >> 1][idx] + rd[1][gt >> 1][idx] + rd[2][gt >> 1][idx] + rd[3][gt >> 1][idx];
                }
                a4[gt] = v;
            }
            const float gi = sigmoidf_(a4[0]), gf = sigmoidf_(a4[1]), gc = tanhf(a4[2]), go = sigmoidf_(a4[3]);
            const long long so = ((long long)eb * T + t) * HH + (long long)d * H + ej;
            c = gf * c + gi * gc;
            const float h = go * tanhf(c);
            // the hand-off first: {h, tag s + 1} in one 8-byte sc1 store
            __hip_atomic_store(xout + (s & 1) * xpar, ((unsigned long long)(unsigned)(s + 1) << 32) | __builtin_bit_cast(unsigned, h),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.c_out[so] = c;
            a.h_out[so] = h;
            float* gp = a.gates + (((long long)eb * T + t) * ndir + d) * G4;
            gp[ej] = gi; gp[H + ej] = gf; gp[2 * H + ej] = gc; gp[3 * H + ej] = go;
        }
    }
}

}  // namespace ptts

using namespace ptts;

// the persistent forward kernel: H = 256, at least two steps, all workgroups resident at once (<= 4 per CU).  OFF by
// default (PTTS_LSTM_PERSISTENT=1 selects it, read at every call): at [64,400] it measured 10.6 us per step with the
// data-tagged granules below and 7.0 us with a drained-store + counter hand-off, against 6.3 us for the per-step launches --
// every sc1 round trip of the hand-off costs about what a kernel boundary does.
static bool lstm_persistent_ok(int B, int T, int H, int ndir) {
    const char* e = getenv("PTTS_LSTM_PERSISTENT");
    const int groups = ((B + 15) / 16) * ndir;
    if (!(e && atoi(e) != 0 && H == 256 && T >= 2 && groups * 32 <= 1024)) return false;
    // every workgroup of the grid must be resident at once (the members of a group wait for each other): refuse the path --
    // the per-step launches run instead -- unless the occupancy of an EMPTY chip covers the grid.  (Kernels of other streams
    // can still hold CUs; the kernel's poll then gives up after ~2^21 rounds and the step shows NaNs: opt-in, experimental.)
    static int per_cu = -1, ncu = 0;
    if (per_cu < 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&lstm_fwd_persistent_kernel), 256, 0) != hipSuccess) {
            (void)hipGetLastError();
            per_cu = 0;
        } else {
            ncu = prop.multiProcessorCount;
        }
    }
    return (long long)per_cu * ncu >= (long long)groups * 32;
}

// ---- the T launches of a recurrence replayed as ONE hipGraph launch ---------------------------------------------------------------
// A generator step spends 2 x 400 launches on the BLSTM; enqueued one by one they hold the host for about a millisecond each way
// (the wide kernels of the other branches wait behind them in the host's launch order), so the chain of a (pointers, shape) tuple is
// captured once, instantiated and replayed: the training loop's allocation pattern repeats, so the same addresses come back step after
// step.  A miss captures anew (least recently used entry dropped); when misses keep coming (addresses that do not repeat) the
// launches go out directly again and instantiation is retried only now and then.  Inside somebody else's capture (the optimiser's
// whole-step graph) the launches simply join that graph.  Off by default (PTTS_LSTM_GRAPH=1 / ptts_set_lstm_graph(1) switch it on):
// measured -0.1 ... -0.4 ms per generator step when the addresses repeat, but a capture costs more than it saves when they do not.
struct LstmGraphKey {
    int kind, B, T, H, ndir, reverse;
    const void* p[6];
    int device = -1;        // filled in by lstm_graph_run: an executable graph belongs to the device it was instantiated on
    bool operator==(const LstmGraphKey& o) const {
        if (kind != o.kind || B != o.B || T != o.T || H != o.H || ndir != o.ndir || reverse != o.reverse || device != o.device) return false;
        for (int i = 0; i < 6; ++i) if (p[i] != o.p[i]) return false;
        return true;
    }
};
struct LstmGraphEntry { LstmGraphKey k; hipGraphExec_t exec; unsigned long long stamp; };
static std::mutex g_lg_mu;
static std::vector<LstmGraphEntry> g_lg;
static unsigned long long g_lg_clock = 0, g_lg_hits = 0, g_lg_misses = 0, g_lg_direct = 0;
static int g_lg_consecutive_misses = 0;
constexpr size_t LSTM_GRAPH_CACHE = 16;

static int g_lstm_graph = -1;      // -1: from the environment (PTTS_LSTM_GRAPH, default off); 0 / 1: ptts_set_lstm_graph
static bool lstm_graph_on() {
    if (g_lstm_graph < 0) { const char* e = getenv("PTTS_LSTM_GRAPH"); g_lstm_graph = e ? (atoi(e) != 0) : 0; }
    return g_lstm_graph != 0;
}
extern "C" int ptts_set_lstm_graph(int on) { g_lstm_graph = on ? 1 : 0; return PTTS_OK; }

template <class F>
static int lstm_graph_run(const LstmGraphKey& key_in, hipStream_t st, const char* what, F&& launch_all) {
    LstmGraphKey key = key_in;
    if (hipGetDevice(&key.device) != hipSuccess) { (void)hipGetLastError(); key.device = -1; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (!lstm_graph_on() || key.T < 8 || hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        launch_all();
        return check_launch(what);
    }
    std::lock_guard<std::mutex> lock(g_lg_mu);
    ++g_lg_clock;
    for (auto& e : g_lg)
        if (e.k == key) {
            e.stamp = g_lg_clock; ++g_lg_hits; g_lg_consecutive_misses = 0;
            if (hipGraphLaunch(e.exec, st) != hipSuccess) { set_error("%s: hipGraphLaunch failed", what); return PTTS_ELAUNCH; }
            return PTTS_OK;
        }
    ++g_lg_misses;
    if (g_lg_consecutive_misses >= 8 && (g_lg_misses & 63) != 0) {      // addresses do not repeat: plain launches
        ++g_lg_direct;
        launch_all();
        return check_launch(what);
    }
    ++g_lg_consecutive_misses;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); launch_all(); return check_launch(what); }
    launch_all();
    // The launches above went into the capture, not to the device: if the capture cannot be ended or instantiated the recurrence has
    // NOT run yet -- clear the error, launch it directly, and stay on direct launches from now on (retried only now and then).
    auto direct_after_failure = [&]() {
        (void)hipGetLastError();
        g_lg_consecutive_misses = 1 << 20;
        ++g_lg_direct;
        launch_all();
        return check_launch(what);
    };
    if (hipStreamEndCapture(st, &graph) != hipSuccess || !graph) return direct_after_failure();
    const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess || !exec) return direct_after_failure();
    if (g_lg.size() >= LSTM_GRAPH_CACHE) {
        size_t lru = 0;
        for (size_t i = 1; i < g_lg.size(); ++i) if (g_lg[i].stamp < g_lg[lru].stamp) lru = i;
        (void)hipGraphExecDestroy(g_lg[lru].exec);
        g_lg.erase(g_lg.begin() + lru);
    }
    g_lg.push_back({key, exec, g_lg_clock});
    if (hipGraphLaunch(exec, st) != hipSuccess) { set_error("%s: hipGraphLaunch failed", what); return PTTS_ELAUNCH; }
    return PTTS_OK;
}

// hits / captures / direct-launch fallbacks of the recurrence graphs so far (bench.py reports them)
extern "C" int ptts_lstm_graph_stats(unsigned long long* hits, unsigned long long* captures, unsigned long long* direct) {
    std::lock_guard<std::mutex> lock(g_lg_mu);
    if (hits) *hits = g_lg_hits;
    if (captures) *captures = g_lg_misses - g_lg_direct;
    if (direct) *direct = g_lg_direct;
    return PTTS_OK;
}

// drop every cached recurrence graph (buffers about to be freed for good, tests)
extern "C" int ptts_lstm_graph_clear(void) {
    std::lock_guard<std::mutex> lock(g_lg_mu);
    for (auto& e : g_lg) (void)hipGraphExecDestroy(e.exec);
    g_lg.clear();
    g_lg_consecutive_misses = 0;
    return PTTS_OK;
}

extern "C" size_t ptts_lstm_fwd_workspace_bytes(int B, int T, int H, int ndir) {
    (void)T;
    const size_t granules = (size_t)2 * ((B + 15) / 16) * ndir * 16 * H * 8;     // the persistent kernel's {h, tag} buffers, behind the packed U
    return lstm_pk_ok(H) ? (size_t)ndir * 4 * H * H * sizeof(float) + granules : 16;
}

extern "C" int ptts_lstm_fwd(const float* xproj, const float* U, float* h_out, float* gates, float* c_out,
                             void* workspace, size_t workspace_bytes, int B, int T, int H, int ndir, int reverse,
                             void* stream) {
    PTTS_REQUIRE(xproj && U && h_out && gates && c_out, "lstm_fwd: null tensor");
    PTTS_REQUIRE(B > 0 && T > 0 && H > 0 && (ndir == 1 || ndir == 2), "lstm_fwd: bad dims");
    PTTS_REQUIRE((size_t)LY * H * sizeof(float) <= 64 * 1024, "lstm_fwd: H=%d too large", H);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((H + LX - 1) / LX, (B + LY - 1) / LY, ndir), block(LX, LY);
    const size_t lds = (size_t)LY * H * sizeof(float);
    if (lstm_pk_ok(H) && workspace && workspace_bytes >= ptts_lstm_fwd_workspace_bytes(B, T, H, ndir)) {
        float* Upk = (float*)workspace;
        if (lstm_persistent_ok(B, T, H, ndir)) {
            if (int rc0 = check_status("lstm_fwd")) return rc0;       // an earlier persistent launch gave up: sticky
            hipLaunchKernelGGL(lstm_pack_u_fwd_kernel, dim3(1024), dim3(256), 0, st, U, Upk, H, ndir);
            LstmPersistArgs pa;
            pa.xproj = xproj; pa.Upk = Upk; pa.h_out = h_out; pa.gates = gates; pa.c_out = c_out;
            pa.xbuf = reinterpret_cast<unsigned long long*>(Upk + (size_t)ndir * 4 * H * H);
            pa.B = B; pa.T = T; pa.ndir = ndir; pa.reverse = reverse; pa.groups = ((B + 15) / 16) * ndir;
            pa.status = status_words();
            int rc = zero_f32(reinterpret_cast<float*>(pa.xbuf), (size_t)2 * pa.groups * 16 * H * 2, st);      // tags 0
            if (rc) return rc;
            hipLaunchKernelGGL(lstm_fwd_persistent_kernel, dim3(pa.groups * 32), dim3(256), 0, st, pa);
            return check_launch("lstm_fwd_persistent");
        }
        dim3 mgrid(H / 8, (B + 15) / 16, ndir);
        const int KS = H / 16;
        const LstmGraphKey key{0, B, T, H, ndir, reverse, {xproj, U, h_out, gates, c_out, workspace}};
        return lstm_graph_run(key, st, "lstm_fwd_pk", [&]() {
            hipLaunchKernelGGL(lstm_pack_u_fwd_kernel, dim3(1024), dim3(256), 0, st, U, Upk, H, ndir);
            for (int s = 0; s < T; ++s) {
                if (KS == 4) hipLaunchKernelGGL(lstm_fwd_step_pk_kernel<4>, mgrid, dim3(256), 0, st, xproj, (const float*)Upk, h_out, gates, c_out, B, T, H, ndir, reverse, s);
                else if (KS == 8) hipLaunchKernelGGL(lstm_fwd_step_pk_kernel<8>, mgrid, dim3(256), 0, st, xproj, (const float*)Upk, h_out, gates, c_out, B, T, H, ndir, reverse, s);
                else hipLaunchKernelGGL(lstm_fwd_step_pk_kernel<16>, mgrid, dim3(256), 0, st, xproj, (const float*)Upk, h_out, gates, c_out, B, T, H, ndir, reverse, s);
            }
        });
    }
    if (H % 16 == 0) {
        dim3 mgrid(H / 8, (B + 15) / 16, ndir);
        for (int s = 0; s < T; ++s)
            hipLaunchKernelGGL(lstm_fwd_step_mfma_kernel, mgrid, dim3(256), 0, st, xproj, U, h_out, gates, c_out, B, T,
                               H, ndir, reverse, s);
        return check_launch("lstm_fwd_mfma");
    }
    for (int s = 0; s < T; ++s) {
        hipLaunchKernelGGL(lstm_fwd_step_kernel, grid, block, lds, st, xproj, U, h_out, gates, c_out, B, T, H, ndir,
                           reverse, s);
    }
    return check_launch("lstm_fwd");
}

static int lstm_bwd_waves() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PTTS_LSTM_BWD_WAVES"); v = e ? atoi(e) : 8; }
    return v;
}

extern "C" size_t ptts_lstm_bwd_workspace_bytes(int B, int T, int H, int ndir) {
    (void)T;
    return ((size_t)2 * ndir * 4 * H * H + (size_t)ndir * B * H) * sizeof(float);
}

extern "C" int ptts_lstm_bwd(const float* dh_out, const float* U, const float* gates, const float* c_out,
                             float* dgates, void* workspace, size_t workspace_bytes, int B, int T, int H, int ndir,
                             int reverse, void* stream) {
    PTTS_REQUIRE(dh_out && U && gates && c_out && dgates, "lstm_bwd: null tensor");
    PTTS_REQUIRE(B > 0 && T > 0 && H > 0 && (ndir == 1 || ndir == 2), "lstm_bwd: bad dims");
    PTTS_REQUIRE((size_t)LY * 4 * H * sizeof(float) <= 64 * 1024, "lstm_bwd: H=%d too large", H);
    const size_t need = ptts_lstm_bwd_workspace_bytes(B, T, H, ndir);
    if (!workspace || workspace_bytes < need) {
        set_error("lstm_bwd: workspace %zu < %zu", workspace_bytes, need);
        return PTTS_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* UT = (float*)workspace;
    float* UTpk = UT + (size_t)ndir * 4 * H * H;
    float* dc_state = UTpk + (size_t)ndir * 4 * H * H;
    if (lstm_pk_ok(H)) {
        dim3 pgrid(H / 16, (B + 15) / 16, ndir);
        const bool w8 = H == 256 && lstm_bwd_waves() == 8;
        const LstmGraphKey key{1, B, T, H, ndir, reverse, {dh_out, U, gates, c_out, dgates, workspace}};
        return lstm_graph_run(key, st, "lstm_bwd_pk", [&]() {
            hipLaunchKernelGGL(lstm_pack_u_bwd_kernel, dim3(1024), dim3(256), 0, st, U, UTpk, H, ndir);
            for (int s = T - 1; s >= 0; --s)
                if (w8)
                    hipLaunchKernelGGL(lstm_bwd_step_pk_kernel<8>, pgrid, dim3(512), 0, st, dh_out,
                                       (const float*)UTpk, gates, c_out, dgates, dc_state, B, T, H, ndir, reverse, s);
                else
                    hipLaunchKernelGGL(lstm_bwd_step_pk_kernel<4>, pgrid, dim3(256), 0, st, dh_out,
                                       (const float*)UTpk, gates, c_out, dgates, dc_state, B, T, H, ndir, reverse, s);
        });
    }
    const long long tot = (long long)ndir * 4 * H * H;
    int tb = (int)((tot + 255) / 256);
    if (tb > 2048) tb = 2048;
    hipLaunchKernelGGL(lstm_transpose_kernel, dim3(tb), dim3(256), 0, st, U, UT, H, ndir);
    dim3 grid((H + LX - 1) / LX, (B + LY - 1) / LY, ndir), block(LX, LY);
    const size_t lds = (size_t)LY * 4 * H * sizeof(float);
    if (H % 16 == 0) {
        dim3 mgrid(H / 16, (B + 15) / 16, ndir);
        for (int s = T - 1; s >= 0; --s)
            hipLaunchKernelGGL(lstm_bwd_step_mfma_kernel, mgrid, dim3(256), 0, st, dh_out, (const float*)UT, gates,
                               c_out, dgates, dc_state, B, T, H, ndir, reverse, s);
        return check_launch("lstm_bwd_mfma");
    }
    for (int s = T - 1; s >= 0; --s) {
        hipLaunchKernelGGL(lstm_bwd_step_kernel, grid, block, lds, st, dh_out, (const float*)UT, gates, c_out,
                           dgates, dc_state, B, T, H, ndir, reverse, s);
    }
    return check_launch("lstm_bwd");
}

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

}  // namespace ptts

using namespace ptts;

extern "C" int ptts_lstm_fwd(const float* xproj, const float* U, float* h_out, float* gates, float* c_out, int B,
                             int T, int H, int ndir, int reverse, void* stream) {
    PTTS_REQUIRE(xproj && U && h_out && gates && c_out, "lstm_fwd: null tensor");
    PTTS_REQUIRE(B > 0 && T > 0 && H > 0 && (ndir == 1 || ndir == 2), "lstm_fwd: bad dims");
    PTTS_REQUIRE((size_t)LY * H * sizeof(float) <= 64 * 1024, "lstm_fwd: H=%d too large", H);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((H + LX - 1) / LX, (B + LY - 1) / LY, ndir), block(LX, LY);
    const size_t lds = (size_t)LY * H * sizeof(float);
    for (int s = 0; s < T; ++s) {
        hipLaunchKernelGGL(lstm_fwd_step_kernel, grid, block, lds, st, xproj, U, h_out, gates, c_out, B, T, H, ndir,
                           reverse, s);
    }
    return check_launch("lstm_fwd");
}

extern "C" size_t ptts_lstm_bwd_workspace_bytes(int B, int T, int H, int ndir) {
    (void)T;
    return ((size_t)ndir * 4 * H * H + (size_t)ndir * B * H) * sizeof(float);
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
    float* dc_state = UT + (size_t)ndir * 4 * H * H;
    const long long tot = (long long)ndir * 4 * H * H;
    int tb = (int)((tot + 255) / 256);
    if (tb > 2048) tb = 2048;
    hipLaunchKernelGGL(lstm_transpose_kernel, dim3(tb), dim3(256), 0, st, U, UT, H, ndir);
    dim3 grid((H + LX - 1) / LX, (B + LY - 1) / LY, ndir), block(LX, LY);
    const size_t lds = (size_t)LY * 4 * H * sizeof(float);
    for (int s = T - 1; s >= 0; --s) {
        hipLaunchKernelGGL(lstm_bwd_step_kernel, grid, block, lds, st, dh_out, (const float*)UT, gates, c_out,
                           dgates, dc_state, B, T, H, ndir, reverse, s);
    }
    return check_launch("lstm_bwd");
}

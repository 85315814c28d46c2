// Weight gradient of the context Conv1D on the fp32 matrix cores over FRAME-MAJOR operands.
//   dW[j][c][n] = sum_{b,t} Xp[b][t + j][c] dY[b][t][n]  =  sum_q Xt[c][q + j] dYt[n][q],   q = b (T + KW - 1) + t
// (Xp: the frames zero-padded in time; reference networktts.py:116-120, TF's Conv1D kernel backprop).  The stream-K
// product of gemm.hip reads this as A^T . dY with A the implicit im2col matrix: its A^T fragments come out of LDS one
// float per lane and instruction (k is the strided index there), and every tap re-reads the frames it shares with the
// 20 others.  Here both operands are transposed once (Xt [Crows][Pp], dYt [N][Pp]: the reduction index q contiguous), a
// lane reads its frames of a step ONCE (six ds_read_b128) and the operand of tap j is simply the register j places
// further: exact fp32 arithmetic (v_mfma_f32_16x16x4_f32), 7 LDS reads per 84 MFMAs, the loop is MFMA-paced.
// A workgroup (8 waves) owns 64 channels x 32 outputs x ALL KW taps over a slice of q; wave (cf, nf): 16 x 16 x KW.
#include "common.h"
#include <cstdlib>

namespace ptts {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int T_THREADS = 512;
constexpr int T_CB = 64, T_NB = 32, T_QS = 32;        // channels, outputs per workgroup; frames per step
constexpr int T_AROW = 64;                            // frames of an Xt row in LDS (32 + taps <= 24, rounded to 256 B)
constexpr int T_ATILE = T_CB * T_AROW;                // floats: 16 KB
constexpr int T_BTILE = T_NB * T_QS;                  // floats: 4 KB
constexpr int T_STAGE = T_ATILE + T_BTILE;            // 20 KB
constexpr int T_STAGES = 3;
constexpr int T_LDS_BYTES = T_STAGES * T_STAGE * 4;   // 60 KB

struct WgradTArgs {
    const float* Xt; const float* Yt; float* dW; float* db;
    int C, N;
    long long Pp;
    int qsteps, nsplit, steps_per_split, tiles_n;
};

// fp32 [B][T][C] -> [Crows][Pp], element (c, b Tp + pad_left + t); everything else zero.  LDS-tiled transpose.
__global__ __launch_bounds__(256) void transpose_frames_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int T,
                                                               int C, int pad_left, int Tp, int Crows, long long Pp) {
    __shared__ float s[32][64 + 1];
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
            s[tx][ql] = v;
        }
    }
    __syncthreads();
    {
        const int c = tid >> 3, g8 = tid & 7;           // 8 frames = two float4 per thread
        if (c0 + c < Crows) {
            float* dst = out + (long long)(c0 + c) * Pp + q0 + 8 * g8;
            const float4 a = make_float4(s[c][8 * g8], s[c][8 * g8 + 1], s[c][8 * g8 + 2], s[c][8 * g8 + 3]);
            const float4 b4 = make_float4(s[c][8 * g8 + 4], s[c][8 * g8 + 5], s[c][8 * g8 + 6], s[c][8 * g8 + 7]);
            *reinterpret_cast<float4*>(dst) = a;
            *reinterpret_cast<float4*>(dst + 4) = b4;
        }
    }
}

typedef void __attribute__((address_space(3)))* t_lptr;
// see gemm.hip dma16: inline assembly so that the kernel, not the compiler, orders the DMAs
__device__ __forceinline__ void dma16f(const float* src, float* lds_wave_base) {
    const unsigned lds_off = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long)(t_lptr)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_off) : "memory", "m0");
}

template <int KW>
__global__ __launch_bounds__(T_THREADS) void wgrad_f32_t_kernel(WgradTArgs g) {
    extern __shared__ __attribute__((aligned(16))) float t_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cf = wave >> 1, nf = wave & 1;
    const int li = lane & 15, kg = lane >> 4;
    const int tile = blockIdx.x / g.nsplit, split = blockIdx.x - tile * g.nsplit;
    const int c0 = (tile / g.tiles_n) * T_CB, n0 = (tile % g.tiles_n) * T_NB;
    const int s_begin = split * g.steps_per_split;
    const int s_end = min(g.qsteps, s_begin + g.steps_per_split);
    const int nsteps = s_end - s_begin;
    if (nsteps <= 0) return;

    f32x4 acc[KW];
#pragma unroll
    for (int j = 0; j < KW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;                                // bias gradient: sum over q of dYt (first channel tile, cf == 0 only)
    const bool want_b = g.db != nullptr && c0 == 0 && cf == 0;

    // DMA sources.  Xt tile (16 KB, two wave-instructions per wave): instruction u = wave, wave + 8 fills rows 4u .. 4u+3
    // (256 B each): lane -> row lane>>4, physical 16-byte unit lane&15 holding logical unit (lane&15) ^ (row&15).
    // dYt tile (4 KB, waves 0..3): rows 8 wave .. +7 (128 B each): lane -> row lane>>3, unit (lane&7) ^ (row&7).
    long long asrc[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int r = 4 * (wave + 8 * x) + (lane >> 4);
        const int u = (lane & 15) ^ (r & 15);
        asrc[x] = (long long)(c0 + r) * g.Pp + 4 * u;
    }
    long long bsrc = 0;
    if (wave < 4) {
        const int r = 8 * wave + (lane >> 3);
        const int u = (lane & 7) ^ (r & 7);
        bsrc = (long long)(n0 + r) * g.Pp + 4 * u;
    }
    auto issue = [&](int s, int stage) {
        float* st = t_lds + stage * T_STAGE;
        const long long q0 = (long long)s * T_QS;
#pragma unroll
        for (int x = 0; x < 2; ++x) dma16f(g.Xt + asrc[x] + q0, st + 4 * (wave + 8 * x) * T_AROW);
        if (wave < 4) dma16f(g.Yt + bsrc + q0, st + T_ATILE + 8 * wave * T_QS);
    };
    const int arow = 16 * cf + li;
    const int abase = arow * T_AROW, asw = arow & 15;
    const int brow = 16 * nf + li;
    const int bbase = T_ATILE + brow * T_QS, bsw = brow & 7;

    issue(s_begin, 0);
    if (nsteps > 1) issue(s_begin + 1, 1);
    for (int s = 0; s < nsteps; ++s) {
        // DMAs of this wave issued after those of step s: the ones of step s+1 (3 for waves 0..3, 2 for the others)
        if (s + 1 < nsteps) { if (wave < 4) __builtin_amdgcn_s_waitcnt(0xF73); else __builtin_amdgcn_s_waitcnt(0xF72); }
        else __builtin_amdgcn_s_waitcnt(0xF70);
        __syncthreads();
        if (s + 2 < nsteps) issue(s_begin + s + 2, (s + 2) % T_STAGES);
        const float* st = t_lds + (s % T_STAGES) * T_STAGE;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // frames 16 h + 4 kg .. + 23 of this lane's channel row (six units), frames 16 h + 4 kg .. + 3 of its output row
            float xw[24];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const float4 v = *reinterpret_cast<const float4*>(st + abase + 4 * ((4 * h + kg + u) ^ asw));
                xw[4 * u] = v.x; xw[4 * u + 1] = v.y; xw[4 * u + 2] = v.z; xw[4 * u + 3] = v.w;
            }
            const float4 yv4 = *reinterpret_cast<const float4*>(st + bbase + 4 * ((4 * h + kg) ^ bsw));
            const float yv[4] = {yv4.x, yv4.y, yv4.z, yv4.w};
            if (want_b) bsum += (yv4.x + yv4.y) + (yv4.z + yv4.w);
            // MFMA e of a tap multiplies the k-set {e, 4 + e, 8 + e, 12 + e} of this half (lane group kg supplies 4 kg + e)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < KW; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xw[j + e], yv[e], acc[j], 0, 0, 0);
        }
    }
    // C/D layout: col = lane&15 -> n, row = 4 (lane>>4) + reg -> c
    const int n = n0 + 16 * nf + li;
#pragma unroll
    for (int j = 0; j < KW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = c0 + 16 * cf + 4 * kg + r;
            if (c < g.C && n < g.N) atomicAdd(g.dW + ((long long)j * g.C + c) * g.N + n, acc[j][r]);
        }
    if (want_b) {
        // the 4 lane groups hold disjoint frame subsets of output row n
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (kg == 0 && n < g.N) atomicAdd(g.db + n, bsum);
    }
}

}  // namespace ptts

using namespace ptts;

extern "C" int ptts_transpose_frames(const float* x, float* out, int B, int T, int C, int pad_left, int Tp, int Crows,
                                     long long Pp, void* stream) {
    PTTS_REQUIRE(x && out, "transpose_frames: null pointer");
    PTTS_REQUIRE(B > 0 && T > 0 && C > 0 && pad_left >= 0 && Tp >= T + pad_left, "transpose_frames: bad dims");
    PTTS_REQUIRE(Crows >= C && Crows % 32 == 0 && Pp % 64 == 0 && Pp >= (long long)B * Tp,
                 "transpose_frames: Crows=%d must be a multiple of 32 >= C, Pp=%lld a multiple of 64 >= B*Tp", Crows, Pp);
    dim3 grid((unsigned)(Pp / 64), (unsigned)(Crows / 32));
    hipLaunchKernelGGL(transpose_frames_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, out, B, T, C, pad_left, Tp, Crows, Pp);
    return check_launch("transpose_frames");
}

extern "C" int ptts_conv1d_wgrad_t(const float* xt, const float* yt, float* dw, float* db, int B, int T, int KW, int C, int N,
                                   int Crows, long long Pp, void* stream) {
    PTTS_REQUIRE(xt && yt && dw, "conv1d_wgrad_t: null pointer");
    PTTS_REQUIRE(B > 0 && T > 0 && C > 0 && N > 0, "conv1d_wgrad_t: bad dims");
    PTTS_REQUIRE(KW == 21 || KW == 5 || KW == 3, "conv1d_wgrad_t: KW=%d is not instantiated (3, 5, 21)", KW);
    PTTS_REQUIRE(N % T_NB == 0 && Crows % T_CB == 0 && Crows >= C, "conv1d_wgrad_t: N %% 32 == 0, Crows %% 64 == 0 >= C");
    const long long frames = (long long)B * (T + KW - 1);
    const int qsteps = (int)((frames + T_QS - 1) / T_QS);
    PTTS_REQUIRE(Pp % 64 == 0 && Pp >= (long long)qsteps * T_QS + T_AROW, "conv1d_wgrad_t: Pp=%lld too short (needs %lld)", Pp,
                 (long long)qsteps * T_QS + T_AROW);
    PTTS_REQUIRE((((size_t)xt | (size_t)yt) & 15) == 0, "conv1d_wgrad_t: operands must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    WgradTArgs g;
    g.Xt = xt; g.Yt = yt; g.dW = dw; g.db = db; g.C = C; g.N = N; g.Pp = Pp; g.qsteps = qsteps;
    g.tiles_n = N / T_NB;
    const int tiles = (Crows / T_CB) * g.tiles_n;
    int nsplit = 256 / tiles;                              // about one workgroup per CU
    if (nsplit < 1) nsplit = 1;
    if (nsplit > qsteps) nsplit = qsteps;
    g.nsplit = nsplit; g.steps_per_split = (qsteps + nsplit - 1) / nsplit;
    if (zero_f32(dw, (size_t)KW * C * N, st) != PTTS_OK) return PTTS_ELAUNCH;
    if (db && zero_f32(db, (size_t)N, st) != PTTS_OK) return PTTS_ELAUNCH;
#define PTTS_WT_LAUNCH(KWV)                                                                                               \
    {                                                                                                                     \
        static bool attr = false;                                                                                         \
        if (!attr) {                                                                                                      \
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_f32_t_kernel<KWV>),                    \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);                 \
            if (e2 != hipSuccess) { set_error("conv1d_wgrad_t: LDS attribute: %s", hipGetErrorString(e2)); return PTTS_ELAUNCH; } \
            attr = true;                                                                                                  \
        }                                                                                                                 \
        hipLaunchKernelGGL((wgrad_f32_t_kernel<KWV>), dim3(tiles * nsplit), dim3(T_THREADS), T_LDS_BYTES, st, g);         \
    }
    if (KW == 21) PTTS_WT_LAUNCH(21) else if (KW == 5) PTTS_WT_LAUNCH(5) else PTTS_WT_LAUNCH(3)
#undef PTTS_WT_LAUNCH
    return check_launch("conv1d_wgrad_t");
}

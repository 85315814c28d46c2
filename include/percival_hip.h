/*
 * percival_hip.h -- C ABI of libpercival_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary "B2" of SURVEY.md section 8(b): the arithmetic that the
 * reference delegates to TensorFlow/Keras on its WGAN-GP training hot path
 * (reference: percivaltts/optimizertts_wgan.py:107-241, networks_critic.py:44-96,
 * modeltts_common.py:65-126, networktts.py:59-134) is provided here as hand-written HIP
 * kernels.  The reference has no FFI of its own (it is pure Python on tf.keras), so each
 * entry point cites the Keras call site it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no framework types.
 *   - every pointer is a DEVICE pointer unless the name ends in _host.
 *   - every function is asynchronous on `stream` (a hipStream_t passed as void*),
 *     never allocates or frees, never synchronises, and is safe to capture in a hipGraph.
 *   - return value: 0 = ok, negative = PTTS_E* (see ptts_last_error()).
 *   - tensors are fp32, channel-last, C-contiguous:
 *       frames      [B, T, D]
 *       conv2d maps [B, T, F, C]           (Keras channels_last)
 *       conv2d kernels [KT, KF, Cin, Cout] (Keras HWIO)
 *       dense / conv1d kernels [K, N] resp. [KW, Cin, N]  (Keras layout)
 *   - "input transform": the layers of the reference are Linear -> (BatchNorm) -> LeakyReLU
 *     (networktts.py:59-63,116-126).  We keep the PRE-activation tensor z in HBM and apply
 *     a = act(scale*z + shift) while the next linear op loads it, so an activation is never
 *     written and re-read.  PTTS_IN_* selects that transform.
 */
#ifndef PERCIVAL_HIP_H
#define PERCIVAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTTS_OK            0
#define PTTS_EINVAL       -1   /* bad argument / unsupported shape */
#define PTTS_ELAUNCH      -2   /* hipLaunch failed */
#define PTTS_EWORKSPACE   -3   /* workspace too small */
#define PTTS_EDEVICE      -4   /* a kernel reported a failed hand-off (sticky device status word, see ptts_device_status) */

/* input transforms (applied on load; x is what lies in HBM) */
#define PTTS_IN_NONE     0   /* a = x                                   */
#define PTTS_IN_LRELU    1   /* p = x*scale[c]+shift[c]; a = p>0?p:alpha*p  (scale/shift may be NULL) */
#define PTTS_IN_MASKMUL  2   /* a = x * (mask_src>0 ? 1 : alpha)        (second-order sweep of the gradient penalty) */

/* conv2d padding modes along time (frequency is always 'same') */
#define PTTS_PAD_SAME    0
#define PTTS_PAD_CAUSAL  1

/* output activations of ptts_affine_act */
#define PTTS_ACT_NONE    0
#define PTTS_ACT_LRELU   1
#define PTTS_ACT_SIGMOID 2
#define PTTS_ACT_TANH    3

const char* ptts_version(void);
const char* ptts_device_arch(void);     /* "gfx950" : the only code object in the library */
const char* ptts_last_error(void);      /* thread-local message of the last failure */
/* Device status.  Kernels whose waves wait for each other with bounded polls (the wave-specialised Conv2D forward's LDS-counter
 * hand-off, the persistent LSTM's granule hand-off) do not hang when a count never arrives -- and do not stay silent either: the
 * wave that gave up stores a code into a block of pinned, device-mapped host memory.  The word is STICKY: ptts_device_status()
 * returns PTTS_EDEVICE (message in ptts_last_error(), bit mask in *word_out: 1 = Conv2D hand-off, 2 = LSTM hand-off) from then on,
 * and so does every later ptts_conv2d_mfma_fwd / ptts_lstm_fwd call, until ptts_device_status_clear().  Reading it is a host
 * memory load (no synchronisation): call it at step boundaries.  The reference has no counterpart (TF raises from Session.run,
 * optimizertts.py:254 turns a NaN cost into a ValueError); this is how a corrupted launch becomes an error instead of a bad step.
 * ptts_device_status_word() returns the HOST address of the first word (tests poke it to exercise the host logic without a GPU);
 * ptts_device_status_message() decodes a mask. */
int ptts_device_status(unsigned* word_out);
int ptts_device_status_clear(void);
unsigned* ptts_device_status_word(void);
int ptts_device_status_message(unsigned word, char* buf, size_t n);
/* Deterministic mode: reductions over workgroups in a fixed order only (no fp32 atomics: stream-K GEMM tiles, the
 * thin weight-gradient kernel and the loss scalars take their single-pass forms).  Returns the previous setting. */
int ptts_set_deterministic(int on);
int ptts_get_deterministic(void);
/* bf16 products (BASELINE configs[2]; the reference is fp32, README.md:166): the split GEMM kernels -- ptts_conv1d_bf16x6,
 * ptts_conv1d_wgrad_bf16x6, ptts_dense_bf16x6, ptts_dense_wgrad_bf16x6* -- form ONE product of the operands' bf16 roundings
 * (plane 1 of the split) instead of the six products of the fp32 split; fp32 accumulation, fp32 master weights and
 * gradients.  Returns the previous setting. */
int ptts_set_bf16_products(int on);
int ptts_get_bf16_products(void);

/* ---------------------------------------------------------------------------------------
 * 2D convolution over (time x frequency), NHWC, stride 1.
 * Replaces kl.Conv2D at networks_critic.py:67, networktts.py:123,129-130, modeltts_common.py:100.
 *   y[b,t,f,:] = bias + sum_{kt,kf} a[b, t+kt*dil_t-pad_t, f+kf-pad_f, :] . w[kt,kf,:,:]
 *   a = transform(x) inside the image, 0 outside ('same' zero padding is in the activation domain).
 * ------------------------------------------------------------------------------------- */
int ptts_conv2d_fwd(const float* x, const float* w, const float* bias /*[Cout] or NULL*/,
                    const float* in_scale /*[Cin] or NULL*/, const float* in_shift /*[Cin] or NULL*/,
                    const float* mask_src /*[B,T,F,Cin], PTTS_IN_MASKMUL only*/,
                    float* y,
                    int B, int T, int F, int Cin, int Cout, int KT, int KF,
                    int dil_t, int pad_mode, int in_mode, float alpha, void* stream);

/* Fused backward of the layer above: given dy = dL/dy and the same (x, transform) as the forward,
 *   da = conv^T(dy, w);  dx = da * d(a)/d(x)          -> dx   (skipped when dx == NULL)
 *   dw = corr(a, dy), dbias = sum dy                  -> dw, dbias (skipped when dw == NULL)
 *   dscale[c] = sum da*lrelu'(p)*x ; dshift[c] = sum da*lrelu'(p)   (PTTS_IN_LRELU with scale; skipped when NULL)
 * One pass over HBM: reads dy, x (and mask_src), writes dx.  Weight-shaped results are reduced
 * deterministically through `workspace` (ptts_conv2d_bwd_workspace_bytes) -- no float atomics. */
size_t ptts_conv2d_bwd_workspace_bytes(int B, int T, int F, int Cin, int Cout, int KT, int KF, int dil_t);
int ptts_conv2d_bwd(const float* dy, const float* x, const float* w,
                    const float* in_scale, const float* in_shift, const float* mask_src,
                    float* dx, float* dw, float* dbias, float* dscale, float* dshift,
                    void* workspace, size_t workspace_bytes,
                    int B, int T, int F, int Cin, int Cout, int KT, int KF,
                    int dil_t, int pad_mode, int in_mode, float alpha, void* stream);

/* The same pass with the weight-shaped sums left unreduced: dx as above (NULL to skip), and the per-workgroup partial sums
 * of dw / dbias as rows [nblocks][npart] (npart = KT*KF*Cin*Cout + Cout + 2*Cin) behind a 4096-byte head of the caller's
 * workspace; *nblocks_out receives nblocks.  Only for shapes with a tiled kernel (workspace_bytes > 16) and without a
 * BatchNorm-fused input.  ptts_conv2d_reduce_grouped then adds up to any number of such passes into their gradient
 * buffers (dw[nw], dbias[cout]; fp32 atomics: several passes may share a buffer) in one launch per 16 passes -- the
 * 8 layers x 3 passes of a critic step otherwise pay one 7 us reduction launch each. */
int ptts_conv2d_bwd_partials(const float* dy, const float* x, const float* w, const float* mask_src,
                             float* dx, void* workspace, size_t workspace_bytes, int* nblocks_out,
                             int B, int T, int F, int Cin, int Cout, int KT, int KF,
                             int dil_t, int pad_mode, int in_mode, float alpha, void* stream);
typedef struct ptts_conv2d_reduce_desc {
    const float* partials;      /* workspace + 4096 bytes */
    int nblocks, npart, nw, cout;
    float* dw; float* dbias;    /* accumulated into; either may be NULL */
} ptts_conv2d_reduce_desc;
int ptts_conv2d_reduce_grouped(const ptts_conv2d_reduce_desc* descs, int n, void* stream);

/* ---------------------------------------------------------------------------------------
 * The same convolution for the 4 -> 4 channel, 5x5 layers (the critic's stack, networks_critic.py:66-68, and the
 * generator's, networktts.py:122-126) on the bf16 matrix cores, in fp32 arithmetic: both operands are split three ways
 * into bf16 (x = x1 + x2 + x3 exactly), the six products of order >= 2^-16 are formed by v_mfma_f32_16x16x32_bf16 and
 * accumulated in fp32 (csrc/conv2d_mfma.hip).  Time dilation 1, 2, 4 or 8; any T, F, B.
 *   ptts_conv2d_mfma_tables   the banded (Toeplitz) operand tables of a kernel w [5,5,4,4], 3 bf16 planes each
 *                             (ptts_conv2d_mfma_table_bytes(5) bytes): table_fwd for the forward, table_bwd (flipped,
 *                             transposed) for the backward-data pass; either may be NULL.  Rebuild when w changes.
 *   ptts_conv2d_mfma_fwd      y = bias + conv(transform(x)) with pad_t rows of zero padding before t = 0; when out_mask
 *                             is given, y *= (out_mask > 0 ? 1 : alpha).  Forward: table_fwd, pad_t = 2 dil ('same') or
 *                             4 dil (causal).  Masked forward of the second-order sweep: in_mode PTTS_IN_MASKMUL.
 *                             Backward data: x = dy, table_bwd, pad_t = 4 dil - pad_t(forward), out_mask = the forward
 *                             layer's pre-activation input (its LeakyReLU mask), in_mode PTTS_IN_NONE.
 *   ptts_conv2d_mfma_wgrad_partials   per-workgroup partial sums of dw / dbias in the row layout of
 *                             ptts_conv2d_bwd_partials (rows of *npart_out floats behind a 4096-byte head), reduced in a
 *                             fixed order inside a workgroup (no atomics); ptts_conv2d_reduce_grouped adds them up.
 *   ptts_conv2d_mfma_debug    measurement hook of tools/conv2d_mfma_probe.py (phase switches, per-workgroup stamps);
 *                             flags 0 = the product path; bit 16 alone keeps the results and launches the four-wave form of
 *                             the dilation-1 fp32 forward kernel instead of the wave-specialised default (A/B in tests).
 * ------------------------------------------------------------------------------------- */
int ptts_conv2d_mfma_supported(int F, int Cin, int Cout, int KT, int KF, int dil_t);
size_t ptts_conv2d_mfma_table_bytes(int KT);
int ptts_conv2d_mfma_tables(const float* w, void* table_fwd, void* table_bwd, int KT, int KF, int Cin, int Cout,
                            int planes /*3: fp32 split, 1: bf16 copy of the kernel*/, void* stream);
/* The same for n kernels (5x5, 4 -> 4) in one launch: w[i] -> table_fwd[i], table_bwd[i] (host arrays of device pointers). */
int ptts_conv2d_mfma_tables_grouped(const float* const* w, void* const* table_fwd, void* const* table_bwd, int n, int planes,
                                    void* stream);
/* planes = 3: fp32 arithmetic (six products), every tensor fp32.  planes = 1: bf16 arithmetic (BASELINE configs[2]; time
 * dilation 1): ONE product per position with the bf16 copy of the kernel, fp32 accumulation; x and mask_src are bf16 in
 * HBM when in_bf16 (else fp32, rounded to bf16 on load), y and out_mask are bf16 when out_bf16 (else fp32). */
int ptts_conv2d_mfma_fwd(const void* x, const void* table, const float* bias, const float* in_scale, const float* in_shift,
                         const void* mask_src, const void* out_mask, void* y,
                         int B, int T, int F, int KT, int dil_t, int pad_t, int in_mode, float alpha,
                         int planes, int in_bf16, int out_bf16, void* stream);
/* The same forward pass (fp32 maps, dilation 1, in_mode NONE / LRELU with or without the BatchNorm affine) that ALSO leaves the
 * per-workgroup sums of the outputs it stores -- stats[*nrows_out][8] doubles: four channel sums, four channel sums of squares -- for
 * the kl.BatchNormalization that follows the kl.Conv2D in pCNN2D (networktts.py:122-126): TF computes the layer's batch moments with a
 * pass of its own over the map (FusedBatchNorm); here they ride on the convolution's stores and ptts_bn_finalize_partials finishes
 * them.  capacity_rows >= 256.  y is bit-identical to ptts_conv2d_mfma_fwd's. */
int ptts_conv2d_mfma_fwd_stats_supported(int F, int dil_t, int in_mode);
int ptts_conv2d_mfma_fwd_stats(const float* x, const void* table, const float* bias, const float* in_scale, const float* in_shift,
                               float* y, int B, int T, int F, int KT, int pad_t, int in_mode, float alpha,
                               double* stats, int capacity_rows, int* nrows_out, void* stream);
size_t ptts_conv2d_mfma_wgrad_workspace_bytes(int B, int T);
/* x (and mask_src) bf16 when x_bf16, dy bf16 when dy_bf16; the partial sums and the reduced gradients are fp32 always. */
int ptts_conv2d_mfma_wgrad_partials(const void* dy, const void* x, const void* mask_src, void* workspace,
                                    size_t workspace_bytes, int* nblocks_out, int* npart_out, int B, int T, int F,
                                    int KT, int dil_t, int pad_t, int in_mode, float alpha,
                                    int planes, int x_bf16, int dy_bf16, void* stream);
/* Fused backward launches (round 4; dilation 1, 'same' padding, fp32 maps): the two passes of a layer that read the same staged tile
 * as ONE launch -- what TF runs as Conv2DBackpropInput + Conv2DBackpropFilter (+ BiasAddGrad) of one kl.Conv2D (networks_critic.py:67),
 * and, for the gradient penalty (optimizertts_wgan.py:53-68), the forward + Conv2DBackpropFilter pair of the backward of
 * Conv2DBackpropInput:
 *   kind 1  p = dy, q = x (the layer's pre-activation input), table = table_bwd:  y = dx = conv^T(dy) . lrelu'(x), and the partial
 *           rows of dw = corr(lrelu(x), dy), dbias = sum dy
 *   kind 2  p = u with mask_src = x, q = dy, table = table_fwd:  y = conv(u . lrelu'(x)), and the partial rows of
 *           dw = corr(u . lrelu'(x), dy) (bias sums zero)
 * pad_t is the forward layer's (2).  y is bit-identical to ptts_conv2d_mfma_fwd's; the partial rows (one per workgroup, behind a
 * 4096-byte head, *npart_out floats each) go to ptts_conv2d_reduce_grouped like ptts_conv2d_mfma_wgrad_partials'. */
size_t ptts_conv2d_mfma_bwd_fused_workspace_bytes(int B, int T);
int ptts_conv2d_mfma_bwd_fused_supported(int F, int dil_t, int planes);
int ptts_conv2d_mfma_bwd_fused(const void* p, const void* q, const void* mask_src, const void* table, void* y,
                               void* workspace, size_t workspace_bytes, int* nblocks_out, int* npart_out,
                               int B, int T, int F, int KT, int pad_t, int kind, float alpha, void* stream);
/* Kind 1 for a layer whose input was lrelu(scale x + shift) -- the kl.BatchNormalization + kl.LeakyReLU in front of the generator's
 * kl.Conv2D layers (networktts.py:122-126): q = the raw map x, y = dx = conv^T(dy) lrelu'(scale x + shift) scale (TF: Conv2DBackpropInput +
 * LeakyReluGrad + the input half of FusedBatchNormGrad's elementwise stage), and every partial row carries, behind the 400 + 4 sums of
 * dw / dbias, the 4 + 4 sums that are the gradients of scale and shift (columns 404 .. 411: a second ptts_conv2d_reduce_desc with
 * partials + 404 floats, nw = 4, cout = 4 reduces them). */
int ptts_conv2d_mfma_bwd_fused_affine(const float* p, const float* q, const void* table, float* y, void* workspace, size_t workspace_bytes,
                                      int* nblocks_out, int* npart_out, int B, int T, int F, int KT, int pad_t, float alpha,
                                      const float* q_scale, const float* q_shift, void* stream);
int ptts_conv2d_mfma_debug(int flags, void* stamp_buf);

/* ---------------------------------------------------------------------------------------
 * The critic's WHOLE Conv2D stack per launch (csrc/conv2d_chain.hip; BASELINE configs[2]: bf16 storage, bf16 products,
 * fp32 accumulation, fp32 master weights and weight gradients).  Replaces the L x (kl.Conv2D 5x5 + kl.LeakyReLU) of
 * networks_critic.py:64-70 -- forward, TF's Conv2DBackpropInput / Conv2DBackpropFilter chains behind it, and the
 * second-order sweep of K.gradients(K.gradients(...)) (optimizertts_wgan.py:53-68) -- with the maps between the layers
 * held in the LDS: a workgroup carries a tile of 32 time rows x all F <= 68 bins through all L <= 8 layers (two halo rows
 * per layer and side, recomputed).  Channels: cin0 (1) -> 4 -> ... -> 4.
 *   maps     a_1 .. a_{L-1} = lrelu(z_l), POST-activation, [L-1][B][T][FP][4] bf16, FP = F rounded up to even, pad bin zero
 *            (ptts_conv2d_chain_map_elems(B, T, F) elements each);   a_last = a_L [B][T][F][4] bf16 (what the dense layers read)
 *   tables   ptts_conv2d_chain_tables_bytes() bytes: the banded operand tables of all layers, both directions (bf16 copies
 *            of the kernels) and the biases; rebuild when a kernel changes.  w / b are HOST arrays of L device pointers.
 *   _fwd       x0 [B][T][ldx] fp32 (the first F values of a row are the spectrum) -> maps, a_last
 *   _bwd       d_last = dL/da_L [B][T][F][4] (fp32, or bf16 when d_bf16) -> per-workgroup partial sums of dW_l / db_l as rows
 *              [L][*nblocks_out][*npart_out] (a row of layer l: KT*KF*Cin_l*4 kernel entries, then 4 bias entries) for
 *              ptts_conv2d_reduce_grouped; fixed summation order (no atomics)
 *   _bwd_data  the backward-data chain alone: g0 = d/dx0 [B][T][F] fp32 (may be NULL) and, when gmaps is given, the masked
 *              gradient maps gamma_l = lrelu'(a_l) . dL/da_l, l = 1 .. L, [L][B][T][FP][4] bf16 (operands of _second)
 *   _second    the backward of _bwd_data w.r.t. d_last and the kernels: u0 = d/d(g0) [B][T][F] fp32 -> out = d/d(d_last)
 *              [B][T][F][4] (fp32 / bf16) and the partial sums of dW_l (rows as _bwd, bias entries zero)
 * ------------------------------------------------------------------------------------- */
int ptts_conv2d_chain_supported(int F, int L, int Cin0, int C, int KT, int KF);
int ptts_conv2d_chain_debug(void* stamp_buf);   /* measurement hook of tools/chain_probe.py: 256 x 32 uint64 phase stamps, NULL = off */
size_t ptts_conv2d_chain_tables_bytes(void);
size_t ptts_conv2d_chain_partials_bytes(int L);
long long ptts_conv2d_chain_map_elems(int B, int T, int F);
int ptts_conv2d_chain_tables(const float* const* w, const float* const* b, void* tables, int L, int cin0, void* stream);
int ptts_conv2d_chain_fwd(const float* x0, long long ldx, const void* tables, void* maps, void* a_last,
                          int B, int T, int F, int L, float alpha, void* stream);
int ptts_conv2d_chain_bwd(const void* d_last, int d_bf16, const float* x0, long long ldx, const void* maps, const void* a_last,
                          const void* tables, float* partials, size_t partials_bytes, int* nblocks_out, int* npart_out,
                          int B, int T, int F, int L, int cin0, float alpha, void* stream);
int ptts_conv2d_chain_bwd_data(const void* d_last, int d_bf16, const void* maps, const void* a_last, const void* tables,
                               void* gmaps, float* g0, int B, int T, int F, int L, float alpha, void* stream);
int ptts_conv2d_chain_second(const float* u0, const void* gmaps, const void* maps, const void* a_last, const void* tables,
                             void* out, int out_bf16, float* partials, size_t partials_bytes, int* nblocks_out, int* npart_out,
                             int B, int T, int F, int L, int cin0, float alpha, void* stream);

/* ---------------------------------------------------------------------------------------
 * fp32 GEMM on the MFMA pipe (v_mfma_f32_32x32x2_f32), with implicit-convolution row
 * addressing for the context Conv1D.  Replaces keras Dense (networktts.py:60; heads at
 * modeltts_common.py:84,95,121; networks_critic.py:96) and kl.Conv1D (networktts.py:117).
 *
 *   C[M,N] (+)= opA(A)[M,K] . opB(B)[K,N] (+ bias[N])
 *
 * A element (m,k):   transA==0:  A[(m / rows_per_seg)*seg_stride + (m % rows_per_seg)*lda + k]
 *                    transA==1:  A[(k / rows_per_seg)*seg_stride + (k % rows_per_seg)*lda + m]
 *   (rows_per_seg = T, seg_stride = (T+KW-1)*Cin, lda = Cin, K = KW*Cin walks a zero-padded
 *    [B, T+KW-1, Cin] frame buffer as the im2col matrix of a 'same' Conv1D without building it;
 *    a plain matrix has rows_per_seg = its row count, seg_stride = 0.)
 * B element (k,n):   transB==0: B[k*ldb + n] ;  transB==1: B[n*ldb + k]
 * The input transform acts on the stored A element; its channel index is the stored column.
 * accumulate != 0 adds to C (beta = 1).
 * out_mask (NULL or laid out like C with ldc): the product is multiplied by (out_mask>0 ? 1 : alpha) before it is
 * stored -- the LeakyReLU mask of a dense layer's backward-data pass, fused into the epilogue.
 * colsum_b (NULL or [N], needs transA==1 and transB==0): receives sum_k B[k,n] -- the bias gradient that goes with a
 * weight-gradient product dW = a^T . dy, taken from the B tiles while they are staged (fp32 atomics across workgroups). */
int ptts_gemm(const float* A, const float* Bm, const float* bias, float* C,
              int M, int N, int K,
              int transA, long long lda, long long rows_per_seg, long long seg_stride,
              int transB, long long ldb, long long ldc,
              int in_mode, const float* in_scale, const float* in_shift, const float* mask_src,
              float alpha, int accumulate, const float* out_mask, float* colsum_b, void* stream);

/* Weight gradients of several layers in ONE launch: for every product  C[M,N] += T(A)[K,M]^T . B[K,N]  and, when
 * colsum_b is given,  colsum_b[N] += sum_k B[k,:]  (the bias gradient).  A is the layer input as stored ([K rows][lda]),
 * B the incoming gradient; T is the same fused transform as in ptts_gemm (in_mode on the stored A element, channel =
 * stored column).  Every C / colsum_b is ACCUMULATED into with fp32 atomics (zero it beforehand for a plain product):
 * the caller's gradient buffers, as keras accumulates the gradients of the layers of one K.gradients call.  One
 * workgroup grid walks the (product, tile, k-step) space, so a 256x256x25600 product no longer pays its own launch,
 * zero-fill and 128-way split. */
typedef struct ptts_wgrad_desc {
    const float* A; const float* B; float* C; float* colsum_b;
    const float* in_scale; const float* in_shift; const float* mask_src;
    int M, N, K;
    long long lda, ldb, ldc;
    int in_mode; float alpha;
} ptts_wgrad_desc;
int ptts_gemm_wgrad_grouped(const ptts_wgrad_desc* descs, int n, void* stream);

/* The Dense products with M >> N (reference networktts.py:59-63 pFC -> kl.Dense, networks_critic.py:86-93, the LSTM input
 * projections of networktts.py:85-96, and TF's MatMul gradients) as fp32 products on the bf16 matrix cores by the three-way
 * operand split of split.hip ("bf16x6": six bf16 MFMA products, fp32 accumulation; dense.hip).  The weight operand B[K][N] is
 * split once per update into fragment-ordered planes (ptts_split3_dense_weight; transposed = 1 reads w as [N][K], which
 * makes dX = dY.W^T read W as it lies); ptts_dense_bf16x6 then has the contract of ptts_gemm with transA = 0:
 * C[M,N] (+)= T(A).B (+ bias), C *= (out_mask > 0 ? 1 : alpha), T = in_mode (PTTS_IN_*) with in_scale/in_shift [K] or
 * mask_src laid out like A.  N, K, lda, ldc multiples of 4, operands 16-byte aligned (ptts_dense_bf16x6_supported). */
size_t ptts_dense_planes_bytes(int N, int K);
int ptts_split3_dense_weight(const float* w, long long ldw, int K, int N, int transposed, void* planes, void* stream);
/* The same for n weights in ONE launch: after an optimiser update every Dense kernel of a network needs its planes again (forward and
 * transposed), 18 launches of 4 us per critic step otherwise. */
typedef struct ptts_dense_split_desc {
    const float* w; void* planes;
    long long ldw;
    int K, N, transposed, reserved;
} ptts_dense_split_desc;
int ptts_split3_dense_weight_grouped(const ptts_dense_split_desc* descs, int n, void* stream);
/* Planes of windows of frame sequences x [B][T][C] (the segments of the overlap-save form of the frequency-domain Conv1D): matrix
 * z = b*NS + s = the P rows x[b][row_off + s*S + k][:], k < P; rows outside [0, T) and k >= kvalid are zero (never read). */
int ptts_split3_frame_windows(const float* x, int B, int T, int C, int NS, int S, int row_off, int P, int kvalid,
                              void* planes, long long stride_planes_bytes, void* stream);
/* n weights of ONE shape at regular strides: w + i*stride_w (floats) -> planes + i*stride_planes_bytes. */
int ptts_split3_dense_weight_strided(const float* w, long long stride_w, void* planes, long long stride_planes_bytes, int n,
                                     long long ldw, int K, int N, int transposed, void* stream);
int ptts_dense_bf16x6_supported(int M, int N, int K, long long lda, long long ldc);
/* One weight-gradient product of ptts_gemm_wgrad_grouped (TF's MatMul gradient w.r.t. the kernel of a kl.Dense / LSTM
 * projection, plus the bias gradient) as a bf16x6 split product, in two stages without atomics between workgroups:
 * ptts_dense_wgrad_bf16x6_partials writes every workgroup's partial tile of dW[Kin,N] = T(A)[M,Kin]^T . dY[M,N] (and of the
 * column sums of dY) as a row of `workspace` (both operands split on their way into the LDS and read transposed);
 * ptts_dense_wgrad_reduce_grouped sums the rows of up to 16 products per launch in a fixed order and ADDS them into the
 * caller's gradient buffers (one fp32 atomic per element: products of one backward pass may share a buffer).
 * ptts_dense_wgrad_bf16x6 runs both stages for one product. */
typedef struct ptts_dense_wgrad_reduce_desc {
    const float* partials;                   /* the workspace of ptts_dense_wgrad_bf16x6_partials */
    int split, Kin, N;                       /* *split_out of that call; the product's dims */
    long long ldc;
    float* C; float* colsum_b;               /* C[Kin][ldc] += dW ; colsum_b[N] += column sums of dY (or NULL) */
} ptts_dense_wgrad_reduce_desc;
int ptts_dense_wgrad_bf16x6_supported(int Kin, int N, int M, long long lda, long long ldb);
size_t ptts_dense_wgrad_workspace_bytes(int Kin, int N, int M);
int ptts_dense_wgrad_bf16x6_partials(const float* A, const float* dY, const float* mask_src, const float* in_scale,
                                     const float* in_shift, void* workspace, size_t workspace_bytes, int* split_out,
                                     int Kin, int N, int M, long long lda, long long ldb, int in_mode, float alpha,
                                     void* stream);
int ptts_dense_wgrad_reduce_grouped(const ptts_dense_wgrad_reduce_desc* descs, int n, void* stream);
int ptts_dense_wgrad_bf16x6(const float* A, const float* dY, const float* mask_src, const float* in_scale,
                            const float* in_shift, float* C, float* colsum_b, void* workspace, size_t workspace_bytes,
                            int Kin, int N, int M, long long lda, long long ldb, long long ldc, int in_mode, float alpha,
                            void* stream);
int ptts_dense_bf16x6(const float* A, const void* planes, const float* bias, float* C, int M, int N, int K,
                      long long lda, long long ldc, int in_mode, const float* in_scale, const float* in_shift,
                      const float* mask_src, float alpha, int accumulate, const float* out_mask, void* stream);
/* ... + res[m % res_rows][n] (row stride ldr floats) added in the store, res_rows <= M <= 8 res_rows: the product of a concatenation part
 * shared by k stacked evaluations (the critic's context branch, networks_critic.py:78-86, computed once at B rows) joins each of the k
 * row blocks without an add pass.  res == C with res_rows >= M is ptts_dense_bf16x6's accumulate. */
int ptts_dense_bf16x6_res(const float* A, const void* planes, const float* bias, float* C, int M, int N, int K,
                          long long lda, long long ldc, int in_mode, const float* in_scale, const float* in_shift,
                          const float* mask_src, float alpha, const float* res, int res_rows, long long ldr,
                          const float* out_mask, void* stream);
/* The product of a kl.Dense that feeds a kl.BatchNormalization (pFC, networktts.py:59-63) also leaves, per row tile, the column sums and
 * the column sums of squares of what it stores: stats[*nrows_out][2 N] doubles, finished by ptts_bn_finalize_partials -- TF's batch
 * moments are a pass of their own over the activation.  capacity_rows >= ptts_dense_bf16x6_stats_rows(M, N); N % 4 == 0. */
int ptts_dense_bf16x6_stats_rows(int M, int N);
int ptts_dense_bf16x6_stats(const float* A, const void* planes, const float* bias, float* C, int M, int N, int K,
                            long long lda, long long ldc, int in_mode, const float* in_scale, const float* in_shift,
                            float alpha, double* stats, int capacity_rows, int* nrows_out, void* stream);
/* Backward-data product of a kl.Dense whose input was the kl.BatchNormalization + kl.LeakyReLU of pFC (networktts.py:59-63), i.e.
 * lrelu(scale z + shift):  C = dz = (A . B) lrelu'(scale z + shift) scale  with A = dy [M, K], B = W^T (planes of the transposed kernel,
 * [K, N]), z [M, N]; and per row tile the column sums of (A . B) lrelu'(.) z and of (A . B) lrelu'(.) -- the gradients of scale and
 * shift -- as stats[*nrows_out][2 N] doubles, which ptts_partial_rows_sum adds.  What TF runs as MatMul + LeakyReluGrad + the reductions of
 * FusedBatchNormGrad's first stage; replaces ptts_dense_bf16x6 + ptts_affine_act_bwd (a pass over da and z). */
int ptts_dense_bf16x6_bwd_affine(const float* A, const void* planes, float* C, int M, int N, int K, long long lda, long long ldc,
                                 const float* z, const float* scale, const float* shift, float alpha,
                                 double* stats, int capacity_rows, int* nrows_out, void* stream);
/* out[j] = sum_r partials[r][j] (doubles), rows added in index order. */
int ptts_partial_rows_sum(const double* partials, int nrows, int ncols, double* out, void* stream);
/* Frequency-domain context Conv1D, helper: Ap [NB][2][B][2*Kh] floats with [Xr | .] in part 0 and [Xi | .] in part 1 (columns < Cin)
 * -> columns Kh .. Kh+Cin-1 get -Xi (part 0) and Xr (part 1): the rows [Xr | -Xi], [Xi | Xr] of the real form of a complex product. */
int ptts_dft_mirror(float* Ap, int NB, int B, int Cin, int Kh, void* stream);

/* Frequency-domain context Conv1D, the kernel's side: planes (ptts_dense_bf16x6_batched layout, ptts_dense_planes_bytes(N, 2*Kh) per
 * frequency) of [Wr_f ; Wi_f] [2*Kh][N], W^_f = sum_k w[k] tw[(f,.)][k], for f < NB, from w [KW][Cin][N] (KW in 3, 5, 7, 9, 11, 21) and
 * the twiddle rows tw [2*NB][KW] (row 2f: cos, row 2f+1: -sin of 2 pi f (pl - k) / P).  Once per weight update. */
int ptts_conv1d_freq_kernel_planes(const float* w, const float* tw, void* planes, int NB, int KW, int Cin, int N, int Kh, void* stream);

/* dst[z][c][r] = src[z][r][c], nb matrices of rows x cols floats. */
int ptts_transpose_batched(const float* src, float* dst, int nb, int rows, int cols, void* stream);
/* Frequency-domain context Conv1D, weight gradient, last step: dW[k][c][n] = out2[k][n*2Kh + c] + out2[KW+k][n*2Kh + Kh + c], out2
 * [2*KW][N*2*Kh] = the product of the inverse twiddles with the per-frequency correlations (TF's Conv1D kernel backprop). */
int ptts_conv1d_freq_wgrad_combine(const float* out2, float* dW, int KW, int Cin, int N, int Kh, void* stream);

/* ... or both steps in one pass over the correlations Gt [NB][N][2*Kh] (KW in 3, 5, 7, 9, 11, 21; t2 [NB][NBp], NBp >= 2*KW: row f =
 * the KW cosine then the KW sine coefficients of frequency f): dW[k][c][n] = sum_f t2[f][k] Gt[f][n][c] + t2[f][KW+k] Gt[f][n][Kh+c]; the frequencies are shared out over four groups of workgroups
 * whose partial sums (workspace) are added in a fixed order: no atomics. */
size_t ptts_conv1d_freq_wgrad_inverse_workspace_bytes(int KW, int Cin, int N);
int ptts_conv1d_freq_wgrad_inverse(const float* Gt, const float* t2, float* dW, void* workspace, size_t workspace_bytes,
                                   int NB, int NBp, int KW, int Cin, int N, int Kh, void* stream);

/* nbatch products of one shape in one launch: C_z[M,N] = A_z[M,K] . B_z (+ bias), A_z = A + z*strideA and C_z = C + z*strideC (floats;
 * strideA = 0 shares the left operand), B_z = the planes at planes + z*stride_planes_bytes (ptts_split3_dense_weight[_grouped] layout).
 * planes_count 3 = fp32 arithmetic (six bf16 products), 1 = one bf16 product.  The building block of the frequency-domain context
 * Conv1D (kl.Conv1D of networktts.py:116-120 as DFT -> per-frequency products -> inverse DFT). */
int ptts_dense_bf16x6_batched(const float* A, long long strideA, const void* planes, long long stride_planes_bytes,
                              const float* bias, float* C, long long strideC, int nbatch, int M, int N, int K,
                              long long lda, long long ldc, int planes_count, void* stream);

/* The context Conv1D forward (reference networktts.py:116-120: kl.Conv1D(width, winlen, padding='same')) as an fp32
 * product on the bf16 matrix cores by a three-way operand split ("bf16x6"; split.hip): x = x1 + x2 + x3 in bf16 (round
 * to nearest), six bf16 products per operand pair, fp32 accumulation -- fp32-level accuracy at 2.7x less matrix-pipe time.
 *   ptts_split3_frames   x [B][T][C] fp32 -> three bf16 planes [Cp/32][B*(pad_left+T+pad_right)][32]: 32-channel blocks
 *                        of the time-padded frames, zero padded in time and in the channels C..Cp-1 (Cp % 32 == 0);
 *   ptts_split3_weight_t w [KW][C][N] fp32 -> three bf16 planes [Cp/32][N][KW][32] (transposed, zero channels C..Cp-1);
 *   ptts_conv1d_bf16x6   y[b][t][n] = bias[n] + sum_{j,c} x[b][t + j - pad_left][c] w[j][c][n] from those planes, with
 *                        pad_left + pad_right = KW - 1.  N % 128 == 0, Cp % 32 == 0, T >= 128 (or B == 1), KW <= 24. */
int ptts_split3_frames(const float* x, void* p1, void* p2, void* p3, int B, int T, int C, int pad_left, int pad_right,
                       int Cp, void* stream);
int ptts_split3_weight_t(const float* w, void* p1, void* p2, void* p3, int KW, int C, int N, int Cp, void* stream);
int ptts_conv1d_bf16x6(const void* a1, const void* a2, const void* a3, const void* bt1, const void* bt2, const void* bt3,
                       const float* bias, float* y, int B, int T, int KW, int Cp, int N, void* stream);

/* The context Conv1D weight gradient (TF's Conv1D kernel backprop of networktts.py:116-120) as a bf16x6 split product:
 *   dw[j][c][n] = sum_{b,t} xp[b][t + j][c] dy[b][t][n]   (xp: the frames zero-padded in time to T + KW - 1).
 *   ptts_split3_frames_t  x [B][T][C] fp32 -> three FRAME-MAJOR bf16 planes [Crows][Pp], element (c, b Tp + pad_left + t),
 *                         zero elsewhere (Crows % 32 == 0, >= C; Pp % 64 == 0, >= B Tp).  Used for the padded frames
 *                         (B, T + KW - 1, pad_left 0, Tp = T + KW - 1) and for dy (B, T, pad_left 0, Tp = T + KW - 1);
 *   ptts_conv1d_wgrad_bf16x6  dw (overwritten) from those planes; KW in {3, 5, 21}, N % 32 == 0, Crows % 64 == 0,
 *                         Pp >= 32 ceil(B (T + KW - 1) / 32) + 64. */
int ptts_split3_frames_t(const float* x, void* p1, void* p2, void* p3, int B, int T, int C, int pad_left, int Tp, int Crows,
                         long long Pp, void* stream);
int ptts_conv1d_wgrad_bf16x6(const void* xt1, const void* xt2, const void* xt3, const void* yt1, const void* yt2,
                             const void* yt3, float* dw, int B, int T, int KW, int C, int N, int Crows, long long Pp,
                             void* stream);

/* The same weight gradient in exact fp32 (v_mfma_f32_16x16x4_f32) over frame-major fp32 operands (conv1d_wgrad.hip):
 *   ptts_transpose_frames  x [B][T][C] -> out [Crows][Pp], element (c, b Tp + pad_left + t), zero elsewhere;
 *   ptts_conv1d_wgrad_t    dw (overwritten; and db[n] = sum_{b,t} dy[b][t][n] when db != NULL) from xt = the transposed
 *                          padded frames and yt = the transposed gradient (Tp = T + KW - 1 for both).  KW in {3, 5, 21},
 *                          N % 32 == 0, Crows % 64 == 0, Pp % 64 == 0, Pp >= 32 ceil(B (T + KW - 1) / 32) + 64. */
int ptts_transpose_frames(const float* x, float* out, int B, int T, int C, int pad_left, int Tp, int Crows, long long Pp,
                          void* stream);
int ptts_conv1d_wgrad_t(const float* xt, const float* yt, float* dw, float* db, int B, int T, int KW, int C, int N, int Crows,
                        long long Pp, void* stream);

/* ---------------------------------------------------------------------------------------
 * channel-last reductions and elementwise passes
 * ------------------------------------------------------------------------------------- */
/* sums[0:C] = sum_r a[r,c], sums[C:2C] = sum_r a[r,c]^2 in fp64, a = transform(x); deterministic two-stage. */
size_t ptts_colstats_workspace_bytes(long long rows, int C);
int ptts_colstats(const float* x, long long rows, int C,
                  int in_mode, const float* in_scale, const float* in_shift, const float* mask_src, float alpha,
                  double* sums /*[2C]*/, void* workspace, size_t workspace_bytes, void* stream);

/* Keras BatchNormalization (axis=-1, eps, momentum) statistics -> per-channel affine.
 * Replaces kl.BatchNormalization at networktts.py:61,118,124,132.
 *   training: mean,var from sums (biased var); scale = gamma*rsqrt(var+eps); shift = beta-mean*scale;
 *             moving stats updated in place with `momentum` (moving_var uses var*count/(count-1) when unbiased_moving!=0)
 *             when update_moving != 0.   inference: same from moving stats. */
int ptts_bn_finalize(const double* sums, long long count, const float* gamma, const float* beta,
                     float* moving_mean, float* moving_var, float eps, float momentum,
                     int training, int update_moving, int unbiased_moving, int C,
                     float* scale, float* shift, float* mean /*[C] out*/, float* rstd /*[C] out*/, void* stream);

/* Training-mode statistics AND the affine of a few-channel map (C = 4, 8, 16: the conv stacks' BatchNormalization, networktts.py:124)
 * in one launch: ptts_colstats + ptts_bn_finalize(training = 1) fused (the workgroup that finishes last adds the partial rows in a
 * fixed order: bit-reproducible).  workspace: ptts_colstats_workspace_bytes(rows, C); counter: one int, zero before the first call,
 * left zero by every call, not shared between streams. */
int ptts_bn_batch_stats_supported(long long rows, int C);
int ptts_bn_batch_stats(const float* x, long long rows, int C, const float* gamma, const float* beta,
                        float* moving_mean, float* moving_var, float eps, float momentum, int update_moving, int unbiased_moving,
                        float* scale, float* shift, float* mean /*[C] out*/, float* rstd /*[C] out*/,
                        void* workspace, size_t workspace_bytes, int* counter, void* stream);

/* ptts_bn_finalize(training = 1) for sums that lie as nrows partial rows [2 C] doubles (C sums, C sums of squares), added in
 * index order: the finish of ptts_conv2d_mfma_fwd_stats and of ptts_dense_bf16x6_stats.  rows = the number of values per channel the sums cover. */
int ptts_bn_finalize_partials(const double* partials, int nrows, long long rows, int C, const float* gamma, const float* beta,
                              float* moving_mean, float* moving_var, float eps, float momentum, int update_moving, int unbiased_moving,
                              float* scale, float* shift, float* mean /*[C] out*/, float* rstd /*[C] out*/, void* stream);

/* BatchNorm backward through the batch statistics.  Given dscale/dshift (gradients w.r.t. the affine that
 * ptts_bn_finalize produced in training mode):  dgamma = (dscale - dshift*mean)*rstd ;  dbeta = dshift ;
 *   dmean = -dshift*gamma*rstd ;  dvar = -0.5*(dscale - dshift*mean)*gamma*rstd^3 ;
 *   c2 = 2*dvar/count ;  c0 = dmean/count - c2*mean      so that   dz[r,c] += c0[c] + c2[c]*z[r,c]. */
int ptts_bn_bwd_coefs(const float* dscale, const float* dshift, const float* mean, const float* rstd,
                      const float* gamma /*NULL = 1*/, long long count, int C,
                      float* dgamma, float* dbeta, float* c0, float* c2, void* stream);
/* ... with dgamma / dbeta ADDED into the buffers given (the parameters' gradient buffers: no separate accumulation launch per parameter) */
int ptts_bn_bwd_coefs_acc(const float* dscale, const float* dshift, const float* mean, const float* rstd,
                      const float* gamma /*NULL = 1*/, long long count, int C,
                      float* dgamma, float* dbeta, float* c0, float* c2, void* stream);

/* y = act(x*scale[c]+shift[c]) materialised (used where no consumer can fuse it: LSTM input, final outputs). */
int ptts_affine_act(const float* x, const float* scale, const float* shift, float* y,
                    long long rows, int C, int act, float alpha, void* stream);
/* dx = dy * act'(.) * scale ; dscale/dshift reductions (NULL to skip).  `y` is the forward output (sigmoid/tanh use it). */
int ptts_affine_act_bwd(const float* dy, const float* x, const float* y, const float* scale, const float* shift,
                        float* dx, double* dsums /*[2C]: dscale, dshift; or NULL*/,
                        void* workspace, size_t workspace_bytes,
                        long long rows, int C, int act, float alpha, void* stream);
/* The gated product of a gated convolution (networktts.py:128-134, pGCNN2D): y = a * sigmoid(b) over the two Conv2D
 * pre-activations, and its backward da = dy*s, db = dy*a*s*(1-s), s = sigmoid(b).  One pass each; n = element count. */
int ptts_gated_mul_fwd(const float* a, const float* b, float* y, long long n, void* stream);
int ptts_gated_mul_bwd(const float* dy, const float* a, const float* b, float* da, float* db, long long n, void* stream);

/* out[r,c] = (acc? out : 0) + a[r,c]*c1[c] + x[r,c]*c2[c] + c0[c]   (BatchNorm backward fix-up: dz += dmean/N + dvar*2(z-mean)/N) */
int ptts_axpby_cols(const float* a, const float* c1, const float* x, const float* c2, const float* c0,
                    float* out, long long rows, int C, void* stream);

/* ---------------------------------------------------------------------------------------
 * WGAN-GP pieces (optimizertts_wgan.py:44-79)
 * ------------------------------------------------------------------------------------- */
/* RandomWeightedAverage (:44-51): out = alpha_b*real + (1-alpha_b)*fake, alpha per sample */
int ptts_gp_interpolate(const float* real, const float* fake, const float* alpha_b /*[B]*/, float* out,
                        int B, long long TD, void* stream);
/* per-sample squared L2 norm over (T, D) (:58-62), wavefront-reduced; out[B] */
int ptts_gp_sqnorm(const float* g, float* out, int B, long long TD, void* stream);
/* penalty = mean_b (1 - sqrt(sq_b))^2 (:64-68);  coef[b] = dPenalty/dg scale = 2*(n_b-1)/(n_b*B)  (inf/nan at n_b=0 as in the reference) */
int ptts_gp_penalty(const float* sq /*[B]*/, float* penalty /*[1]*/, float* coef /*[B]*/, int B, void* stream);
/* dg = upstream[0] * coef[b] * g */
int ptts_gp_scale_rows(const float* g, const float* coef, const float* upstream /*[1] or NULL(=1)*/, float* dg,
                       int B, long long TD, void* stream);

/* wasserstein_loss (:70-71): out[0] = sign * mean(v) ; specweighted_lse_loss (:73-79): out[0] = mean((y-yhat)^2 * w[d]) */
int ptts_mean_scaled(const float* v, long long n, float sign, float* out, void* stream);
int ptts_wlse_fwd(const float* y, const float* yhat, const float* w /*[D] or NULL*/, float* out,
                  long long rows, int D, void* stream);
/* dyhat = upstream[0] * 2*(yhat-y)*w[d]/(rows*D) */
int ptts_wlse_bwd(const float* y, const float* yhat, const float* w, const float* upstream, float* dyhat,
                  long long rows, int D, void* stream);

/* weight clipping (north_star extra; the reference has no counterpart): p = clamp(p, lo, hi) */
int ptts_weight_clip(float* p, long long n, float lo, float hi, void* stream);

/* Keras-2.2 Adam (keras.optimizers.Adam, optimizertts_wgan.py:145,172):
 *   t = ++(*step);  lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
 *   p -= lr_t * m / (sqrt(v) + eps).   `step` lives on the device so the launch is graph-replayable;
 *   gscale multiplies g first (1/world_size after an all-reduce(sum)). */
int ptts_adam_keras_step(float* p, const float* g, float* m, float* v, long long n,
                         float lr, float b1, float b2, float eps, float gscale,
                         int* step /*device int, incremented by the kernel*/, void* stream);

/* ---------------------------------------------------------------------------------------
 * Keras LSTM (gates i,f,c,o; activation tanh, recurrent_activation sigmoid; networktts.py:72-96).
 * ndir = 1: one direction (reverse != 0 walks time backwards); ndir = 2: kl.Bidirectional(concat),
 * direction 0 forwards and direction 1 backwards in the SAME launches (networktts.py:85-96).
 *   xproj [B,T,ndir*4H] = x.[W_d0 | W_d1] + b (one ptts_gemm);  U [ndir,H,4H]
 *   h_out [B,T,ndir*H] (the Bidirectional concat layout), gates (post-nonlinearity) [B,T,ndir*4H]
 *   and cell states c_out [B,T,ndir*H] are kept for the backward.
 * ------------------------------------------------------------------------------------- */
size_t ptts_lstm_fwd_workspace_bytes(int B, int T, int H, int ndir);   /* packed recurrent kernel */
int ptts_lstm_fwd(const float* xproj, const float* U, float* h_out, float* gates, float* c_out,
                  void* workspace, size_t workspace_bytes,
                  int B, int T, int H, int ndir, int reverse, void* stream);
/* dgates [B,T,ndir*4H] out: gradients w.r.t. the gate PRE-activations; dW, dU, db and dx follow from
 * them as ptts_gemm products.  workspace: ptts_lstm_bwd_workspace_bytes */
size_t ptts_lstm_bwd_workspace_bytes(int B, int T, int H, int ndir);
int ptts_lstm_bwd(const float* dh_out /*[B,T,ndir*H]*/, const float* U, const float* gates, const float* c_out,
                  float* dgates, void* workspace, size_t workspace_bytes,
                  int B, int T, int H, int ndir, int reverse, void* stream);
/* Optionally the T step launches of a recurrence go out as ONE hipGraph launch: the chain of a (pointers, shape) tuple is captured once and
 * replayed while the same addresses come back (a training loop's allocation pattern repeats); misses capture anew (16 entries,
 * least recently used dropped), repeated misses fall back to plain launches, a call inside a stream capture joins that capture.
 * Off by default; PTTS_LSTM_GRAPH=1 or ptts_set_lstm_graph(1) switch it on.  Counters since load: replays, captures,
 * plain-launch fallbacks. */
int ptts_set_lstm_graph(int on);
int ptts_lstm_graph_stats(unsigned long long* hits, unsigned long long* captures, unsigned long long* direct);
int ptts_lstm_graph_clear(void);

#ifdef __cplusplus
}
#endif
#endif /* PERCIVAL_HIP_H */

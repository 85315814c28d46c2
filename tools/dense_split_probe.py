"""csrc/dense.hip against the fp32-MFMA tall kernel: HIP-event timings of the Dense shapes of the train step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops, layers
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
for (M, N, K, tb, mode) in ((25600, 256, 256, 0, ops.IN_LRELU), (25600, 256, 256, 1, ops.IN_NONE), (51200, 256, 256, 0, ops.IN_LRELU),
                            (25600, 256, 260, 0, ops.IN_NONE), (25600, 260, 256, 1, ops.IN_NONE), (25600, 2048, 256, 0, ops.IN_NONE),
                            (25600, 256, 2048, 1, ops.IN_NONE), (25600, 20, 256, 0, ops.IN_LRELU), (25600, 256, 256, 0, ops.IN_MASKMUL)):
    A = torch.randn(M, K, generator=g).cuda()
    class H(torch.nn.Module):
        def __init__(s):
            super().__init__(); s.w = torch.nn.Parameter((torch.randn(N, K, generator=g) if tb else torch.randn(K, N, generator=g)) / K ** 0.5)
    h = H(); layers.FlatParams(h, 'cuda'); w = h.w
    C = torch.empty(M, N, device='cuda'); msk = torch.randn(M, K, generator=g).cuda() if mode == ops.IN_MASKMUL else None
    om = torch.randn(M, N, generator=g).cuda() if tb else None
    def run(): ops.gemm_raw(A, w, C, M, N, K, transB=tb, ldb=(K if tb else N), mode=mode, mask_src=msk, alpha=0.3, out_mask=om)
    ops.dense_split(False); t32 = timeit(run); c32 = C.clone()
    ops.dense_split(True); t6 = timeit(run)
    err = float((C - c32).abs().max() / c32.abs().mean())
    fl = 2.0 * M * N * K
    print('M {:6d} N {:5d} K {:5d} tb {} mode {}: fp32 {:7.1f} us ({:5.1f} TF)   bf16x6 {:7.1f} us ({:5.1f} TF)   max diff / mean {:.2e}'.format(
        M, N, K, tb, mode, t32, fl / t32 / 1e6, t6, fl / t6 / 1e6, err))
ops.dense_split(None)
print('weight gradients dW[Kin,N] = A^T . dY over M frames:')
for (Kin, N, M, mode) in ((256, 256, 25600, ops.IN_LRELU), (256, 256, 51200, ops.IN_LRELU), (260, 256, 25600, ops.IN_NONE), (256, 2048, 25600, ops.IN_NONE),
                          (256, 1024, 25600, ops.IN_NONE), (256, 256, 25600, ops.IN_MASKMUL)):
    A = torch.randn(M, Kin, generator=g).cuda(); dY = torch.randn(M, N, generator=g).cuda()
    msk = torch.randn(M, Kin, generator=g).cuda() if mode == ops.IN_MASKMUL else None
    C = torch.empty(Kin, N, device='cuda'); db = torch.empty(N, device='cuda')
    def run(): ops.gemm_raw(A, dY, C, Kin, N, M, transA=1, lda=Kin, rows_per_seg=M, mode=mode, mask_src=msk, alpha=0.3, colsum_b=db)
    ops.dense_split(False); t32 = timeit(run); c32 = C.clone()
    ops.dense_split(True); t6 = timeit(run)
    err = float((C - c32).abs().max() / c32.abs().mean())
    fl = 2.0 * M * N * Kin
    print('Kin {:5d} N {:5d} M {:6d} mode {}: fp32 {:7.1f} us ({:5.1f} TF)   bf16x6 {:7.1f} us ({:5.1f} TF)   max diff / mean {:.2e}'.format(
        Kin, N, M, mode, t32, fl / t32 / 1e6, t6, fl / t6 / 1e6, err))
ops.dense_split(None)

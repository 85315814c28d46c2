"""Per-launch times of the critic's 4 -> 4 Conv2D kernels at [64,400,65,4] / [128,400,65,4] (tools/ab_c2m.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops, _hip
def t_us(fn, n=40):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
g = torch.Generator().manual_seed(3)
out = []
for B in (64, 128, 192):
    x = torch.randn(B, 400, 65, 4, generator=g).cuda(); dy = torch.randn(B, 400, 65, 4, generator=g).cuda()
    w = (torch.randn(5, 5, 4, 4, generator=g) * 0.2).cuda()
    tf, tb = ops._C2M.table(w, False), ops._C2M.table(w, True)
    fwd = t_us(lambda: ops._conv2d_mfma_fwd(x, w, tf, None, None, None, None, None, ops.IN_LRELU, 0.3, 1, 2))
    dxm = t_us(lambda: ops._conv2d_mfma_fwd(dy, w, tb, None, None, None, None, x, ops.IN_NONE, 0.3, 1, 2))
    k1 = t_us(lambda: ops._conv2d_mfma_bwd_fused(1, dy, x, None, w, 0.3))
    k2 = t_us(lambda: ops._conv2d_mfma_bwd_fused(2, dy, x, x, w, 0.3))
    out.append('B=%d fwd %.1f dx %.1f fused1 %.1f fused2 %.1f' % (B, fwd, dxm, k1, k2))
print(' | '.join(out))

"""How much of a Dense launch is ramp: ptts_dense_bf16x6 at 256 x 256 for M = 6400 .. 153600 rows (forward with LeakyReLU + bias), and back-to-back chains."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops, layers
def t_us(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
g = torch.Generator().manual_seed(1)
class H(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.randn(256, 256, generator=g) / 16)
h = H(); layers.FlatParams(h, 'cuda'); W = h.w
bias = torch.randn(256, generator=g).cuda()
for M in (6400, 12800, 25600, 51200, 76800, 102400, 153600):
    A = torch.randn(M, 256, generator=g).cuda(); C = torch.empty(M, 256, device='cuda')
    t = t_us(lambda: ops.gemm_raw(A, W, C, M, 256, 256, bias=bias, mode=ops.IN_LRELU))
    print('M = %6d  %.1f us  (%.1f TF fp32-equivalent, %.2f TB/s of A + C)' % (M, t, 2.0 * M * 65536 / t / 1e6, M * 2048 / t / 1e6))

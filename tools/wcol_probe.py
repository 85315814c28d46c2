import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops, _hip
M, H = 25600, 256
x = torch.randn(M, H, device='cuda'); dy = torch.randn(M, 1, device='cuda'); dw = torch.empty(H, 1, device='cuda')
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    with _hip.KernelTimer() as kt:
        for _ in range(n): fn()
    d = sorted(t for (_, _, t) in kt.durations_ms()); return d[len(d)//2] * 1e3
for mode in (0, 1, 2):
    us = timed(lambda: ops.gemm_raw(x, dy, dw, H, 1, M, transA=1, lda=H, rows_per_seg=M, mode=mode, mask_src=x if mode == 2 else None))
    print('wcol mode', mode, round(us, 1), 'us')

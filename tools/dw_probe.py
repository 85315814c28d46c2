"""Weight-gradient product of a 256-wide Dense layer (256 x 256 x K) for K sweeps / rocprofv3 --kernel-trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops, _hip
H = 256
Ks = [int(a) for a in sys.argv[1:]] or [25600]
for M in Ks:
    x = torch.randn(M, H, device='cuda'); dy = torch.randn(M, H, device='cuda'); dw = torch.empty(H, H, device='cuda')
    fn = lambda: ops.gemm_raw(x, dy, dw, H, H, M, transA=1, lda=H, rows_per_seg=M, mode=int(os.environ.get('MODE', 1)), mask_src=x if os.environ.get('MODE') == '2' else None)
    fn(); torch.cuda.synchronize()
    with _hip.KernelTimer() as kt:
        for _ in range(20): fn()
    d = sorted(t for (_, _, t) in kt.durations_ms())
    print('K', M, 'us', round(d[len(d) // 2] * 1e3, 1))

"""K sweep of the tall (M=25600, N=256) dense product: slope = per-k-step cost, intercept = fixed cost per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops, _hip
M, N = 25600, 256
for K in (64, 128, 256, 512, 1024, 2048):
    x = torch.randn(M, K, device='cuda'); w = torch.randn(K, N, device='cuda') * 0.05; y = torch.empty(M, N, device='cuda')
    fn = lambda: ops.gemm_raw(x, w, y, M, N, K)
    fn(); torch.cuda.synchronize()
    with _hip.KernelTimer() as kt:
        for _ in range(20): fn()
    d = sorted(t for (_, _, t) in kt.durations_ms())
    us = d[len(d) // 2] * 1e3
    print('K', K, 'us', round(us, 1), 'TF', round(2.0 * M * N * K / us / 1e6, 1))

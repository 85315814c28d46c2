"""BatchNorm statistics of a [64,400,65,4] map: the one-launch form (ptts_bn_batch_stats) against colstats + bn_finalize.  (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops, _hip
lib = _hip.lib()
z = torch.randn(64, 400, 65, 4, device='cuda')
C, rows = 4, z.numel() // 4
gamma = torch.ones(C, device='cuda'); beta = torch.zeros(C, device='cuda')
mm = torch.zeros(C, device='cuda'); mv = torch.ones(C, device='cuda')
scale, shift, mean, rstd = [torch.empty(C, device='cuda') for _ in range(4)]
ws = torch.empty(lib.ptts_colstats_workspace_bytes(rows, C), dtype=torch.uint8, device='cuda')
cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
sums = torch.empty(2 * C, dtype=torch.float64, device='cuda')
for rep in range(3):
    with _hip.KernelTimer() as kt:
        ops.call('ptts_bn_batch_stats', ops.ptr(z), rows, C, ops.ptr(gamma), ops.ptr(beta), ops.ptr(mm), ops.ptr(mv), 1e-3, 0.99, 1, 0,
                 ops.ptr(scale), ops.ptr(shift), ops.ptr(mean), ops.ptr(rstd), ops.ptr(ws), ws.numel(), ops.ptr(cnt), ops.stream())
        ops.call('ptts_colstats', ops.ptr(z), rows, C, 0, None, None, None, 0.3, ops.ptr(sums), ops.ptr(ws), ws.numel(), ops.stream())
        ops.call('ptts_bn_finalize', ops.ptr(sums), rows, ops.ptr(gamma), ops.ptr(beta), ops.ptr(mm), ops.ptr(mv), 1e-3, 0.99, 1, 1, 0, C,
                 ops.ptr(scale), ops.ptr(shift), ops.ptr(mean), ops.ptr(rstd), ops.stream())
print(' '.join('%s %.1f' % (n.replace('ptts_', ''), t * 1e3) for n, _, t in kt.durations_ms()))

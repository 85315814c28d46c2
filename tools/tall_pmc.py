import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops
M, N, K = 25600, 256, int(os.environ.get('K', 2048))
x = torch.randn(M, K, device='cuda'); w = torch.randn(K, N, device='cuda') * 0.05; y = torch.empty(M, N, device='cuda')
for _ in range(4): ops.gemm_raw(x, w, y, M, N, K)
torch.cuda.synchronize()

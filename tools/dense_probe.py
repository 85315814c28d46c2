import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops, _hip
M, H = 25600, 256
x = torch.randn(M, H, device='cuda'); w = torch.randn(H, H, device='cuda') * 0.05; dy = torch.randn(M, H, device='cuda')
w516 = torch.randn(516, H, device='cuda') * 0.05; x260 = torch.randn(M, 260, device='cuda')
y = torch.empty(M, H, device='cuda'); dw = torch.empty(H, H, device='cuda'); b = torch.randn(H, device='cuda')
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    with _hip.KernelTimer() as kt:
        for _ in range(n): fn()
    d = sorted(t for (_, _, t) in kt.durations_ms()); return d[len(d)//2] * 1e3
cases = {
 'fwd  y=lrelu(x)W+b  25600x256x256': (lambda: ops.gemm_raw(x, w, y, M, H, H, bias=b, mode=ops.IN_LRELU), 2.0*M*H*H),
 'fwd  y=xW (NONE)': (lambda: ops.gemm_raw(x, w, y, M, H, H), 2.0*M*H*H),
 'da = dy W^T (+mask)': (lambda: ops.gemm_raw(dy, w, y, M, H, H, transB=1, ldb=H, out_mask=x), 2.0*M*H*H),
 'da = dy W^T': (lambda: ops.gemm_raw(dy, w, y, M, H, H, transB=1, ldb=H), 2.0*M*H*H),
 'dW = a^T dy 256x256x25600': (lambda: ops.gemm_raw(x, dy, dw, H, H, M, transA=1, lda=H, rows_per_seg=M, mode=ops.IN_LRELU), 2.0*M*H*H),
 'fwd K=260 part': (lambda: ops.gemm_raw(x260, w516[:260], y, M, H, 260, mode=ops.IN_LRELU), 2.0*M*H*260),
}
for name, (fn, fl) in cases.items():
    us = timed(fn)
    print('{:<40} {:7.1f} us  {:6.1f} TF'.format(name, us, fl / us / 1e6))

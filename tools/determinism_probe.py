import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops
torch.manual_seed(0)
def rep(name, fn, n=20):
    ref = [t.clone() for t in fn() if t is not None]
    worst = 0.0
    for _ in range(n):
        out = [t for t in fn() if t is not None]
        for a, b in zip(out, ref):
            worst = max(worst, float((a - b).abs().max() / (b.abs().max() + 1e-30)))
    print('{:<40} max rel diff over {} repeats: {:.3e}'.format(name, n, worst))
B, T, F = 3, 50, 65
for (ci, co) in ((4, 1), (4, 4), (1, 4)):
    x = torch.randn(B, T, F, ci, device='cuda'); dy = torch.randn(B, T, F, co, device='cuda')
    w = torch.randn(5, 5, ci, co, device='cuda') * 0.2
    sc = torch.rand(ci, device='cuda') + 0.5; sh = torch.randn(ci, device='cuda') * 0.1
    rep('conv2d_bwd<%d,%d> affine dx,dw,aff' % (ci, co), lambda: ops._conv2d_bwd_raw(dy, x, w, sc, sh, None, ops.IN_LRELU, 0.3, 1, 0, True, True, True, True))
    rep('conv2d_bwd<%d,%d> lrelu dx,dw' % (ci, co), lambda: ops._conv2d_bwd_raw(dy, x, w, None, None, None, ops.IN_LRELU, 0.3, 1, 0, True, True, True, False))
    rep('conv2d_fwd<%d,%d> affine' % (ci, co), lambda: [ops._conv2d_fwd_raw(x, w, None, sc, sh, None, ops.IN_LRELU, 0.3, 1, 0)])
z = torch.randn(B * T * F, 4, device='cuda')
rep('colsums C=4', lambda: [ops.colsums(z)])
z2 = torch.randn(B * T, 32, device='cuda')
rep('colsums C=32', lambda: [ops.colsums(z2)])
a = torch.randn(B * T, 32, device='cuda'); wt = torch.randn(32, 65, device='cuda')
def g1():
    c = torch.empty(B * T, 65, device='cuda'); ops.gemm_raw(a, wt, c, B * T, 65, 32); return [c]
rep('gemm 150x65x32', g1)
def g2():
    c = torch.empty(32, 65, device='cuda'); d = torch.randn(B * T, 65, device='cuda', generator=None) * 0 + 1
    ops.gemm_raw(a, d, c, 32, 65, B * T, transA=1, lda=32, rows_per_seg=B * T); return [c]
rep('gemm dW 32x65x150', g2)
xl = torch.randn(B, T, 32, device='cuda'); W = torch.randn(32, 256, device='cuda') * 0.2; U = torch.randn(2, 32, 128, device='cuda') * 0.2; bb = torch.randn(256, device='cuda') * 0.1
def l1():
    xx = xl.clone().requires_grad_(True); Wp = W.clone().requires_grad_(True); Up = U.clone().requires_grad_(True)
    h = ops.lstm(xx, Wp, Up, bb); h.backward(torch.ones_like(h)); return [h.detach(), xx.grad, Wp.grad, Up.grad]
rep('blstm fwd+bwd H=32', l1)

# alternating weights (the transposed copy in the workspace head changes between launches)
x = torch.randn(B, T, F, 4, device='cuda'); dy = torch.randn(B, T, F, 4, device='cuda')
ws_ = [torch.randn(5, 5, 4, 4, device='cuda') * 0.2 for _ in range(4)]
refs = []
for w_ in ws_:
    torch.cuda.synchronize()
    refs.append(ops._conv2d_bwd_raw(dy, x, w_, None, None, None, ops.IN_LRELU, 0.3, 1, 0, True, False, False, False)[0].clone())
    torch.cuda.synchronize()
worst = 0.0
for it in range(50):
    outs = [ops._conv2d_bwd_raw(dy, x, w_, None, None, None, ops.IN_LRELU, 0.3, 1, 0, True, False, False, False)[0] for w_ in ws_]
    for o, r in zip(outs, refs):
        worst = max(worst, float((o - r).abs().max() / r.abs().max()))
print('conv2d_bwd dx with alternating weights, back-to-back: max rel diff', worst)

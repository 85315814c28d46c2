"""Where do the gradient maps of the chain kernels differ from the step-by-step restatement?  (debug aid)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import test_chain_gpu as tc
from oracle import percival_oracle as O
from percivaltts_amd import ops, _hip

B, T, F, L = 1, 100, 65, 8
ws, bs = tc._weights(L, 1, 5)
g = torch.Generator().manual_seed(11)
x0 = torch.randn(B, T, F, generator=g, dtype=torch.float64)
R = torch.randn(B, T, F, 4, generator=g, dtype=torch.float64)
wq = [tc._bf(w) for w in ws]
a = [tc._bf(x0).unsqueeze(-1)]
for l in range(L):
    a.append(tc._bf(O.lrelu(O.conv2d_nhwc(a[-1], wq[l], bs[l]))))
mask = lambda t: torch.where(t > 0, torch.ones_like(t), torch.full_like(t, 0.3))
d = [None] * (L + 1)
v = tc._bf(R) * mask(a[L])
for l in range(L, 0, -1):
    d[l] = tc._bf(v)
    back = tc._convT(d[l], wq[l - 1], a[l - 1])
    if l > 1: v = back * mask(a[l - 1])

wd = [w.float().cuda() for w in ws]; bd = [b.float().cuda() for b in bs]
xd = x0.float().cuda()
tab = ops._C2C.table(wd, bd)
FP = (F + 1) & ~1
maps = torch.zeros((L - 1, B, T, FP, 4), dtype=torch.bfloat16, device='cuda')
gm = torch.zeros((L, B, T, FP, 4), dtype=torch.bfloat16, device='cuda')
al = torch.zeros((B, T, F, 4), dtype=torch.bfloat16, device='cuda')
g0 = torch.zeros((B, T, F), device='cuda')
P, st = _hip.ptr, _hip.stream
_hip.call('ptts_conv2d_chain_fwd', P(xd), xd.stride(1), P(tab), P(maps), P(al), B, T, F, L, 0.3, st())
# feed the kernels the oracle's own maps so that only the backward chain is compared
for l in range(1, L):
    maps[l - 1, :, :, :F] = a[l].to(torch.bfloat16).cuda()
al.copy_(a[L].to(torch.bfloat16).cuda())
dl = tc._bf(R).float().cuda()
_hip.call('ptts_conv2d_chain_bwd_data', P(dl), 0, P(maps), P(al), P(tab), P(gm), P(g0), B, T, F, L, 0.3, st())
torch.cuda.synchronize()
for l in range(L, 0, -1):
    got = gm[l - 1, :, :, :F].double().cpu()
    diff = (got - d[l]).abs()
    rel = float((diff ** 2).sum() ** 0.5 / (d[l] ** 2).sum() ** 0.5)
    big = diff > 0.05 * d[l].abs().max()
    idx = big.nonzero()
    print('gamma_{}: rel {:.2e}, values that differ {} of {}, big {}'.format(l, rel, int((diff > 0).sum()), diff.numel(), int(big.sum())))
    if len(idx):
        ts = sorted(set(int(i[1]) for i in idx)); fs = sorted(set(int(i[2]) for i in idx))
        print('   rows', ts[:40], ' bins', fs[:40])
print('g0 rel', tc._rel(g0.double().cpu(), back[..., 0]))

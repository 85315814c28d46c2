"""Context-Conv1D forward as a bf16x6 split product (split.hip) against the fp32 MFMA kernel: error vs fp64 and time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops
from percivaltts_amd.ops import call, ptr, stream

B, T, Cin, N, KW = int(os.environ.get("BATCH", 64)), int(os.environ.get("FRAMES", 400)), 601, 256, 21
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = 'cuda'
torch.manual_seed(0)
x = torch.randn(B, T, Cin, device=dev)
w = torch.randn(KW, Cin, N, device=dev) * 0.02
bias = torch.randn(N, device=dev)
pl = (KW - 1) // 2
Cp = (Cin + 31) // 32 * 32
Tp = T + KW - 1

ap = ops._pad_time(x, pl, KW - 1 - pl)
y32 = torch.empty(B, T, N, device=dev)
def f32():
    ops.gemm_raw(ap, w, y32, B * T, N, KW * Cin, lda=Cin, rows_per_seg=T, seg_stride=Tp * Cin, bias=bias)

xa = torch.empty(3, Cp // 32, B, Tp, 32, dtype=torch.bfloat16, device=dev)
wb = torch.empty(3, Cp // 32, N, KW, 32, dtype=torch.bfloat16, device=dev)
y6 = torch.empty(B, T, N, device=dev)
def split_x():
    call('ptts_split3_frames', ptr(x), ptr(xa[0]), ptr(xa[1]), ptr(xa[2]), B, T, Cin, pl, KW - 1 - pl, Cp, stream())
def split_w():
    call('ptts_split3_weight_t', ptr(w), ptr(wb[0]), ptr(wb[1]), ptr(wb[2]), KW, Cin, N, Cp, stream())
def x6():
    call('ptts_conv1d_bf16x6', ptr(xa[0]), ptr(xa[1]), ptr(xa[2]), ptr(wb[0]), ptr(wb[1]), ptr(wb[2]), ptr(bias), ptr(y6),
         B, T, KW, Cp, N, stream())

split_x(); split_w(); torch.cuda.synchronize()
# the planes add up to the operand (to 2^-24 relative) and are laid out as documented
xs = xa.float().sum(0).permute(1, 2, 0, 3).reshape(B, Tp, Cp)
print('planes(x): max |sum - x| / |x|max = %.3e' % ((xs[:, pl:pl + T, :Cin] - x).abs().max() / x.abs().max()).item(),
      ' pad rows/channels zero:', bool((xs[:, :pl] == 0).all() and (xs[:, pl + T:] == 0).all() and (xs[..., Cin:] == 0).all()))
ws = wb.float().sum(0).permute(1, 2, 0, 3).reshape(N, KW, Cp)
print('planes(w): max |sum - w| / |w|max = %.3e' % ((ws[..., :Cin].permute(1, 2, 0) - w).abs().max() / w.abs().max()).item())
f32(); x6(); torch.cuda.synchronize()
# fp64 reference on a sample of frames
gen = torch.Generator().manual_seed(1)
bs = torch.randint(0, B, (64,), generator=gen); ts = torch.randint(0, T, (64,), generator=gen)
bs[:4] = torch.tensor([0, 0, B - 1, B - 1]); ts[:4] = torch.tensor([0, T - 1, 0, T - 1])
apd = ap.double(); wd = w.double().reshape(KW * Cin, N)
ref = torch.stack([apd[b, t:t + KW].reshape(-1) @ wd for b, t in zip(bs.tolist(), ts.tolist())]) + bias.double()
sc = ref.abs().mean()
e32 = (y32[bs, ts].double() - ref).abs().max() / sc
e6 = (y6[bs, ts].double() - ref).abs().max() / sc
print('max err / mean|ref|: fp32 MFMA %.3e   bf16x6 %.3e   (y6 vs y32 everywhere: %.3e)' % (e32.item(), e6.item(), ((y6 - y32).abs().max() / sc).item()))

# weight gradient
dy = torch.randn(B, T, N, device=dev)
dw32 = torch.empty_like(w); dw6 = torch.empty_like(w)
def bww32():
    ops.gemm_raw(ap, dy, dw32, KW * Cin, N, B * T, transA=1, lda=Cin, rows_per_seg=T, seg_stride=Tp * Cin)
Pp = ops._C1Split.plane_len(B, T, KW)
def split_xt():
    global xt, Crows
    xt, Crows = ops._C1Split.transposed(ap, B, Tp, Cin, 0, Tp, Pp)
def split_yt():
    global yt
    yt, _ = ops._C1Split.transposed(dy, B, T, N, 0, Tp, Pp)
def bww6():
    call('ptts_conv1d_wgrad_bf16x6', ptr(xt[0]), ptr(xt[1]), ptr(xt[2]), ptr(yt[0]), ptr(yt[1]), ptr(yt[2]), ptr(dw6),
         B, T, KW, Cin, N, Crows, Pp, stream())
dwt = torch.empty_like(w); dbt = torch.empty(N, device=dev)
def tr_x():
    global xtf, Crt
    xtf, Crt, _ = ops._C1WgradT.frames_t(None, ap, KW)
def tr_y():
    global ytf
    ytf = torch.empty((N, Pp), device=dev)
    call('ptts_transpose_frames', ptr(dy), ptr(ytf), B, T, N, 0, Tp, N, Pp, stream())
def bwwt():
    call('ptts_conv1d_wgrad_t', ptr(xtf), ptr(ytf), ptr(dwt), ptr(dbt), B, T, KW, Cin, N, Crt, Pp, stream())
tr_x(); tr_y(); bwwt()
split_xt(); split_yt(); bww32(); bww6(); torch.cuda.synchronize()
print('fp32 frame-major wgrad: dw vs stream-K fp32 everywhere %.3e (of mean |dw|), db rel err %.3e' % (
    ((dwt - dw32).abs().max() / dw32.abs().mean()).item(), ((dbt - dy.sum((0, 1))).abs().max() / dy.sum((0, 1)).abs().mean()).item()))
xts = xt.float().sum(0)[:Cin, :B * Tp].reshape(Cin, B, Tp).permute(1, 2, 0)
print('planes(xt): max |sum - xp| = %.3e' % (xts - ap).abs().max().item(), ' slack zero:', bool((xt[:, :, B * Tp:] == 0).all() and (xt[:, Cin:] == 0).all()))
js = [0, 1, 10, 19, 20]; cs = [0, 1, 300, 599, 600]
refw = torch.stack([torch.stack([(apd[:, j:j + T, c].reshape(-1, 1) * dy.double().reshape(B * T, N)).sum(0) for c in cs]) for j in js])
scw = refw.abs().mean()
jj = torch.tensor(js, device=dev)[:, None]; cc = torch.tensor(cs, device=dev)[None, :]
print('dW max err / mean|ref|: fp32 MFMA %.3e   bf16x6 %.3e   (dw6 vs dw32 everywhere: %.3e)' % (
    ((dw32[jj, cc].double() - refw).abs().max() / scw).item(), ((dw6[jj, cc].double() - refw).abs().max() / scw).item(),
    ((dw6 - dw32).abs().max() / scw).item()))
print('dW max err / mean|ref|: fp32 frame-major %.3e' % ((dwt[jj, cc].double() - refw).abs().max() / scw).item())

def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
flop = 2.0 * B * T * N * KW * Cin
for name, fn in (('fp32 mfma', f32), ('bf16x6', x6), ('split x', split_x), ('split w', split_w), ('bww fp32', bww32), ('bww x6', bww6), ('split xt', split_xt), ('split yt', split_yt), ('bww f32 T', bwwt), ('transp x', tr_x), ('transp y', tr_y)):
    ms = timeit(fn)
    print('%-10s %.3f ms' % (name, ms) + ('  %.1f TF (algorithmic)' % (flop / ms / 1e9) if fn in (f32, x6, bww32, bww6, bwwt) else ''))

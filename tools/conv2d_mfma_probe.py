"""Probe of csrc/conv2d_mfma.hip at the critic's layer shape [64,400,65,4]: each entry point against the packed-FMA kernels of
conv2d.hip (which the test-suite pins against the fp64 oracle) and against the fp64 oracle on crops, then HIP-event timings.
usage: python tools/conv2d_mfma_probe.py [B T F]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops, _hip
from percivaltts_amd._hip import call, ptr, stream
from oracle import percival_oracle as O

B, T, F = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 400, 65)
dil = int(os.environ.get('DIL', '1'))
g = torch.Generator().manual_seed(1)
x = torch.randn(B, T, F, 4, generator=g).cuda()
msk = torch.randn(B, T, F, 4, generator=g).cuda()
dy = torch.randn(B, T, F, 4, generator=g).cuda()
w = (torch.randn(5, 5, 4, 4, generator=g) * 0.2).cuda()
b = torch.randn(4, generator=g).cuda()
lib = _hip.lib()
nb = lib.ptts_conv2d_mfma_table_bytes(5)
tf = torch.empty(nb, dtype=torch.uint8, device='cuda'); tb = torch.empty(nb, dtype=torch.uint8, device='cuda')
call('ptts_conv2d_mfma_tables', ptr(w), ptr(tf), ptr(tb), 5, 5, 4, 4, 3, stream())
pad = 2 * dil

if os.environ.get('C2M_ONLY'):
    # counter runs (tools/c2m_pmc.sh): only the new kernels, a few launches each
    nws = lib.ptts_conv2d_mfma_wgrad_workspace_bytes(B, T)
    ws = torch.empty(nws, dtype=torch.uint8, device='cuda')
    y = torch.empty_like(x)
    nblocks = ctypes.c_int(0); npart = ctypes.c_int(0)
    for _ in range(int(os.environ.get('C2M_REPS', '10'))):
        call('ptts_conv2d_mfma_fwd', ptr(x), ptr(tf), ptr(b), None, None, None, None, ptr(y), B, T, F, 5, dil, pad, ops.IN_LRELU, 0.3, 3, 0, 0, stream())
        call('ptts_conv2d_mfma_fwd', ptr(dy), ptr(tb), None, None, None, None, ptr(x), ptr(y), B, T, F, 5, dil, 4 * dil - pad, ops.IN_NONE, 0.3, 3, 0, 0, stream())
        call('ptts_conv2d_mfma_wgrad_partials', ptr(dy), ptr(x), None, ptr(ws), ws.numel(), ctypes.byref(nblocks), ctypes.byref(npart),
             B, T, F, 5, dil, pad, ops.IN_LRELU, 0.3, 3, 0, 0, stream())
        if dil == 1:
            # round 4: the fused backward launches (c2m::bwd_ws_kernel<1 / 2, 3>) and the masked forward they replace half of
            call('ptts_conv2d_mfma_bwd_fused', ptr(dy), ptr(x), None, ptr(tb), ptr(y), ptr(ws), ws.numel(), ctypes.byref(nblocks), ctypes.byref(npart),
                 B, T, F, 5, 2, 1, 0.3, stream())
            call('ptts_conv2d_mfma_bwd_fused', ptr(msk), ptr(dy), ptr(x), ptr(tf), ptr(y), ptr(ws), ws.numel(), ctypes.byref(nblocks), ctypes.byref(npart),
                 B, T, F, 5, 2, 2, 0.3, stream())
            call('ptts_conv2d_mfma_fwd', ptr(msk), ptr(tf), None, None, None, ptr(x), None, ptr(y), B, T, F, 5, dil, pad, ops.IN_MASKMUL, 0.3, 3, 0, 0, stream())
    torch.cuda.synchronize()
    sys.exit(0)

def rel(a, ref):
    return float((a.double() - ref.double()).abs().max() / ref.double().abs().mean())

def fwd_new(xx, table, bias, mask_src, out_mask, mode, pad_t):
    y = torch.empty_like(xx)
    call('ptts_conv2d_mfma_fwd', ptr(xx), ptr(table), ptr(bias), None, None, ptr(mask_src), ptr(out_mask), ptr(y),
         B, T, F, 5, dil, pad_t, mode, 0.3, 3, 0, 0, stream())
    return y

def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

# ---- forward, LeakyReLU on load
y_old = ops._conv2d_fwd_raw(x, w, b, None, None, None, ops.IN_LRELU, 0.3, dil, ops.PAD_SAME)
y_new = fwd_new(x, tf, b, None, None, ops.IN_LRELU, pad)
print('fwd lrelu    : new vs old {:.3e}'.format(rel(y_new, y_old)))
for (bi, t0, n) in ((0, 0, 9), (B // 2, T // 2 - 15, 30), (B - 1, T - 7, 7)):
    lo, hi = max(0, t0 - 2 * dil), min(T, t0 + n + 2 * dil)
    ref = O.conv2d_nhwc(O.lrelu(x[bi:bi + 1, lo:hi].double().cpu()), w.double().cpu(), b.double().cpu(), dil_t=dil)
    r = ref[:, t0 - lo:t0 - lo + n]
    print('   crop b={} t0={}: new {:.3e}  old {:.3e} (max err / mean |ref|, fp64 oracle)'.format(
        bi, t0, rel(y_new[bi:bi + 1, t0:t0 + n].cpu(), r), rel(y_old[bi:bi + 1, t0:t0 + n].cpu(), r)))
# ---- masked forward (second-order sweep)
y_old = ops._conv2d_fwd_raw(x, w, None, None, None, msk, ops.IN_MASKMUL, 0.3, dil, ops.PAD_SAME)
y_new = fwd_new(x, tf, None, msk, None, ops.IN_MASKMUL, pad)
print('fwd maskmul  : new vs old {:.3e}'.format(rel(y_new, y_old)))
# ---- backward data with the LeakyReLU mask of the layer input
dx_old, _, _, _, _ = ops._conv2d_bwd_raw(dy, x, w, None, None, None, ops.IN_LRELU, 0.3, dil, ops.PAD_SAME, True, False, False, False)
dx_new = fwd_new(dy, tb, None, None, x, ops.IN_NONE, 4 * dil - pad)
print('bwd data     : new vs old {:.3e}'.format(rel(dx_new, dx_old)))
# ---- weight gradient
_, dw_old, db_old, _, _ = ops._conv2d_bwd_raw(dy, x, w, None, None, None, ops.IN_LRELU, 0.3, dil, ops.PAD_SAME, False, True, True, False)
def wgrad_new(mode, mask_src):
    nws = lib.ptts_conv2d_mfma_wgrad_workspace_bytes(B, T)
    ws = torch.empty(nws, dtype=torch.uint8, device='cuda')
    nblocks = ctypes.c_int(0); npart = ctypes.c_int(0)
    call('ptts_conv2d_mfma_wgrad_partials', ptr(dy), ptr(x), ptr(mask_src), ptr(ws), ws.numel(), ctypes.byref(nblocks), ctypes.byref(npart),
         B, T, F, 5, dil, pad, mode, 0.3, 3, 0, 0, stream())
    return ws, nblocks.value, npart.value
ws, nblocks, npart = wgrad_new(ops.IN_LRELU, None)
parts = ws[4096:].view(torch.float32)[:nblocks * npart].view(nblocks, npart)
dw_new = parts[:, :400].double().sum(0).view(5, 5, 4, 4); db_new = parts[:, 400:404].double().sum(0)
print('weight grad  : new vs old dW {:.3e}  db {:.3e}'.format(rel(dw_new, dw_old), rel(db_new, db_old)))
dw64 = torch.zeros(5, 5, 4, 4, dtype=torch.float64)
for b0 in range(0, B, 8):
    wq = w.double().cpu().requires_grad_(True)
    O.conv2d_nhwc(O.lrelu(x[b0:b0 + 8].double().cpu()), wq, None, dil_t=dil).backward(dy[b0:b0 + 8].double().cpu())
    dw64 += wq.grad
print('   fp64 oracle: new {:.3e}  old {:.3e}'.format(rel(dw_new.cpu(), dw64), rel(dw_old.cpu(), dw64)))
_, dw_old2, _, _, _ = ops._conv2d_bwd_raw(dy, x, w, None, None, msk, ops.IN_MASKMUL, 0.3, dil, ops.PAD_SAME, False, True, False, False)
ws, nblocks, npart = wgrad_new(ops.IN_MASKMUL, msk)
parts = ws[4096:].view(torch.float32)[:nblocks * npart].view(nblocks, npart)
print('weight grad (maskmul): new vs old {:.3e}'.format(rel(parts[:, :400].double().sum(0).view(5, 5, 4, 4), dw_old2)))

# ---- timings
print('timings [us] at [{},{},{},4], dil {}:'.format(B, T, F, dil))
print('  fwd lrelu   old {:7.1f}   new {:7.1f}'.format(
    timeit(lambda: ops._conv2d_fwd_raw(x, w, b, None, None, None, ops.IN_LRELU, 0.3, dil, ops.PAD_SAME)),
    timeit(lambda: fwd_new(x, tf, b, None, None, ops.IN_LRELU, pad))))
print('  fwd maskmul old {:7.1f}   new {:7.1f}'.format(
    timeit(lambda: ops._conv2d_fwd_raw(x, w, None, None, None, msk, ops.IN_MASKMUL, 0.3, dil, ops.PAD_SAME)),
    timeit(lambda: fwd_new(x, tf, None, msk, None, ops.IN_MASKMUL, pad))))
print('  bwd data    old {:7.1f}   new {:7.1f}'.format(
    timeit(lambda: ops._conv2d_bwd_raw(dy, x, w, None, None, None, ops.IN_LRELU, 0.3, dil, ops.PAD_SAME, True, False, False, False)),
    timeit(lambda: fwd_new(dy, tb, None, None, x, ops.IN_NONE, 4 * dil - pad))))
print('  weight grad old {:7.1f}   new {:7.1f}'.format(
    timeit(lambda: ops._conv2d_bwd_raw(dy, x, w, None, None, None, ops.IN_LRELU, 0.3, dil, ops.PAD_SAME, False, True, True, False)),
    timeit(lambda: wgrad_new(ops.IN_LRELU, None))))
print('  fused bwd   old {:7.1f}'.format(
    timeit(lambda: ops._conv2d_bwd_raw(dy, x, w, None, None, None, ops.IN_LRELU, 0.3, dil, ops.PAD_SAME, True, True, True, False))))
print('  tables          {:7.1f}'.format(timeit(lambda: call('ptts_conv2d_mfma_tables', ptr(w), ptr(tf), ptr(tb), 5, 5, 4, 4, 3, stream()))))

# ---- work-list parameters: pieces of the first tile (fs) and of a left-over tile (ts)
print('work list sweep [us]  (fs = pieces of a workgroup\'s first tile, ts = pieces of a left-over tile):')
for fs, ts in ((1, 1), (1, 4), (2, 4), (1, 2), (1, 6)):
    lib.ptts_conv2d_mfma_debug((fs << 8) | (ts << 12), None)
    print('  fs {} ts {}: fwd {:6.1f}  maskmul {:6.1f}  dx {:6.1f}  wgrad {:6.1f}'.format(fs, ts,
        timeit(lambda: fwd_new(x, tf, b, None, None, ops.IN_LRELU, pad)),
        timeit(lambda: fwd_new(x, tf, None, msk, None, ops.IN_MASKMUL, pad)),
        timeit(lambda: fwd_new(dy, tb, None, None, x, ops.IN_NONE, 4 * dil - pad)),
        timeit(lambda: wgrad_new(ops.IN_LRELU, None))))
lib.ptts_conv2d_mfma_debug(0, None)
# ---- where the time goes: phase switches and per-workgroup stamps (s_memtime = shader clock / ... 100 MHz constant clock)
nblk = B * ((T + 15) // 16)
def phases(name, fn):
    out = []
    for flags, what in ((0, 'all'), (5, 'mfma+lds+barrier'), (5 | 64, 'mfma+lds'), (1, 'no stage'), (2, 'no mfma'), (4, 'no store'), (3, 'skeleton+store'), (7, 'skeleton')):
        lib.ptts_conv2d_mfma_debug(flags, None)
        out.append('{} {:.1f}'.format(what, timeit(fn)))
    lib.ptts_conv2d_mfma_debug(0, None)
    print('  {:12s} '.format(name) + ' | '.join(out))
    buf = torch.zeros(nblk * 8, dtype=torch.int64, device='cuda')
    lib.ptts_conv2d_mfma_debug(8, ctypes.c_void_p(buf.data_ptr()))
    fn(); torch.cuda.synchronize()
    lib.ptts_conv2d_mfma_debug(0, None)
    s = buf.view(nblk, 8); s = s[s[:, 4] != 0].double()
    STAMPS[name] = s.cpu().numpy()
    d = [(s[:, i + 1] - s[:, i]).median().item() for i in range(4)]
    span = (s[:, 4].max() - s[:, 0].min()).item()
    print('     stamps (s_memtime ticks, median over workgroups): stage {:.0f}  barrier {:.0f}  first mfma pass {:.0f}  rest {:.0f}   lifetime {:.0f}   | start->loads issued {:.0f}  ->table copied {:.0f}'.format(
        d[0], d[1], d[2], d[3], (s[:, 4] - s[:, 0]).median().item(), (s[:, 5] - s[:, 0]).median().item(), (s[:, 6] - s[:, 0]).median().item()))
STAMPS = {}
print('phase switches [us]:')
phases('fwd lrelu', lambda: fwd_new(x, tf, b, None, None, ops.IN_LRELU, pad))
phases('bwd data', lambda: fwd_new(dy, tb, None, None, x, ops.IN_NONE, 4 * dil - pad))
phases('weight grad', lambda: wgrad_new(ops.IN_LRELU, None))
import numpy as np
os.makedirs('gpurun_out', exist_ok=True)
np.savez('gpurun_out/c2m_stamps.npz', **{k.replace(' ', '_'): v for k, v in STAMPS.items()})

"""Phase switches of the wave-specialised Conv2D forward kernel (c2m::fwd_ws_kernel) at the shapes of the critic step: the pair
forward [3B,400,65,4] and the penalty's backward-data pass [B,400,65,4].   python tools/c2m_ws_phases.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops, _hip
from percivaltts_amd._hip import call, ptr, stream

lib = _hip.lib()
g = torch.Generator().manual_seed(1)
w = (torch.randn(5, 5, 4, 4, generator=g) * 0.2).cuda()
b = torch.randn(4, generator=g).cuda()
nb = lib.ptts_conv2d_mfma_table_bytes(5)
tf = torch.empty(nb, dtype=torch.uint8, device='cuda'); tb = torch.empty(nb, dtype=torch.uint8, device='cuda')
call('ptts_conv2d_mfma_tables', ptr(w), ptr(tf), ptr(tb), 5, 5, 4, 4, 3, stream())

def timeit(fn, n=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for B in (192, 64):
    T, F = 400, 65
    x = torch.randn(B, T, F, 4, generator=g).cuda(); y = torch.empty_like(x); m = torch.randn(B, T, F, 4, generator=g).cuda()
    fns = {'fwd lrelu': lambda: call('ptts_conv2d_mfma_fwd', ptr(x), ptr(tf), ptr(b), None, None, None, None, ptr(y), B, T, F, 5, 1, 2, ops.IN_LRELU, 0.3, 3, 0, 0, stream()),
           'bwd data ': lambda: call('ptts_conv2d_mfma_fwd', ptr(x), ptr(tb), None, None, None, None, ptr(m), ptr(y), B, T, F, 5, 1, 2, ops.IN_NONE, 0.3, 3, 0, 0, stream())}
    for name, fn in fns.items():
        out = []
        for flags, what in ((0, 'all'), (1, 'no stage'), (2, 'no mfma'), (4, 'no store'), (5, 'mfma only'), (6, 'stage only'), (3, 'store only'), (7, 'skeleton')):
            lib.ptts_conv2d_mfma_debug(flags, None)
            out.append('{} {:.1f}'.format(what, timeit(fn)))
        lib.ptts_conv2d_mfma_debug(0, None)
        _hip.clear_status()
        print('[{},{},{},4] {}: '.format(B, T, F, name) + ' | '.join(out))

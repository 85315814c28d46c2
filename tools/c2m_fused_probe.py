"""A/B of the fused backward launches of csrc/conv2d_mfma.hip (c2m::bwd_ws_kernel) against the separate launches: values (dx / cot_dy
bit-identical, dW against an fp64 chunked reduction) and time per layer at B = 64 / 128, [*,400,65,4]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops

def t_ms(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

g = torch.Generator().manual_seed(3)
for B in (64, 128):
    T, F = 400, 65
    x = torch.randn(B, T, F, 4, generator=g).cuda(); dy = torch.randn(B, T, F, 4, generator=g).cuda(); u = torch.randn(B, T, F, 4, generator=g).cuda()
    w = (torch.randn(5, 5, 4, 4, generator=g) * 0.2).cuda()
    res = {}
    for fused in (False, True):
        ops.conv2d_fused(fused)
        dx, dw, db, _, _ = ops._conv2d_bwd_raw(dy, x, w, None, None, None, ops.IN_LRELU, 0.3, 1, ops.PAD_SAME, True, True, True, False)
        torch.cuda.synchronize()
        t1 = t_ms(lambda: ops._conv2d_bwd_raw(dy, x, w, None, None, None, ops.IN_LRELU, 0.3, 1, ops.PAD_SAME, True, True, True, False))
        if fused:
            cot, buf, nb, npart = ops._conv2d_mfma_bwd_fused(2, u, dy, x, w, 0.3)
            t2 = t_ms(lambda: ops._conv2d_mfma_bwd_fused(2, u, dy, x, w, 0.3))
            # reduce the rows by hand
            rows = buf[4096:].view(torch.float32)[:nb * npart].view(nb, npart)
            dw2 = rows[:, :400].sum(0).view(5, 5, 4, 4)
        else:
            cot = ops._conv2d_fwd_raw(u, w, None, None, None, x, ops.IN_MASKMUL, 0.3, 1, ops.PAD_SAME)
            _, dw2, _, _, _ = ops._conv2d_bwd_raw(dy, u, w, None, None, x, ops.IN_MASKMUL, 0.3, 1, ops.PAD_SAME, False, True, False, False)
            t2 = t_ms(lambda: (ops._conv2d_fwd_raw(u, w, None, None, None, x, ops.IN_MASKMUL, 0.3, 1, ops.PAD_SAME),
                               ops._conv2d_bwd_raw(dy, u, w, None, None, x, ops.IN_MASKMUL, 0.3, 1, ops.PAD_SAME, False, True, False, False)))
        res[fused] = (dx, dw, db, cot, dw2.clone())
        print('B={} fused={}: first-order dx+dW+db {:.1f} us, second-order fwd+dW {:.1f} us (with the reduce launch)'.format(B, fused, t1, t2))
    a, b = res[False], res[True]
    print('   dx equal', torch.equal(a[0], b[0]), ' cot_dy equal', torch.equal(a[3], b[3]),
          ' dW rel', float((a[1] - b[1]).norm() / a[1].norm()), ' db rel', float((a[2] - b[2]).norm() / a[2].norm()),
          ' dW2 rel', float((a[4] - b[4]).norm() / a[4].norm()))
ops.conv2d_fused(None)
from percivaltts_amd import _hip
_hip.check_status(); print('status clear')

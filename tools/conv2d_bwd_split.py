"""conv2d backward of the critic's 4->4 5x5 layer at configs[1] size: time with dx only, dw only, both."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops, _hip
B, T, F, C = 64, 400, 65, 4
x = torch.randn(B, T, F, C, device='cuda'); dy = torch.randn(B, T, F, C, device='cuda')
w = torch.randn(5, 5, C, C, device='cuda') * 0.2
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    with _hip.KernelTimer() as kt:
        for _ in range(n): fn()
    d = sorted(t for (_, _, t) in kt.durations_ms()); return d[len(d) // 2] * 1e3
for name, (dx, dw) in (('dx only', (True, False)), ('dw only', (False, True)), ('dx + dw', (True, True))):
    us = timed(lambda: ops._conv2d_bwd_raw(dy, x, w, None, None, None, ops.IN_LRELU, 0.3, 1, ops.PAD_SAME, dx, dw, dw, False))
    print('{:<10} {:6.1f} us'.format(name, us))
us = timed(lambda: ops._conv2d_fwd_raw(x, w, None, None, None, None, ops.IN_LRELU, 0.3, 1, ops.PAD_SAME))
print('{:<10} {:6.1f} us'.format('forward', us))

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops, _hip
x = torch.randn(64, 400, 65, 4, device='cuda'); w = torch.randn(5,5,4,4, device='cuda')*0.1; b = torch.randn(4, device='cuda')
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    with _hip.KernelTimer() as kt:
        for _ in range(n): fn()
    d = [t for (_, _, t) in kt.durations_ms()]
    d.sort(); return d[len(d)//2] * 1e3
for name, bits in (('full', 0), ('no compute', 1), ('no compute, no store', 5), ('no compute/store/prefetch-next', 7), ('nothing but first prefetch+barriers', 15), ('no prefetch-next', 2), ('no store', 4)):
    mode = 1 | (bits << 8)
    print('{:<38} {:7.1f} us'.format(name, timed(lambda: ops._conv2d_fwd_raw(x, w, b, None, None, None, mode, 0.3, 1, 0))))

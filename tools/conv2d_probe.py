"""Micro-probe of the critic's 5x5 conv2d kernels at the config-2 shape for rocprofv3 runs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops

B, T, F, C = 64, 400, 65, 4
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = 'cuda'
x = torch.randn(B, T, F, C, device=dev)
x1 = torch.randn(B, T, F, 1, device=dev)
w = torch.randn(5, 5, C, C, device=dev) * 0.1
w1 = torch.randn(5, 5, 1, C, device=dev) * 0.1
b = torch.randn(C, device=dev)
dy = torch.randn(B, T, F, C, device=dev)
cases = {
    'fwd44_lrelu': lambda: ops._conv2d_fwd_raw(x, w, b, None, None, None, ops.IN_LRELU, 0.3, 1, 0),
    'fwd14_none': lambda: ops._conv2d_fwd_raw(x1, w1, b, None, None, None, ops.IN_NONE, 0.3, 1, 0),
    'fwd44_maskmul': lambda: ops._conv2d_fwd_raw(x, w, None, None, None, dy, ops.IN_MASKMUL, 0.3, 1, 0),
    'bwd44_dx_dw': lambda: ops._conv2d_bwd_raw(dy, x, w, None, None, None, ops.IN_LRELU, 0.3, 1, 0, True, True, True, False),
    'bwd44_dx': lambda: ops._conv2d_bwd_raw(dy, x, w, None, None, None, ops.IN_LRELU, 0.3, 1, 0, True, False, False, False),
    'bwd44_dw_maskmul': lambda: ops._conv2d_bwd_raw(dy, x, w, None, None, dy, ops.IN_MASKMUL, 0.3, 1, 0, False, True, False, False),
}
px = B * T * F
for name, fn in cases.items():
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    dt = (time.time() - t) / reps
    print('{:<18} {:8.1f} us'.format(name, dt * 1e6))

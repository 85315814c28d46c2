import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops, _hip
def t_ms(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
g = torch.Generator().manual_seed(3)
lib = _hip.lib()
for B in (64, 128):
    T, F = 400, 65
    x = torch.randn(B, T, F, 4, generator=g).cuda(); dy = torch.randn(B, T, F, 4, generator=g).cuda()
    w = (torch.randn(5, 5, 4, 4, generator=g) * 0.2).cuda()
    out = []
    for ts in (0, 1, 2):
        lib.ptts_conv2d_mfma_debug(ts << 12, None)
        out.append((ts, round(t_ms(lambda: ops._conv2d_mfma_bwd_fused(1, dy, x, None, w, 0.3)), 1), round(t_ms(lambda: ops._conv2d_mfma_bwd_fused(2, dy, x, x, w, 0.3)), 1)))
    lib.ptts_conv2d_mfma_debug(0, None)
    print(B, 'tiles', B * 25, out, flush=True)
# phase switches of c2m::bwd_ws_kernel at B = 128: 1 no staging, 2 no MFMA phase at all, 16 no convolution, 32 no weight gradient, 4 no stores
B, T, F = 128, 400, 65
x = torch.randn(B, T, F, 4, generator=g).cuda(); dy = torch.randn(B, T, F, 4, generator=g).cuda()
for flags, name in ((0, 'full'), (1, 'no staging'), (2, 'staging only'), (16, 'no convolution'), (32, 'no weight gradient'), (4, 'no stores'), (1 | 4, 'no staging, no stores'), (1 | 16, 'wgrad MFMA only'), (1 | 32, 'conv MFMA only')):
    lib.ptts_conv2d_mfma_debug(flags, None)
    print('%-24s kind1 %.1f us  kind2 %.1f us' % (name, t_ms(lambda: ops._conv2d_mfma_bwd_fused(1, dy, x, None, w, 0.3)), t_ms(lambda: ops._conv2d_mfma_bwd_fused(2, dy, x, x, w, 0.3))), flush=True)
lib.ptts_conv2d_mfma_debug(0, None)
_hip.check_status()

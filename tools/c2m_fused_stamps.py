"""Timeline of one steady-state piece (the fourth) of the fused Conv2D backward kernel (c2m::bwd_ws_kernel) from s_memtime stamps of
the first multiplying and the first staging wave of every workgroup -- needs the library built with -DC2M_PROBE_STAMPS=1 (PTTS_LIB_PATH)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from percivaltts_amd import ops, _hip

lib = _hip.lib()
if not hasattr(lib, 'ptts_conv2d_mfma_probe_stamps'):
    print('library not built with C2M_PROBE_STAMPS=1'); sys.exit(0)
lib.ptts_conv2d_mfma_probe_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.ptts_conv2d_mfma_probe_stamps.restype = ctypes.c_int
g = torch.Generator().manual_seed(3)
w = (torch.randn(5, 5, 4, 4, generator=g) * 0.2).cuda()
for kind, B in ((1, 128), (2, 64)):
    x = torch.randn(B, 400, 65, 4, generator=g).cuda(); dy = torch.randn(B, 400, 65, 4, generator=g).cuda()
    fn = (lambda: ops._conv2d_mfma_bwd_fused(1, dy, x, None, w, 0.3)) if kind == 1 else (lambda: ops._conv2d_mfma_bwd_fused(2, dy, x, x, w, 0.3))
    for flags, what in ((0, 'all'), (4, 'no store'), (1, 'no stage')):
        lib.ptts_conv2d_mfma_debug(flags, None)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        assert lib.ptts_conv2d_mfma_probe_stamps(None, 1) == 0
        fn(); torch.cuda.synchronize()
        host = np.zeros(256 * 16, dtype=np.uint64)
        assert lib.ptts_conv2d_mfma_probe_stamps(host.ctypes.data_as(ctypes.c_void_p), 0) == 0
        lib.ptts_conv2d_mfma_debug(0, None)
        _hip.clear_status()
        s = host.reshape(256, 16).astype(np.float64)
        s = s[s[:, 0] != 0]
        d = lambda a, b_: float((s[:, b_] - s[:, a]).mean())
        print('kind {} B={} {:9s} multiplying wave [ticks]: wait {:6.0f}  convolution {:6.0f}  weight gradient {:6.0f}  signal {:5.0f}  walk {:5.0f} | staging wave: wait {:6.0f}  commit {:6.0f}  signal+walk {:5.0f}  loads issued {:6.0f}'.format(
            kind, B, what, d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), d(8, 9), d(9, 10), d(10, 11), d(11, 12)))

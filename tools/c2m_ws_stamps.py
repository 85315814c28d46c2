"""Timeline of one steady-state piece (the fourth) of the wave-specialised Conv2D forward kernel, from s_memtime stamps (100 MHz) of
the first multiplying and the first staging wave of every workgroup -- needs the library built with -DC2M_PROBE_STAMPS=1:
    bash tools/ab_file.sh conv2d_mfma tools/c2m_ws_stamps.py C2M_PROBE_STAMPS=1      (the "as built" legs print zeros)"""
import sys, os, ctypes
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
B, T, F = 192, 400, 65
x = torch.randn(B, T, F, 4, generator=g).cuda(); y = torch.empty_like(x); m = torch.randn(B, T, F, 4, generator=g).cuda()
fns = {'fwd lrelu': lambda: call('ptts_conv2d_mfma_fwd', ptr(x), ptr(tf), ptr(b), None, None, None, None, ptr(y), B, T, F, 5, 1, 2, ops.IN_LRELU, 0.3, 3, 0, 0, stream()),
       'bwd data ': lambda: call('ptts_conv2d_mfma_fwd', ptr(x), ptr(tb), None, None, None, None, ptr(m), ptr(y), B, T, F, 5, 1, 2, ops.IN_NONE, 0.3, 3, 0, 0, stream())}
for name, fn in fns.items():
    for flags, what in ((0, 'all'), (4, 'no store'), (1, 'no stage'), (5, 'mfma only'), (7, 'skeleton')):
        buf = torch.zeros(4096 * 16, dtype=torch.int64, device='cuda')      # (>= 16 words for every workgroup of the launch: at most 256)
        lib.ptts_conv2d_mfma_debug(flags, ctypes.c_void_p(buf.data_ptr()))
        for _ in range(3): fn()
        torch.cuda.synchronize(); buf.zero_()
        fn(); torch.cuda.synchronize()
        lib.ptts_conv2d_mfma_debug(0, None)
        _hip.clear_status()
        s = buf.view(4096, 16)[:256].double()
        ok = s[:, 0] != 0
        s = s[ok]
        if s.numel() == 0:
            print(name, what, 'no stamps (library not built with C2M_PROBE_STAMPS=1)'); continue
        d = lambda a, b_: float((s[:, b_] - s[:, a]).mean()) * 10.0       # ns
        print('{} {:9s} multiplying wave [ns]: bookkeeping+wait {:6.0f}  piece {:6.0f}  signal {:5.0f}   | staging wave: wait {:6.0f}  commit {:6.0f}  signal {:5.0f}  loads issued {:5.0f}   | mult starts after stager\'s top {:6.0f}'.format(
            name, what, d(0, 1), d(1, 2), d(2, 3), d(8, 9), d(9, 10), d(10, 11), d(11, 12), d(8, 0)))

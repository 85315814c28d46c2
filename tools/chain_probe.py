"""Times of the fused Conv2D-stack kernels (csrc/conv2d_chain.hip) at BASELINE size: per launch, HIP events, back to back."""
import ctypes, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops, _hip

def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

def main():
    T, F, L = 400, 65, 8
    g = torch.Generator().manual_seed(1)
    ws = [(torch.rand(5, 5, 1 if l == 0 else 4, 4, generator=g) - 0.5).mul(0.3).cuda() for l in range(L)]
    bs = [torch.zeros(4).cuda() for _ in range(L)]
    tab = ops._C2C.table(ws, bs)
    FP = (F + 1) & ~1
    out = {}
    if os.environ.get('CHAIN_PROBE_FWD') == '1':     # the forward kernel alone (PTTS_CHAIN_DBG phase switches)
        B = 128
        x0 = torch.randn(B, T, F, generator=g).cuda()
        maps = torch.empty((L - 1, B, T, FP, 4), dtype=torch.bfloat16, device='cuda'); a_last = torch.empty((B, T, F, 4), dtype=torch.bfloat16, device='cuda')
        fwd = lambda: _hip.call('ptts_conv2d_chain_fwd', _hip.ptr(x0), x0.stride(1), _hip.ptr(tab), _hip.ptr(maps), _hip.ptr(a_last), B, T, F, L, 0.3, _hip.stream())
        print('fwdonly B=128 us:', round(timeit(fwd), 1))
        return
    short = os.environ.get('CHAIN_PROBE_SHORT') == '1'     # one batch size, few repetitions: the run under rocprofv3 --pmc
    for B in ((128,) if short else (64, 128, 192)):
        x0 = torch.randn(B, T, F, generator=g).cuda()
        maps = torch.empty((L - 1, B, T, FP, 4), dtype=torch.bfloat16, device='cuda')
        gmaps = torch.empty((L, B, T, FP, 4), dtype=torch.bfloat16, device='cuda')
        a_last = torch.empty((B, T, F, 4), dtype=torch.bfloat16, device='cuda')
        d_last = torch.randn(B, T, F, 4, generator=g).cuda()
        d16 = d_last.to(torch.bfloat16)
        g0 = torch.empty((B, T, F), device='cuda'); u0 = torch.randn(B, T, F, generator=g).cuda()
        o2 = torch.empty((B, T, F, 4), device='cuda')
        parts = torch.empty(_hip.lib().ptts_conv2d_chain_partials_bytes(L), dtype=torch.uint8, device='cuda')
        nb, npart = ctypes.c_int(0), ctypes.c_int(0)
        st = _hip.stream
        P = _hip.ptr
        fwd = lambda: _hip.call('ptts_conv2d_chain_fwd', P(x0), x0.stride(1), P(tab), P(maps), P(a_last), B, T, F, L, 0.3, st())
        bwd = lambda: _hip.call('ptts_conv2d_chain_bwd', P(d_last), 0, P(x0), x0.stride(1), P(maps), P(a_last), P(tab), P(parts), parts.numel(), ctypes.byref(nb), ctypes.byref(npart), B, T, F, L, 1, 0.3, st())
        bwd16 = lambda: _hip.call('ptts_conv2d_chain_bwd', P(d16), 1, P(x0), x0.stride(1), P(maps), P(a_last), P(tab), P(parts), parts.numel(), ctypes.byref(nb), ctypes.byref(npart), B, T, F, L, 1, 0.3, st())
        dat = lambda: _hip.call('ptts_conv2d_chain_bwd_data', P(d_last), 0, P(maps), P(a_last), P(tab), P(gmaps), P(g0), B, T, F, L, 0.3, st())
        dat0 = lambda: _hip.call('ptts_conv2d_chain_bwd_data', P(d_last), 0, P(maps), P(a_last), P(tab), None, P(g0), B, T, F, L, 0.3, st())
        sec = lambda: _hip.call('ptts_conv2d_chain_second', P(u0), P(gmaps), P(maps), P(a_last), P(tab), P(o2), 0, P(parts), parts.numel(), ctypes.byref(nb), ctypes.byref(npart), B, T, F, L, 1, 0.3, st())
        fwd(); dat()
        n = 3 if short else 30
        r = {'fwd': timeit(fwd, n, 1), 'bwd': timeit(bwd, n, 1), 'bwd_d16': timeit(bwd16, n, 1), 'bwd_data+gamma': timeit(dat, n, 1), 'bwd_data': timeit(dat0, n, 1), 'second': timeit(sec, n, 1)}
        out['B{}'.format(B)] = {k: round(v, 1) for k, v in r.items()}
        print('B =', B, out['B{}'.format(B)], flush=True)
    if short:
        return
    # phase stamps of the first tile of every workgroup (100 MHz ticks -> us), B = 128
    B = 128
    x0 = torch.randn(B, T, F, generator=g).cuda()
    maps = torch.empty((L - 1, B, T, FP, 4), dtype=torch.bfloat16, device='cuda'); gmaps = torch.empty((L, B, T, FP, 4), dtype=torch.bfloat16, device='cuda')
    a_last = torch.empty((B, T, F, 4), dtype=torch.bfloat16, device='cuda'); d_last = torch.randn(B, T, F, 4, generator=g).cuda()
    g0 = torch.empty((B, T, F), device='cuda'); u0 = torch.randn(B, T, F, generator=g).cuda(); o2 = torch.empty((B, T, F, 4), device='cuda')
    parts = torch.empty(_hip.lib().ptts_conv2d_chain_partials_bytes(L), dtype=torch.uint8, device='cuda')
    nb, npart = ctypes.c_int(0), ctypes.c_int(0)
    P, st = _hip.ptr, _hip.stream
    runs = {
        'fwd': lambda: _hip.call('ptts_conv2d_chain_fwd', P(x0), x0.stride(1), P(tab), P(maps), P(a_last), B, T, F, L, 0.3, st()),
        'bwd': lambda: _hip.call('ptts_conv2d_chain_bwd', P(d_last), 0, P(x0), x0.stride(1), P(maps), P(a_last), P(tab), P(parts), parts.numel(), ctypes.byref(nb), ctypes.byref(npart), B, T, F, L, 1, 0.3, st()),
        'data': lambda: _hip.call('ptts_conv2d_chain_bwd_data', P(d_last), 0, P(maps), P(a_last), P(tab), P(gmaps), P(g0), B, T, F, L, 0.3, st()),
        'second': lambda: _hip.call('ptts_conv2d_chain_second', P(u0), P(gmaps), P(maps), P(a_last), P(tab), P(o2), 0, P(parts), parts.numel(), ctypes.byref(nb), ctypes.byref(npart), B, T, F, L, 1, 0.3, st()),
    }
    runs['fwd'](); runs['data']()
    for name, fn in runs.items():
        buf = torch.zeros(256 * 32, dtype=torch.int64, device='cuda')
        fn(); torch.cuda.synchronize()
        _hip.lib().ptts_conv2d_chain_debug(ctypes.c_void_p(buf.data_ptr()))
        fn(); torch.cuda.synchronize()
        _hip.lib().ptts_conv2d_chain_debug(None)
        sb = buf.view(256, 32).cpu().double()
        rel = (sb - sb[:, :1]) / 100.0
        med = rel.median(dim=0).values
        last = 12 if name == 'fwd' else 10
        print(name, 'stamps (us from kernel start, median over workgroups):', [round(float(v), 1) for v in med[:last + 1]], flush=True)
        if name in ('bwd', 'data'):
            for w, o in ((0, 16), (7, 24)):
                seg = (sb[:, o:o + 8] - sb[:, o:o + 1]) / 100.0
                print('   step 2, wave', w, '[start, loads issued, dW done, acc added, conv done, bias sums, committed, barrier]:', [round(float(v), 2) for v in seg.median(dim=0).values], flush=True)
    # a critic step's stack: forward 2B + B, backward 2B, backward-data (+gamma) B, second order B
    tot = out['B128']['fwd'] + out['B64']['fwd'] + out['B128']['bwd'] + out['B64']['bwd_data+gamma'] + out['B64']['second']
    print('critic-step stack, us:', round(tot, 1), ' -> fraction of 8 TB/s on 1.993 GB:', round(1.993e9 / (tot * 1e-6) / 8e12, 3))
    json.dump(out, open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out', 'chain_probe.json'), 'w'), indent=1)

if __name__ == '__main__':
    main()

"""BLSTM step chain (B=64, T=400, H=256, both directions): stream launches vs one hipGraph replay, GPU time and host time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops
B, T, In, H = 64, 400, 256, 256
x = torch.randn(B, T, In, device='cuda') * 0.3
W = torch.randn(In, 2 * 4 * H, device='cuda') * 0.05; U = torch.randn(2, H, 4 * H, device='cuda') * 0.05; b = torch.zeros(2 * 4 * H, device='cuda')
def fwd():
    return ops.lstm_raw(x, W, U, b) if hasattr(ops, 'lstm_raw') else ops.lstm(x, W, U, b)
with torch.no_grad():
    for _ in range(3): y = fwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time(); e0.record(); y = fwd(); e1.record(); th = time.time() - t0; torch.cuda.synchronize()
    print('stream launches: GPU %.3f ms, host enqueue %.3f ms' % (e0.elapsed_time(e1), th * 1e3))
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): y = fwd()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = fwd()
    g.replay(); torch.cuda.synchronize()
    t0 = time.time(); e0.record(); g.replay(); e1.record(); th = time.time() - t0; torch.cuda.synchronize()
    print('graph replay   : GPU %.3f ms, host enqueue %.3f ms' % (e0.elapsed_time(e1), th * 1e3))

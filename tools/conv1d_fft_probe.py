"""Frequency-domain context Conv1D (ops._C1FFT) against the time-domain bf16x6 kernel at BASELINE size: per-call HIP-event times
of every stage, accuracy of both against an fp64 sample.  python3 tools/conv1d_fft_probe.py  (on the GPU box)"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops, _hip, layers
import torch.nn as nn

B, T, Cin, N, KW = 64, 400, 601, 256, 21
g = torch.Generator().manual_seed(1)


class Net(nn.Module):
    def __init__(self):
        super(Net, self).__init__()
        self.w = nn.Parameter(torch.randn(KW, Cin, N, generator=g) / math.sqrt(KW * Cin))
        self.b = nn.Parameter(torch.randn(N, generator=g))


net = Net()
flat = layers.FlatParams(net, torch.device('cuda'))
x = torch.randn(B, T, Cin, generator=g).cuda()
x2 = torch.randn(B, T, Cin, generator=g).cuda()


def run(fft, xs, bump):
    ops.conv1d_fft(fft)
    with torch.no_grad():
        for _ in range(2):
            ops.conv1d(xs[0], net.w, net.b)
        torch.cuda.synchronize()
        out = []
        for xi in xs:
            if bump:
                flat.epoch += 1                      # as after an optimiser update: the kernel's operands are rebuilt
            with _hip.KernelTimer() as kt:
                y = ops.conv1d(xi, net.w, net.b)
            out.append((y, kt.durations_ms()))
    return out


for fft in (False, True):
    for bump in (False, True):
        res = run(fft, [x, x2, x2], bump)
        for i, (y, d) in enumerate(res):
            tot = sum(t for _, _, t in d)
            print('fft=%d update=%d call %d (%s): %.3f ms  ' % (fft, bump, i, 'new x' if i < 2 else 'same x', tot) +
                  ' '.join('%s %.0f' % (n.replace('ptts_', ''), t * 1e3) for n, _, t in d))
ops.conv1d_fft(False)
with torch.no_grad():
    y0 = ops.conv1d(x, net.w, net.b)
    ops.conv1d_fft(True)
    y1 = ops.conv1d(x, net.w, net.b)
xs, ws = x[3, 150:200].double().cpu(), net.w.detach().double().cpu()
xp = torch.zeros(50 + KW - 1, Cin, dtype=torch.float64)
xp[:] = x[3, 140:210].double().cpu()
ref = torch.stack([sum(xp[t + k] @ ws[k] for k in range(KW)) for t in range(50)]) + net.b.detach().double().cpu()
sc = float(ref.abs().mean())
print('max error / mean|y|: time domain %.3e   frequency domain %.3e' % (float((y0[3, 150:200].double().cpu() - ref).abs().max()) / sc,
                                                                         float((y1[3, 150:200].double().cpu() - ref).abs().max()) / sc))

# weight gradient: frequency domain against the frame-major time-domain kernel
dy = torch.randn(B, T, N, generator=g).cuda()
for wg in (False, True):
    ops._C1FFT.wgrad_enabled = wg
    ops.conv1d_fft(True)
    for rep in range(3):
        net.w.grad = None; flat.zero_grad()
        y = ops.conv1d(x, net.w, net.b)
        with _hip.KernelTimer() as kt:
            y.backward(dy)
        d = kt.durations_ms()
    print('wgrad freq=%d: %.3f ms  ' % (wg, sum(t for _, _, t in d)) + ' '.join('%s %.0f' % (n.replace('ptts_', ''), t * 1e3) for n, _, t in d))

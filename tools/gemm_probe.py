"""Micro-probe of the two context-Conv1D GEMM shapes (forward and weight gradient) for rocprofv3 --pmc runs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percivaltts_amd import ops

B, T, Cin, N, KW = 64, 400, int(os.environ.get("CIN", 601)), int(os.environ.get("NOUT", 256)), 21
which = sys.argv[1] if len(sys.argv) > 1 else 'both'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = 'cuda'
ap = torch.randn(B, T + KW - 1, Cin, device=dev)
w = torch.randn(KW, Cin, N, device=dev) * 0.01
dy = torch.randn(B, T, N, device=dev)
y = torch.empty(B, T, N, device=dev)
dw = torch.empty_like(w)
def fwd():
    ops.gemm_raw(ap, w, y, B * T, N, KW * Cin, lda=Cin, rows_per_seg=T, seg_stride=(T + KW - 1) * Cin)
def bww():
    ops.gemm_raw(ap, dy, dw, KW * Cin, N, B * T, transA=1, lda=Cin, rows_per_seg=T, seg_stride=(T + KW - 1) * Cin)
for name, fn in (('fwd', fwd), ('bww', bww)):
    if which not in ('both', name):
        continue
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    dt = (time.time() - t) / reps
    print(name, 'ms', dt * 1e3, 'TF', 2.0 * B * T * N * KW * Cin / dt / 1e12)

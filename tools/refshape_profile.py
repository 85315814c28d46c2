"""The reference's own training geometry (run.py:76,89,125-126: B=10, T=400, 425 -> 163) for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

class A: pass
args = bench.parse.__wrapped__() if hasattr(bench.parse, '__wrapped__') else None
sys.argv = [sys.argv[0]]
args = bench.parse()
from percivaltts_amd import backend_hip
dev = backend_hip.device()
rB, rT, rctx, rspec, rnm = 10, 400, 425, 129, 33
_, rvoc, _, _, ropt = bench.build_optimizer(args, rctx, rspec, rnm, rB, 'WLSWGAN')
rb = [bench.synthetic(rB, rT, rctx, rvoc.featuressize(), rspec, 900 + i, dev) for i in range(3)]
dt, cyc = bench.timed_loop(ropt, rb, 30, 10, dev)
print('reference shape: {:.3f} ms per step'.format(dt / 30 * 1e3), cyc)

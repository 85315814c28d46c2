"""Host-side profile (cProfile) of one step kind at the headline geometry: where the Python time of the ~350 enqueued operations of
a generator step (or the ~250 of a critic step) goes.  python3 tools/host_profile.py generator|critic [bf16]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kind = sys.argv[1] if len(sys.argv) > 1 else 'generator'
bf16 = len(sys.argv) > 2 and sys.argv[2] == 'bf16'
import torch
import bench
from percivaltts_amd import parallel, backend_hip
sys.argv = ['bench.py', '--no-graph'] + (['--dtype', 'bf16'] if bf16 else [])
args = bench.parse()
parallel.init()
dev = backend_hip.device()
cfg, voc, mod, crit, opt = bench.build_optimizer(args, args.ctx, 65, 20, args.batch, args.errtype)
X, Y = bench.synthetic(args.batch, args.frames, args.ctx, voc.featuressize(), 65, 123, dev)
fn = (lambda: opt.generator_step(X, Y)) if kind == 'generator' else (lambda: opt.critic_step(X, Y))
for _ in range(3):
    opt.critic_step(X, Y); opt.generator_step(X, Y)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    fn()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
st.sort_stats('cumulative').print_stats(60)

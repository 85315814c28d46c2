"""Does a second optimiser built in the same process run slower than the first?  (bench.py's side legs build one each.)"""
import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
sys.argv = ['bench.py']
args = bench.parse()
from percivaltts_amd import backend_hip, parallel
parallel.init()
dev = backend_hip.device()
spec, nm = 65, 20
def build(bf16=None):
    return bench.build_optimizer(args, args.ctx, spec, nm, args.batch, args.errtype, bf16=bf16)
def run(opt, batches, tag):
    dt, _ = bench.timed_loop(opt, batches, 12, 6, dev)
    print('{:40s} {:6.2f} ms/step'.format(tag, dt / 12 * 1e3), flush=True)
cfg, voc, mod, crit, A = build()
batches = [bench.synthetic(args.batch, args.frames, args.ctx, voc.featuressize(), spec, 123 + i, dev) for i in range(3)]
run(A, batches, 'A (first)')
_, _, _, _, Bo = build()
run(Bo, batches, 'B (second, A alive)')
run(A, batches, 'A again')
run(Bo, batches, 'B again')
mode = os.environ.get('MODE', '')
if mode == 'wait':
    A.wait_updates(); torch.cuda.synchronize()
del A, mod, crit
gc.collect(); torch.cuda.empty_cache()
run(Bo, batches, 'B after deleting A')
_, _, _, _, Co = build()
run(Co, batches, 'C (third)')
run(Bo, batches, 'B once more')

"""Which stock torch (aten) operators still launch kernels inside a step, with their call sites' operator names and launch counts.
python3 tools/torch_ops_profile.py critic|generator [bf16]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kind = sys.argv[1] if len(sys.argv) > 1 else 'critic'
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
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    fn()
    torch.cuda.synchronize()
seen = {}
for e in prof.events():
    if not e.name.startswith('aten::') or e.device_time_total <= 0:
        continue
    if e.cpu_parent is not None and e.cpu_parent.name.startswith('aten::') and e.cpu_parent.device_time_total > 0:
        continue                      # count the outermost operator only
    st = [q for q in (e.stack or []) if 'percivaltts_amd' in q or 'bench.py' in q]
    # inside the backward pass there is no Python frame: name the autograd node instead
    par, node = e.cpu_parent, None
    while par is not None:
        if 'Backward' in par.name or 'AccumulateGrad' in par.name or 'Fn' in par.name:
            node = par.name
        par = par.cpu_parent
    where = (st[0].split('/')[-1] if st else '') + ((' [' + node + ']') if node else '')
    k = (e.name, where, tuple(e.input_shapes[0]) if e.input_shapes else ())
    c = seen.setdefault(k, [0, 0.0])
    c[0] += 1; c[1] += e.device_time_total
for (name, where, shp), (n, t) in sorted(seen.items(), key=lambda kv: -kv[1][1]):
    print('%-22s n=%2d %7.1f us  %-26s %s' % (name, n, t, shp, where))

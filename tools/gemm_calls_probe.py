"""Which ptts_gemm / grouped products a critic step and a generator step launch, with HIP-event times (one stream)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
sys.argv = ['bench.py']
args = bench.parse()
from percivaltts_amd import backend_hip, parallel, _hip
parallel.init()
dev = backend_hip.device()
cfg, voc, mod, crit, opt = bench.build_optimizer(args, args.ctx, 65, 20, args.batch, args.errtype)
X, Y = bench.synthetic(args.batch, args.frames, args.ctx, voc.featuressize(), 65, 123, dev)
opt.cfg.train_wgan_parallel_streams = False
opt._model.kerasmodel.parallel_branches = False
for kind, fn in (('critic', lambda: opt.critic_step(X, Y)), ('generator', lambda: opt.generator_step(X, Y))):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    recs = []
    for _ in range(4):
        with _hip.KernelTimer() as kt:
            fn()
        recs.append(kt.durations_ms())
    agg = collections.OrderedDict()
    for i, (nm, tag, _) in enumerate(recs[0]):
        d = sum(r[i][2] for r in recs if len(r) == len(recs[0]) and r[i][0] == nm) / len(recs)
        k = (nm, tag)
        a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
    tot = sum(v[1] for v in agg.values())
    print('== {} step: {} calls, {:.3f} ms of kernels'.format(kind, len(recs[0]), tot))
    for (nm, tag), (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if d > 0.02: print('  {:34s} {:40s} x{:3d}  {:7.3f} ms  ({:6.1f} us each)'.format(nm, str(tag), n, d, d / n * 1e3))

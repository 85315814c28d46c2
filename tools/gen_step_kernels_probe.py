"""C-ABI calls of ONE eager generator step at BASELINE configs[1] (single stream), summed per entry point and shape tag, HIP-event times
(median of N steps); the BLSTM chain's launches are listed apart.    python tools/gen_step_kernels_probe.py [nsteps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from percivaltts_amd import _hip, backend_hip


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    sys.argv = sys.argv[:1]
    args = bench.parse()
    cfg, voc, mod, crit, opt = bench.build_optimizer(args, args.ctx, 65, 20, args.batch, args.errtype, graph=False)
    dev = backend_hip.device()
    X, Y = bench.synthetic(args.batch, args.frames, args.ctx, voc.featuressize(), 65, 123, dev)
    opt.cfg.train_wgan_parallel_streams = False
    opt._model.kerasmodel.parallel_branches = False
    for _ in range(2):
        opt.generator_step(X, Y)
    torch.cuda.synchronize()
    recs = []
    for _ in range(n):
        with _hip.KernelTimer() as kt:
            opt.generator_step(X, Y)
        recs.append(kt.durations_ms())
    med = lambda xs: sorted(xs)[len(xs) // 2]
    per = {}
    for i, (nm, tag, _) in enumerate(recs[0]):
        d = med([r[i][2] for r in recs])
        k = (nm, str(tag)[:60])
        c, t = per.get(k, (0, 0.0))
        per[k] = (c + 1, t + d)
    tot = sum(t for _, t in per.values())
    print('generator step: {} C-ABI calls, {:.3f} ms of kernel time (HIP events around each call)'.format(len(recs[0]), tot))
    for (nm, tag), (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:45]:
        print('{:8.1f} us  x{:3d}  {:36s} {}'.format(t * 1e3, c, nm, tag))


if __name__ == '__main__':
    main()

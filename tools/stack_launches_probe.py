"""Every conv2d launch of ONE eager critic step at BASELINE configs[1] (fake sample given: the critic's own launches only), in launch
order, with its HIP-event time (median over N steps).    python tools/stack_launches_probe.py [nsteps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from percivaltts_amd import _hip, backend_hip


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    sys.argv = sys.argv[:1]
    args = bench.parse()
    cfg, voc, mod, crit, opt = bench.build_optimizer(args, args.ctx, 65, 20, args.batch, args.errtype, graph=False)
    dev = backend_hip.device()
    X, Y = bench.synthetic(args.batch, args.frames, args.ctx, voc.featuressize(), 65, 123, dev)
    opt.cfg.train_wgan_parallel_streams = False
    opt._model.kerasmodel.parallel_branches = False
    with torch.no_grad():
        fake = opt._fake_sample(X, True)
    for _ in range(3):
        opt.critic_step(X, Y, None, fake)
    torch.cuda.synchronize()
    recs = []
    for _ in range(n):
        with _hip.KernelTimer() as kt:
            opt.critic_step(X, Y, None, fake)
        recs.append(kt.durations_ms())
    med = lambda xs: sorted(xs)[len(xs) // 2]
    tot = 0.0
    for i, (nm, tag, _) in enumerate(recs[0]):
        d = med([r[i][2] for r in recs])
        if 'conv2d' in nm:
            tot += d
            print('{:3d} {:34s} {:8.1f} us  {}'.format(i, nm, d * 1e3, tag))
    print('conv2d launches total {:.1f} us;   whole step {:.1f} us in {} calls'.format(
        tot * 1e3, sum(med([r[i][2] for r in recs]) for i in range(len(recs[0]))) * 1e3, len(recs[0])))


if __name__ == '__main__':
    main()

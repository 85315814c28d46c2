"""C-ABI calls of ONE eager critic step (fake sample given) at BASELINE configs[1], summed per entry point and shape tag (median of N steps).
python tools/critic_step_kernels_probe.py [name-filter]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from percivaltts_amd import _hip, backend_hip


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ''
    sys.argv = sys.argv[:1]
    args = bench.parse()
    cfg, voc, mod, crit, opt = bench.build_optimizer(args, args.ctx, 65, 20, args.batch, args.errtype, graph=False)
    dev = backend_hip.device()
    X, Y = bench.synthetic(args.batch, args.frames, args.ctx, voc.featuressize(), 65, 123, dev)
    opt.cfg.train_wgan_parallel_streams = False
    with torch.no_grad():
        fake = opt._fake_sample(X, True)
    for _ in range(3):
        opt.critic_step(X, Y, None, fake)
    torch.cuda.synchronize()
    recs = []
    for _ in range(7):
        with _hip.KernelTimer() as kt:
            opt.critic_step(X, Y, None, fake)
        recs.append(kt.durations_ms())
    med = lambda xs: sorted(xs)[len(xs) // 2]
    per = {}
    for i, (nm, tag, _) in enumerate(recs[0]):
        d = med([r[i][2] for r in recs])
        k = (nm, str(tag)[:70])
        c, t = per.get(k, (0, 0.0))
        per[k] = (c + 1, t + d)
    print('critic step: {} calls, {:.3f} ms'.format(len(recs[0]), sum(t for _, t in per.values())))
    for (nm, tag), (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        if flt in nm:
            print('{:8.1f} us  x{:3d}  {:36s} {}'.format(t * 1e3, c, nm, tag))


if __name__ == '__main__':
    main()

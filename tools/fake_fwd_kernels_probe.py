"""C-ABI calls of the frozen generator's fake-sample forward (the spectral branch, batch statistics, no autograd) at BASELINE configs[1],
HIP-event times, in launch order.    python tools/fake_fwd_kernels_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from percivaltts_amd import _hip, backend_hip


def main():
    sys.argv = sys.argv[:1]
    args = bench.parse()
    cfg, voc, mod, crit, opt = bench.build_optimizer(args, args.ctx, 65, 20, args.batch, args.errtype, graph=False)
    dev = backend_hip.device()
    X, Y = bench.synthetic(args.batch, args.frames, args.ctx, voc.featuressize(), 65, 123, dev)
    for _ in range(3):
        opt._fake_sample(X, True)
    torch.cuda.synchronize()
    recs = []
    for _ in range(7):
        with _hip.KernelTimer() as kt:
            opt._fake_sample(X, True)
        recs.append(kt.durations_ms())
    med = lambda xs: sorted(xs)[len(xs) // 2]
    tot = 0.0
    for i, (nm, tag, _) in enumerate(recs[0]):
        d = med([r[i][2] for r in recs]); tot += d
        print('{:3d} {:34s} {:8.1f} us  {}'.format(i, nm, d * 1e3, str(tag)[:70]))
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): opt._fake_sample(X, True)
    e1.record(); torch.cuda.synchronize()
    print('sum of the calls {:.1f} us;  wall per forward (eager, 20 in a row) {:.1f} us'.format(tot * 1e3, e0.elapsed_time(e1) / 20 * 1e3))


if __name__ == '__main__':
    main()

"""Host time (no synchronisation) spent issuing each part of a generator batch in the steady-state loop with the look-ahead on: the
generator's early forward (issued in front of the previous batch's critic step), the critic step's replay, the generator step's
remainder -- against the device time of the critic step that runs meanwhile.   python tools/host_time_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from percivaltts_amd import backend_hip


def main():
    sys.argv = sys.argv[:1]
    args = bench.parse()
    cfg, voc, mod, crit, opt = bench.build_optimizer(args, args.ctx, 65, 20, args.batch, args.errtype)
    dev = backend_hip.device()
    batches = [bench.synthetic(args.batch, args.frames, args.ctx, voc.featuressize(), 65, 123 + i, dev) for i in range(3)]
    acc = {}
    def timed(name, fn):
        def w(*a, **k):
            t = time.perf_counter(); r = fn(*a, **k); acc.setdefault(name, []).append(time.perf_counter() - t); return r
        return w
    opt.generator_forward_early = timed('generator_forward_early', opt.generator_forward_early)
    opt.generator_step = timed('generator_step (remainder)', opt.generator_step)
    opt._graphed = timed('_graphed (critic replay)', opt._graphed)
    opt.critic_step = timed('critic_step (eager)', opt.critic_step)
    n = 60
    for i in range(n + 20):
        if i == 20:
            torch.cuda.synchronize(); acc.clear(); t0 = time.perf_counter()
        X, Y = batches[i % 3]
        opt.device_step(i, X, Y, nxt=batches[(i + 1) % 3])
    host_total = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print('per batch: wall {:.3f} ms, host issue time {:.3f} ms'.format(wall / n * 1e3, host_total / n * 1e3))
    for k, v in acc.items():
        print('  {:32s} calls {:3d}  host {:.3f} ms per call'.format(k, len(v), sum(v) / len(v) * 1e3))


if __name__ == '__main__':
    main()

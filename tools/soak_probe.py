"""Soak: N train_on_batch steps of the steady-state schedule with the look-ahead; device memory and the device status word every 500
steps (a leak or a drifting allocation pattern would show as growth).    python tools/soak_probe.py [nsteps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from percivaltts_amd import backend_hip, _hip


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    sys.argv = sys.argv[:1]
    args = bench.parse()
    cfg, voc, mod, crit, opt = bench.build_optimizer(args, args.ctx, 65, 20, args.batch, args.errtype)
    dev = backend_hip.device()
    batches = [bench.synthetic(args.batch, args.frames, args.ctx, voc.featuressize(), 65, 123 + i, dev) for i in range(3)]
    t0 = time.time()
    for i in range(n):
        X, Y = batches[i % 3]
        lc, lg = opt.device_step(i, X, Y, nxt=batches[(i + 1) % 3])
        if (i + 1) % 500 == 0:
            torch.cuda.synchronize()
            _hip.check_status()
            print('step {:5d}: {:.2f} ms/step, allocated {:.1f} MB, reserved {:.1f} MB, peak {:.1f} MB, critic loss {:.4f}'.format(
                i + 1, (time.time() - t0) / (i + 1) * 1e3, torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20,
                torch.cuda.max_memory_allocated() / 2**20, float(lc)), flush=True)


if __name__ == '__main__':
    main()

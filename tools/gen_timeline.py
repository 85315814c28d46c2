"""Timeline of ONE generator step from a rocprofv3 kernel trace (run on the GPU box):

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/gen_timeline.py run [bf16]
    python3 tools/gen_timeline.py report gpurun_out/tl > gpurun_out/tl/report.txt

`run` does 3 warm generator steps and 2 traced ones separated by marker kernels (an erfinv_ of 64 elements: no other kernel of the step has that name);
`report` cuts the last step out of the trace and prints, per HIP queue, the busy time and the span of the kernels,
and the sequence of (queue, kernel family) segments in time order - which chain is the critical path of the step."""
import csv
import glob
import os
import sys


def run(bf16):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    from percivaltts_amd import parallel, backend_hip
    graph = os.environ.get('TL_GRAPH', '0') == '1'       # whole-step hipGraph (with PTTS_GRAPH_STREAMS=1: fork / join kept): no host in the way
    sys.argv = ['bench.py', '--graph' if graph else '--no-graph'] + (['--dtype', 'bf16'] if bf16 else [])
    args = bench.parse()
    parallel.init()
    dev = backend_hip.device()
    cfg, voc, mod, crit, opt = bench.build_optimizer(args, args.ctx, 65, 20, args.batch, args.errtype)
    X, Y = bench.synthetic(args.batch, args.frames, args.ctx, voc.featuressize(), 65, 123, dev)
    kind = os.environ.get('TL_KIND', 'generator')        # which step is cut out of the trace
    if kind == 'critic':
        gen = (lambda: opt._graphed('critic', X, Y)) if graph else (lambda: opt.critic_step(X, Y))
    else:
        gen = (lambda: opt._graphed('generator', X, Y)) if graph else (lambda: opt.generator_step(X, Y))
    for _ in range(3):
        opt.critic_step(X, Y)
        gen()
    torch.cuda.synchronize()
    mark = torch.zeros(64, device=dev)
    for _ in range(2):
        mark.erfinv_()
        torch.cuda.synchronize()
        gen()
        torch.cuda.synchronize()
    mark.erfinv_()
    torch.cuda.synchronize()


def family(name):
    n = name.replace('void ', '').replace('ptts::', '')
    n = n.split('(')[0]
    return n.split('<')[0][:40]


def report(d):
    f = max(glob.glob(d + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # markers: elementwise adds over 7777 elements = grid 7777 rounded... find by Grid_Size; fall back on the gaps
    marks = [i for i, r in enumerate(rows) if 'erfinv' in r['Kernel_Name']]
    if len(marks) < 3:
        print('markers not found (%d); grid sizes of elementwise kernels:' % len(marks), sorted({r.get('Grid_Size_X', r.get('Grid_Size', '')) for r in rows if 'elementwise' in r['Kernel_Name']})[:20])
        return
    a, b = marks[-2], marks[-1]
    step = rows[a + 1:b]
    t0 = int(step[0]['Start_Timestamp'])
    t1 = max(int(r['End_Timestamp']) for r in step)
    print('generator step: %d kernels, %.3f ms from first start to last end' % (len(step), (t1 - t0) / 1e6))
    qs = {}
    for r in step:
        qs.setdefault(r['Queue_Id'], []).append(r)
    for q, rs in sorted(qs.items(), key=lambda kv: int(kv[1][0]['Start_Timestamp'])):
        busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
        print('queue %s: %4d kernels, busy %.3f ms, span %.3f .. %.3f ms' % (q, len(rs), busy / 1e6, (int(rs[0]['Start_Timestamp']) - t0) / 1e6,
              (max(int(r['End_Timestamp']) for r in rs) - t0) / 1e6))
    fam = {}
    for r in step:
        f = family(r['Kernel_Name'])
        c = fam.setdefault(f, [0, 0])
        c[0] += 1; c[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    print('\nkernel families of the step (launches, total us, mean us):')
    for f, (n, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print('  %4d %9.1f %8.1f  %s' % (n, t / 1e3, t / 1e3 / n, f))
    if os.environ.get('TL_NOSEG', '0') == '1':
        return
    # segments: consecutive kernels of the same (queue, family)
    print('\nsegments (start ms, end ms, queue, n, busy ms, family):')
    for q, rs in sorted(qs.items(), key=lambda kv: int(kv[1][0]['Start_Timestamp'])):
        seg = None
        for r in rs:
            fam = family(r['Kernel_Name'])
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            if seg and seg[3] == fam:
                seg[1] = e
                seg[4] += 1
                seg[5] += e - s
            else:
                if seg:
                    print('  %8.3f %8.3f  q%s %4d %7.3f  %s' % ((seg[0] - t0) / 1e6, (seg[1] - t0) / 1e6, seg[2], seg[4], seg[5] / 1e6, seg[3]))
                seg = [s, e, q, fam, 1, e - s]
        if seg:
            print('  %8.3f %8.3f  q%s %4d %7.3f  %s' % ((seg[0] - t0) / 1e6, (seg[1] - t0) / 1e6, seg[2], seg[4], seg[5] / 1e6, seg[3]))
        print()


def events(bf16):
    """No profiler: HIP events at the start / end of the step and around the BLSTM chains (python3 tools/gen_timeline.py events [bf16])."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    from percivaltts_amd import parallel, backend_hip, ops
    sys.argv = ['bench.py', '--no-graph'] + (['--dtype', 'bf16'] if bf16 else [])
    args = bench.parse()
    parallel.init()
    dev = backend_hip.device()
    cfg, voc, mod, crit, opt = bench.build_optimizer(args, args.ctx, 65, 20, args.batch, args.errtype)
    X, Y = bench.synthetic(args.batch, args.frames, args.ctx, voc.featuressize(), 65, 123, dev)
    for _ in range(3):
        opt.critic_step(X, Y)
        opt.generator_step(X, Y)
    torch.cuda.synchronize()
    whole = os.environ.get('TL_STEP', '0') == '1'       # the whole train_on_batch of a batch that trains both networks (device_step)
    for rep in range(3):
        ops.lstm_trace = []
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.time()
        e0.record()
        if whole:
            opt.device_step(0, X, Y)
        else:
            opt.generator_step(X, Y)
        th = time.time() - t0
        opt.wait_updates()
        e1.record()
        torch.cuda.synchronize()
        tr, ops.lstm_trace = ops.lstm_trace, None
        print('generator step %.3f ms (host enqueue %.3f ms): ' % (e0.elapsed_time(e1), th * 1e3) + '  '.join('%s %.3f' % (tag, e0.elapsed_time(ev)) for tag, ev in tr))


if __name__ == '__main__':
    if sys.argv[1] == 'events':
        import time
        events(len(sys.argv) > 2 and sys.argv[2] == 'bf16')
    elif sys.argv[1] == 'run':
        run(len(sys.argv) > 2 and sys.argv[2] == 'bf16')
    else:
        report(sys.argv[2])

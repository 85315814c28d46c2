"""CPU, world_size 2, gloo: the data-parallel plumbing of the WGAN-GP step (flat-gradient all-reduce, shard of the
minibatch, replica broadcast, max-over-ranks timing).  The reference is single-GPU (run.py:28-31); this is new."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank),
                       'WORLD_SIZE': str(world), 'LOCAL_RANK': str(rank)})
    from percivaltts_amd import parallel
    w, r = parallel.init(backend='gloo')
    assert (w, r) == (world, rank) and parallel.world_size() == world and parallel.rank() == rank
    # each rank's flat gradient; the Adam kernel multiplies by the returned 1/world (gscale)
    g = torch.arange(10, dtype=torch.float32) * (rank + 1)
    scale = parallel.allreduce_sum_(g)
    ok = torch.allclose(g * scale, torch.arange(10, dtype=torch.float32) * 1.5) and abs(scale - 0.5) < 1e-12
    # the explicit form the optimiser's update uses: start the collective, order it with work.wait(), then consume
    g2 = torch.arange(6, dtype=torch.float32) * (rank + 1)
    work, scale2 = parallel.allreduce_sum_async(g2)
    ok = ok and work is not None
    work.wait()
    ok = ok and torch.allclose(g2 * scale2, torch.arange(6, dtype=torch.float32) * 1.5) and abs(scale2 - 0.5) < 1e-12
    lo, hi = parallel.shard_batch(128)
    ok = ok and (lo, hi) == (rank * 64, (rank + 1) * 64)
    p = torch.full((5,), float(rank + 7))
    parallel.broadcast_(p, src=0)
    ok = ok and bool((p == 7.0).all())
    t = parallel.max_over_ranks(1.0 + rank)
    ok = ok and t == 2.0
    try:
        parallel.shard_batch(7)
        ok = False
    except ValueError:
        pass
    parallel.barrier()
    q.put((rank, ok))


def test_two_rank_gradient_allreduce_and_sharding():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_single_process_is_a_noop():
    from percivaltts_amd import parallel
    g = torch.ones(4)
    assert parallel.allreduce_sum_(g) == 1.0 and parallel.world_size() == 1 and parallel.shard_batch(10) == (0, 10)
    assert parallel.allreduce_sum_async(g) == (None, 1.0)

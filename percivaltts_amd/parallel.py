"""Data parallelism for the WGAN-GP step: one process per GPU, minibatch sharded across ranks, one all-reduce
of the flat gradient buffer per optimiser step (the reference is single-GPU, run.py:28-31).

Backend "nccl" is RCCL on ROCm (xGMI inside a node); "gloo" serves the CPU tests.  The critic's gradient is
3.6 M floats (14.5 MB), the generator's 4.7 M (18.8 MB): one flat bucket each, fp32 sum, scaled by 1/world inside
the Adam kernel (gscale).  Per-rank BatchNorm statistics (B=64 per rank) are the benchmarked semantics.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment; no-op for a single process."""
    world, rank, local = env_world()
    if world <= 1:
        return 1, 0
    if not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('PTTS_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist.get_world_size(), dist.get_rank()


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_sum_(flat):
    """In-place sum of a flat buffer over all ranks; returns the factor that turns it into the mean."""
    w = world_size()
    if w > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return 1.0 / w


def allreduce_sum_async(flat):
    """Start the in-place sum of a flat buffer over all ranks on the CURRENT stream's behalf and return (work, factor): the
    caller orders it explicitly -- `work.wait()` makes the current stream (RCCL) or the host (gloo) wait for the result, whatever
    stream the backend ran the collective on -- instead of relying on what a blocking call does with the current stream.
    work is None in a single process."""
    w = world_size()
    if w <= 1:
        return None, 1.0
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), 1.0 / w


def broadcast_(flat, src=0):
    if world_size() > 1:
        dist.broadcast(flat, src=src)
    return flat


def is_main():
    """True on the rank that writes files (checkpoints, training state, logs): rank 0."""
    return rank() == 0


def average_buffers_(tensors):
    """In-place mean over the ranks of a list of (small) tensors, as ONE flat all-reduce: the BatchNorm moving statistics,
    which every rank accumulates from its own shard of the batches, before they are validated with and saved."""
    w = world_size()
    tensors = [t for t in tensors if t is not None and t.numel() > 0]
    if w <= 1 or not tensors:
        return tensors
    flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.mul_(1.0 / w)
    off = 0
    with torch.no_grad():
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view(t.shape))
            off += n
    return tensors


def shard_batch(n_items, world=None, r=None):
    """Contiguous shard [lo, hi) of a global batch for this rank; the global batch must divide evenly."""
    world = world_size() if world is None else world
    r = rank() if r is None else r
    if n_items % world != 0:
        raise ValueError('global batch {} does not divide over {} ranks'.format(n_items, world))
    per = n_items // world
    return r * per, (r + 1) * per


def barrier():
    if world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device=None):
    """Max of a python float over the ranks (bench timing)."""
    if world_size() <= 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

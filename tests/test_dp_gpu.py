"""Data-parallel WGAN-GP step on real kernels: two ranks (gloo, both on the one GPU of the test box) must end with the
same critic / generator weights as one process that averages the two shard gradients by hand."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _setup():
    import io, contextlib
    import test_model_gpu as tm
    with contextlib.redirect_stdout(io.StringIO()):
        cfg, voc, mod, crit, a, gw, cw, X, Y, al = tm.build('test')
    return tm, cfg, mod, crit, X, Y, al


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank), 'WORLD_SIZE': str(world),
                       'LOCAL_RANK': '0', 'PTTS_DIST_BACKEND': 'gloo'})
    import io, contextlib
    from percivaltts_amd import optimizertts_wgan, parallel
    tm, cfg, mod, crit, X, Y, al = _setup()
    with contextlib.redirect_stdout(io.StringIO()):
        opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype='WLSWGAN', critic=crit)
        opt.prepare()
    assert opt.world == world
    lo, hi = parallel.shard_batch(X.shape[0], world, rank)
    Xd, Yd, ald = tm.f32(X[lo:hi]), tm.f32(Y[lo:hi]), tm.f32(al[lo:hi])
    for _ in range(2):
        opt.critic_step(Xd, Yd, ald)
    opt.generator_step(Xd, Yd)
    torch.cuda.synchronize()
    q.put((rank, opt.critic_opti.flat.flat.cpu().numpy(), opt.gen_opti.flat.flat.cpu().numpy()))
    parallel.barrier()


def test_two_ranks_equal_manual_gradient_averaging():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=0, atol=0)      # replicas stay identical
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=0, atol=0)

    # single process: same two shards, gradients averaged by hand
    import io, contextlib
    from percivaltts_amd import optimizertts_wgan
    tm, cfg, mod, crit, X, Y, al = _setup()
    with contextlib.redirect_stdout(io.StringIO()):
        opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype='WLSWGAN', critic=crit)
        opt.prepare()
    shards = [(tm.f32(X[i:i + 1]), tm.f32(Y[i:i + 1]), tm.f32(al[i:i + 1])) for i in range(2)]
    for _ in range(2):
        acc = torch.zeros_like(opt.critic_opti.flat.grad)
        for (xs, ys, as_) in shards:
            opt.critic_opti.zero_grad()
            total, _ = opt.critic_loss(xs, ys, as_, training=True)
            total.backward()
            acc += opt.critic_opti.flat.grad
        opt.critic_opti.flat.grad.copy_(acc)
        opt.critic_opti.step(0.5)
    acc = torch.zeros_like(opt.gen_opti.flat.grad)
    snap = [b.clone() for b in opt._model.kerasmodel.buffers()]
    for (xs, ys, _) in shards:
        opt.gen_opti.zero_grad()
        for p in opt.critic_opti.flat.params: p.requires_grad_(False)
        total, _ = opt.generator_loss(xs, ys, training=True)
        total.backward()
        for p in opt.critic_opti.flat.params: p.requires_grad_(True)
        acc += opt.gen_opti.flat.grad
    opt.gen_opti.flat.grad.copy_(acc)
    opt.gen_opti.step(0.5)
    torch.cuda.synchronize()
    np.testing.assert_allclose(res[0][1], opt.critic_opti.flat.flat.cpu().numpy(), rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(res[0][2], opt.gen_opti.flat.flat.cpu().numpy(), rtol=2e-4, atol=2e-6)


def _worker_syncbn(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank), 'WORLD_SIZE': str(world),
                       'LOCAL_RANK': '0', 'PTTS_DIST_BACKEND': 'gloo'})
    import io, contextlib
    from percivaltts_amd import optimizertts_wgan, parallel
    tm, cfg, mod, crit, X, Y, al = _setup()
    cfg.train_sync_batchnorm = True
    cfg.train_wgan_critic_LSWGANtransidx = 30.0
    with contextlib.redirect_stdout(io.StringIO()):
        opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype='WLSWGAN', critic=crit)
        opt.prepare()
    lo, hi = parallel.shard_batch(X.shape[0], world, rank)
    Xd, Yd = tm.f32(X[lo:hi]), tm.f32(Y[lo:hi])
    lg = opt.generator_step(Xd, Yd)
    torch.cuda.synchronize()
    moving = np.concatenate([t.detach().cpu().numpy().ravel() for k, t in mod.kerasmodel.weights() if 'moving' in k])
    q.put((rank, opt.gen_opti.flat.grad.cpu().numpy() / world, float(lg), moving))
    parallel.barrier()


def test_syncbn_two_ranks_equal_one_process_on_the_whole_batch():
    """cfg.train_sync_batchnorm (SURVEY 8e note 1): with the BatchNorm sums all-reduced, two ranks on the two halves of a
    batch take the generator step of ONE process on the whole batch -- same moving statistics after the step, the mean of
    the two ranks' losses is the loss of the batch, and the all-reduced, 1/W-scaled gradient is the gradient of the batch.
    (Per-rank statistics, the default, are the single-GPU semantics of the rank's own B samples instead.)"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_syncbn, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][3], res[1][3])
    # the same step by one process on the whole batch
    from percivaltts_amd import optimizertts_wgan
    import io, contextlib
    tm, cfg, mod, crit, X, Y, al = _setup()
    cfg.train_wgan_critic_LSWGANtransidx = 30.0
    with contextlib.redirect_stdout(io.StringIO()):
        opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype='WLSWGAN', critic=crit)
        opt.prepare()
    lg = opt.generator_step(tm.f32(X), tm.f32(Y))
    torch.cuda.synchronize()
    g_full = opt.gen_opti.flat.grad.cpu().numpy()
    moving = np.concatenate([t.detach().cpu().numpy().ravel() for k, t in mod.kerasmodel.weights() if 'moving' in k])
    np.testing.assert_allclose(0.5 * (res[0][2] + res[1][2]), float(lg), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(res[0][3], moving, rtol=1e-5, atol=1e-6)
    err = np.linalg.norm(res[0][1] - g_full) / np.linalg.norm(g_full)
    assert err < 1e-4, err


def test_bench_two_ranks_prints_one_line():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank; here gloo and both
    ranks on the one GPU): every collective must be entered by both ranks, rank 0 prints ONE JSON line."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PTTS_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--batch', '4', '--frames', '40', '--no-cpu-baseline', '--no-gated-leg', '--no-reference-shape', '--no-bf16-leg']
    out = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['config']['global_batch'] == 8 and res['scaling'] == 'weak'
    assert res['value'] > 0 and 'roofline' in res
    # a multi-GPU run prints the headline loop, the two step times and the two all-reduce times -- no side legs unless --side-legs
    assert res['allreduce_ms']['backend'] == 'gloo' and res['allreduce_ms']['critic_grads'] > 0 and res['allreduce_ms']['generator_grads'] > 0
    assert 'critic_step_ms' in res and 'generator_step_ms' in res and 'variant_fp32_mfma_gemms' not in res and 'pcie_inclusive' not in res
    assert 'per rank' in res['config']['batchnorm_statistics'] and res['config']['collective_backend'] == 'gloo'

#!/usr/bin/env python
"""bench.py -- acoustic frames/sec of the WGAN-GP hot path on MI355X (BASELINE.json metric).

A "step" is one `train_on_batch` (reference optimizertts_wgan.py:216-241) on one resident synthetic batch:
one critic step, plus one generator step on every 5th batch (the steady-state schedule, :225-231).
Workload at N=1 is BASELINE.json configs[1]: [64,400,601] -> [64,400,86] (f0 1 + spec 65 + noise 20), DCNN generator
+ 2D-conv critic, fp32, errtype WLSWGAN (run.py's default), lambda = 10.  With N ranks each rank works on its own
64-sample shard (weak scaling, global batch 64*N) and the flat gradients are all-reduced over RCCL.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline       SURVEY.md 8(d)'s quantity (round 4): the critic's whole 2D-conv stack, (149 C + 3) B T F s algorithmic bytes per critic
                 step over the summed HIP-event time of every conv2d launch of the step, against the 8 TB/s HBM roofline; the stack's
                 dominant kernel per launch is the sub-field `dominant_kernel` (the 4 -> 4 Conv2D layer kernel; --dtype bf16: the
                 stack is its four fused conv2d_chain launches)
  roofline_conv2d the same figures under their old name
  roofline_conv1d the context Conv1D's big product against the matrix-core peak (frequency domain: batched dense_bf16x6_kernel; time domain:
                 gemm_bf16x6_kernel / gemm_bf16x1_kernel)
  roofline_lstm  the BLSTM recurrence (2 x T step launches per generator step: by time in the loop the top kernel), fp32 MFMA peak
  cpu_baseline   the CPU oracle (PyTorch-CPU fp32 restatement of the reference path) on a bounded sample
With more than one rank: `allreduce_ms` (both buckets, backend, the exposed -- not overlapped -- update time per critic step).
--graph-critic / --graph-generator {on,off} pin the form of each step kind instead of the 'tune' timing comparison.
"""
from __future__ import print_function

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 (same table); the bf16x6 split product spends six bf16 products per fp32 product
PEAK_HBM_GBPS = 8000.0          # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500, help='timed train_on_batch steps (default 500 = 100 cycles of 5 critic + 1 generator step)')
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--frames', type=int, default=400)
    ap.add_argument('--ctx', type=int, default=601)
    ap.add_argument('--errtype', default='WLSWGAN')
    ap.add_argument('--graph', action='store_true', help='capture each step once and replay it as a hipGraph (single stream)')
    ap.add_argument('--no-hoist', action='store_true', help="do not launch the generator's forward before the critic step of a batch that trains both (cfg.train_wgan_hoist_generator)")
    ap.add_argument('--no-lookahead', action='store_true', help="do not launch a generator step's forward one batch ahead (cfg.train_wgan_generator_lookahead)")
    ap.add_argument('--no-graph', action='store_true', help="never replay a step as a hipGraph (the default, cfg.train_wgan_hipgraph = 'tune', times eager launches against a replay per step kind on the first batch and keeps the faster)")
    ap.add_argument('--eager', action='store_true', help='(default) eager launches; kept for compatibility')
    ap.add_argument('--graph-critic', choices=['on', 'off'], default=None, help="pin the critic step's form: hipGraph replay (on) or eager launches (off), instead of the 'tune' timing comparison on the first batch -- so that a profile and a bench run time ONE program")
    ap.add_argument('--graph-generator', choices=['on', 'off'], default=None, help="pin the generator step's form likewise (on: its forward is then not hoisted in front of the critic step)")
    ap.add_argument('--no-prune', action='store_true', help="also run G's f0/noise branches in the critic step")
    ap.add_argument('--no-stack', action='store_true', help='evaluate critic(real) and critic(fake) separately instead of as one 2B pass')
    ap.add_argument('--no-ctx-reuse', action='store_true', help="recompute the generator's context Conv1D in the generator step instead of taking the critic step's product of the same batch")
    ap.add_argument('--no-early-critic', action='store_true', help='generator step: critic(G(x)) on the concatenated output (waits for the BLSTM branch) instead of on the spectral branch')
    ap.add_argument('--no-streams', action='store_true', help='single HIP stream (default: the three critic evaluations and the BLSTM branch on side streams)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--split-bf16', action='store_true', help='(default) context Conv1D forward and weight gradient as bf16x6 split products on the bf16 matrix cores: fp32 arithmetic by three-way bf16 splits, six products, fp32 accumulation')
    ap.add_argument('--fp32-mfma', action='store_true', help='context Conv1D forward and weight gradient on the fp32 MFMA pipe instead of the bf16x6 split products')
    ap.add_argument('--no-reference-shape', action='store_true', help="skip the leg at the reference's own training geometry (B=10, T=400, 425 -> 163)")
    ap.add_argument('--no-unreduced', action='store_true', help='skip the timed loop with every exact work reduction switched off')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'], help="bf16: BASELINE configs[2] -- the maps between the critic's 4->4 Conv2D layers and their gradients stored as bf16, bf16 products with fp32 accumulation; master weights, weight gradients and everything outside the stack fp32")
    ap.add_argument('--no-bf16-leg', action='store_true', help='skip the configs[2] side leg of the default (fp32) run')
    ap.add_argument('--sync-bn', action='store_true', help='data parallelism: BatchNorm statistics all-reduced over the ranks (default: per rank)')
    ap.add_argument('--no-gated-leg', action='store_true', help='skip the BASELINE configs[4] leg (gated dilated-causal generator, T=2000)')
    ap.add_argument('--gated-batch', type=int, default=64)
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-variants', action='store_true', help='skip the extra timed loop of the fp32-MFMA context-Conv1D variant')
    ap.add_argument('--no-host-leg', action='store_true', help='skip the PCIe-inclusive leg (host batches through the prefetcher)')
    ap.add_argument('--cpu-batch', type=int, default=32)
    ap.add_argument('--side-legs', action='store_true', help='with --gpus N > 1: also run the side legs (variants, reference shape, bf16, gated, host-fed); by default a multi-GPU run prints the headline loop, the two step times and the all-reduce times only')
    return ap.parse_args()


def make_cfg(args):
    import percivaltts_amd
    cfg = percivaltts_amd.configuration()
    # architecture defaults of the reference's run.py:114-120
    cfg.arch_hiddenwidth = 256
    cfg.arch_ctx_nbcnnlayers = 1
    cfg.arch_ctx_winlen = 21
    cfg.arch_gen_nbcnnlayers = 8
    cfg.arch_gen_nbfilters = 4
    cfg.arch_gen_winlen = 5
    cfg.arch_spec_freqlen = 5
    cfg.train_batch_size = args.batch
    cfg.train_wgan_LScoef = 0.25
    return cfg


def synthetic(B, T, ctx, out, spec, seed, device):
    """X ~ U(-1,1); Y: f0+spec ~ N(0,1), noise mask ~ U(0,1)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(B, T, ctx, generator=g) * 2 - 1
    Y = torch.randn(B, T, out, generator=g)
    Y[:, :, 1 + spec:] = torch.rand(B, T, out - 1 - spec, generator=g)
    return X.to(device).contiguous(), Y.to(device).contiguous()


def cpu_baseline(args, cfg_dims):
    """The oracle, fp32, all host cores, one cycle (5 critic steps + 1 generator step) at a reduced batch."""
    import torch
    from oracle import percival_oracle as O
    # the GPU box gives one job a share of the host (16 cores per GPU): more threads than that only contend
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    ncores = max(1, min(16, avail))
    torch.set_num_threads(ncores)
    print('[bench] cpu_baseline: oracle on {} threads ...'.format(ncores), file=sys.stderr, flush=True)
    ctx, spec, nm = cfg_dims
    a = O.Arch(ctx, spec, nm)
    B, T = args.cpu_batch, args.frames
    f32 = lambda ws: [w.to(torch.float32) for w in ws]
    gw = f32(O.random_weights(O.generator_weight_shapes(a), seed=1))
    cw = f32(O.random_weights(O.critic_weight_shapes(a), seed=2))
    g = torch.Generator().manual_seed(123)
    X = torch.rand(B, T, ctx, generator=g) * 2 - 1
    Y = torch.randn(B, T, a.outsize, generator=g)
    al = torch.rand(B, generator=g)
    w_ls, ww = O.wls_weights(spec, nm, 0, 0.25, 30.0)
    w_ls = torch.tensor(w_ls, dtype=torch.float32)
    gshapes = O.generator_weight_shapes(a)
    train_idx, i = [], 0
    while i < len(gshapes):
        if len(gshapes[i]) == 1 and i + 3 < len(gshapes) and all(gshapes[i + k] == gshapes[i] for k in range(4)):
            train_idx += [i, i + 1]; i += 4
        else:
            train_idx.append(i); i += 1
    cm = [torch.zeros_like(w) for w in cw]; cv = [torch.zeros_like(w) for w in cw]
    gm = [torch.zeros_like(gw[i]) for i in train_idx]; gv = [torch.zeros_like(gw[i]) for i in train_idx]
    t0 = time.time()
    for k in range(5):
        for w in cw: w.requires_grad_(True)
        total, _ = O.critic_step_loss(cw, gw, a, X, Y, al)
        grads = torch.autograd.grad(total, cw)
        for w in cw: w.requires_grad_(False)
        O.adam_keras(cw, grads, cm, cv, k + 1, 1e-4, 0.5, 0.9)
        print('[bench] cpu_baseline: critic step {}/5 at {:.1f} s'.format(k + 1, time.time() - t0), file=sys.stderr, flush=True)
    for i in train_idx: gw[i].requires_grad_(True)
    lt, _ = O.generator_step_loss(cw, gw, a, X, Y, 'WLSWGAN', w_ls, ww)
    gg = torch.autograd.grad(lt, [gw[i] for i in train_idx], allow_unused=True)
    for i in train_idx: gw[i].requires_grad_(False)
    gg = [g_ if g_ is not None else torch.zeros_like(gw[i]) for g_, i in zip(gg, train_idx)]
    O.adam_keras([gw[i] for i in train_idx], gg, gm, gv, 1, 1e-3, 0.5, 0.9)
    dt = time.time() - t0
    return {'value': 5.0 * B * T / dt, 'unit': 'frames/s', 'cores': ncores, 'kind': 'port',
            'sample': 'one cycle (5 critic + 1 generator step) at B={} T={} ctx={}, fp32 PyTorch-CPU oracle, {:.1f} s'.format(B, T, ctx, dt)}


def roofline_leg(opt, X, Y, args):
    """HIP-event timing of every C-ABI call of one eager critic step (and one generator step)."""
    import torch
    from percivaltts_amd import _hip, ops
    B, T = X.shape[0], X.shape[1]
    voc = opt._model.vocoder
    F, C, L = voc.specsize(), opt.cfg.arch_gen_nbfilters, opt.cfg.arch_gen_nbcnnlayers
    # per-kernel HIP events need every launch on the one stream the events are recorded on
    opt.cfg.train_wgan_parallel_streams = False
    opt._model.kerasmodel.parallel_branches = False
    for _ in range(2):
        opt.critic_step(X, Y)
    torch.cuda.synchronize()
    reps = 5
    recs = []
    for _ in range(reps):
        with _hip.KernelTimer() as kt:
            opt.critic_step(X, Y)
        recs.append(kt.durations_ms())
    # average per call position
    n = len(recs[0])
    # (median over the repetitions: the first launch of a kernel variant pays its code object's load -- tens of milliseconds, once)
    med = lambda xs: sorted(xs)[len(xs) // 2]
    avg = [(recs[0][i][0], recs[0][i][1], med([r[i][2] for r in recs])) for i in range(n)]
    M, N, K = B * T, opt.cfg.arch_hiddenwidth, opt.cfg.arch_ctx_winlen * X.shape[2]
    conv1d_fwd = [d for (nm, tag, d) in avg if nm == 'ptts_gemm' and tag == (M, N, K, 0, 0, 1)]
    conv1d_bww = [d for (nm, tag, d) in avg if (nm == 'ptts_gemm' and tag == (K, N, M, 1, 0, 1)) or nm in ('ptts_conv1d_wgrad_t', 'ptts_conv1d_wgrad_bf16x6')]
    conv1d_split = [d for (nm, tag, d) in avg if nm == 'ptts_conv1d_bf16x6']
    # the critic's conv2d stack: every conv2d call without a BatchNorm affine whose batch is B (G's convs carry scale/shift
    # or are 1->C without bias; separate them by running G first)
    classes = {}
    for nm, tag, d in avg:
        classes[nm] = classes.get(nm, 0.0) + d
    out = {}
    if conv1d_fwd:
        t = sum(conv1d_fwd) / len(conv1d_fwd) * 1e-3
        fl = 2.0 * M * N * K
        out['roofline_conv1d'] = {'bound': 'mfma', 'kernel': 'gemm_dma_kernel<0, 1> (context Conv1D as implicit GEMM, M={} N={} K={})'.format(M, N, K),
                           'achieved': fl / t / 1e12, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                           'frac': fl / t / 1e12 / PEAK_FP32_MFMA_TFLOPS, 'traffic': None,
                           'launch_ms': t * 1e3, 'launches_per_critic_step': len(conv1d_fwd)}
    if conv1d_split:
        # six bf16 products per fp32 product: the fp32-equivalent peak of this formulation is the dense bf16 peak / 6;
        # with cfg.train_wgan_bf16_products (configs[2]) ONE bf16 product per position: the dense bf16 peak itself
        t = sum(conv1d_split) / len(conv1d_split) * 1e-3
        fl = 2.0 * M * N * K
        one = bool(getattr(opt.cfg, 'train_wgan_bf16_products', False))
        peak = PEAK_BF16_MFMA_TFLOPS if one else PEAK_BF16_MFMA_TFLOPS / 6.0
        out['roofline_conv1d'] = {'bound': 'mfma', 'kernel': ('gemm_bf16x1_kernel (context Conv1D, M={} N={} K={}; bf16 products of the operands\' bf16 roundings, fp32 accumulation)' if one else
                                                       'gemm_bf16x6_kernel (context Conv1D, M={} N={} K={}; algorithmic fp32 flop, 6 bf16 MFMA products each)').format(M, N, K),
                           'achieved': fl / t / 1e12, 'peak': peak, 'unit': 'TFLOP/s',
                           'frac': fl / t / 1e12 / peak, 'traffic': None,
                           'launch_ms': t * 1e3, 'launches_per_critic_step': len(conv1d_split)}
    conv1d_freq = [(tag, d) for (nm, tag, d) in avg if nm == 'ptts_dense_bf16x6_batched' and tag and tag[0] == 'freq']
    if conv1d_freq:
        # the context Conv1D in the frequency domain (ops._C1FFT): its big product, NB per-frequency GEMMs [2B x 2Kh] . [2Kh x N].
        # flops = what THIS formulation executes (the time-domain product it replaces has 2 M N K = 8-10 x more)
        (_, nb, m2, n2, k2), _ = conv1d_freq[0]
        t = sum(d for _, d in conv1d_freq) / len(conv1d_freq) * 1e-3
        fl = 2.0 * nb * m2 * n2 * k2
        per_conv = {}
        for nm, tag, d in avg:
            if nm in ('ptts_dense_bf16x6_batched', 'ptts_split3_dense_weight_strided', 'ptts_dft_mirror', 'ptts_conv1d_freq_kernel_planes',
                      'ptts_transpose_batched', 'ptts_conv1d_freq_wgrad_inverse'):
                k = nm.replace('ptts_', '') + ('.' + str(tag[0]) if tag and isinstance(tag[0], str) else '')
                per_conv[k] = round(per_conv.get(k, 0.0) + d, 4)
        out['roofline_conv1d'] = {'bound': 'mfma', 'kernel': 'dns::dense_bf16x6_kernel, batched over {} frequencies (context Conv1D in the frequency domain: per-frequency product '
                                  '[{} x {}] . [{} x {}], fp32 as 6 bf16 MFMA products)'.format(nb, m2, k2, k2, n2),
                                  'achieved': fl / t / 1e12, 'peak': PEAK_BF16_MFMA_TFLOPS / 6.0, 'unit': 'TFLOP/s', 'frac': fl / t / 1e12 / (PEAK_BF16_MFMA_TFLOPS / 6.0),
                                  'traffic': None, 'launch_ms': t * 1e3, 'launches_per_critic_step': len(conv1d_freq),
                                  'flop_per_launch': fl, 'time_domain_flop_per_launch': 2.0 * M * N * K,
                                  'all_stages_ms_per_critic_step': per_conv}
    if conv1d_bww:
        t = sum(conv1d_bww) / len(conv1d_bww) * 1e-3
        out['conv1d_bwd_weight'] = {'achieved': 2.0 * M * N * K / t / 1e12, 'unit': 'TFLOP/s', 'launch_ms': t * 1e3}
    # critic conv2d stack (HBM roofline).  Generator convs are timed separately.
    with torch.no_grad():
        fake = opt._fake_sample(X, True)
    torch.cuda.synchronize()
    crit_recs = []
    for _ in range(reps):
        opt.critic_opti.zero_grad()
        with _hip.KernelTimer() as kt:
            with ops.deferred_weight_grads():          # as critic_step runs it (grouped reductions included in the sum)
                total, _ = opt.critic_loss(X, Y, None, training=True, fake=fake)
                total.backward()
        crit_recs.append(kt.durations_ms())
    # (summed per repetition: the call list of the first one may hold a refresh of cached operands the others do not)
    t_conv2d = med([sum(d for (nm, _, d) in r if nm.startswith('ptts_conv2d')) for r in crit_recs]) * 1e-3
    n_conv2d = sum(1 for (nm, _, _) in crit_recs[-1] if nm.startswith('ptts_conv2d'))
    bf16_stack = getattr(opt.cfg, 'arch_critic_bf16', False)
    # (149 C + 3) B T F s bytes per critic step, s = 4 (configs[1], fp32) or 2 (configs[2], every map bf16)
    alg_bytes = (149.0 * C + 3.0) * B * T * F * (2.0 if bf16_stack else 4.0) if L == 8 else None
    if alg_bytes and t_conv2d > 0:
        out['roofline_conv2d'] = {'bound': 'hbm', 'kernel': 'critic 2D-conv stack: {} conv2d fwd/bwd launches per critic step'.format(n_conv2d),
                                  'achieved': alg_bytes / t_conv2d / 1e9, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s',
                                  'frac': alg_bytes / t_conv2d / 1e9 / PEAK_HBM_GBPS, 'traffic': None,
                                  'algorithmic_bytes_per_critic_step': alg_bytes, 'time_ms': t_conv2d * 1e3,
                                  'maps': (('bf16, every layer (post-activation), held in the LDS between the layers of a launch: conv2d_chain kernels' if bf16_stack is True else
                                            'bf16 between the 4->4 layers (6 of 8 maps and their gradients), fp32 at both ends of the stack') if bf16_stack else 'fp32'),
                                  'launches': sorted(set(nm for (nm, _, _) in crit_recs[-1] if nm.startswith('ptts_conv2d')))}
    out['critic_step_kernel_ms'] = {k: round(v, 4) for k, v in sorted(classes.items(), key=lambda kv: -kv[1])}
    # ---- `roofline`: the dominant kernel of the critic step.  fp32: the 4 -> 4 Conv2D layers on the matrix cores (c2m::fwd_kernel: forward,
    # masked forward, backward-data), per launch (2 or 3 maps of 16 bytes per time-frequency bin) against the HBM roofline; bf16: the
    # fused stack (the roofline_conv2d figures: four launches).  Before round 3's frequency-domain Conv1D the context Conv1D was.
    # ---- `roofline` (round 4): SURVEY 8(d)'s quantity -- the critic's whole 2D-conv STACK, (149 C + 3) B T F s algorithmic bytes per
    # critic step over the summed time of all its conv2d launches (forward, backward-data, weight gradients, the 1 -> 4 layer, the
    # grouped reductions) -- with the stack's dominant kernel, per launch, as the sub-field `dominant_kernel` (fp32: the 4 -> 4 layer
    # kernel, 2 or 3 maps of 16 bytes per time-frequency bin a launch; bf16: the stack IS its four chain launches).
    c2m = [(tag, d) for (nm, tag, d) in crit_recs[-1] if nm == 'ptts_conv2d_mfma_fwd']
    if 'roofline_conv2d' in out:
        out['roofline'] = dict(out['roofline_conv2d'])
        out['roofline']['definition'] = 'SURVEY 8(d): algorithmic bytes of the critic 2D-conv stack per critic step / summed HIP-event time of every conv2d launch of the step'
        if bf16_stack is True:
            out['roofline']['kernel'] = 'c2c::chain_fwd / chain_bwd / chain_bwd_data / chain_second kernels (critic Conv2D stack, 8 layers per launch, bf16 maps): all launches of a critic step'
        elif c2m:
            by = 0.0
            for tag, d in c2m:
                Bq, Tq, Fq, _, _, mode, has_om, planes = tag
                by += Bq * Tq * Fq * 4 * 4.0 * (2 + (1 if (mode == 2 or has_om) else 0))
            tt = med([sum(d for (nm, _, d) in r if nm == 'ptts_conv2d_mfma_fwd') for r in crit_recs]) * 1e-3
            out['roofline']['dominant_kernel'] = {
                'kernel': 'c2m::fwd_ws_kernel (4 -> 4 Conv2D 5x5 layer on the matrix cores, wave-specialised: forward, masked forward, backward-data; fp32 maps, '
                          'fp32 arithmetic as 6 bf16 products), {} launches per critic step'.format(len(c2m)),
                'achieved': by / tt / 1e9, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'frac': by / tt / 1e9 / PEAK_HBM_GBPS, 'traffic': None,
                'algorithmic_bytes_per_launch': by / len(c2m), 'launch_ms': tt / len(c2m) * 1e3, 'launches_per_critic_step': len(c2m)}
    elif 'roofline_conv1d' in out:
        out['roofline'] = dict(out['roofline_conv1d'])
    # ---- `roofline_lstm`: by time in the LOOP the BLSTM's recurrence is the top kernel (2 x T step launches per generator step on the
    # side stream): one step = h_{t-1} [B x H] . U [H x 4H] per direction on the fp32 matrix pipe -- latency-bound by construction
    # (T sequential launches), priced against the fp32 MFMA peak so that the line says how far
    H = int(opt.cfg.arch_hiddenwidth)
    try:
        for _ in range(2):
            opt.generator_step(X, Y)
        torch.cuda.synchronize()
        lrecs = []
        for _ in range(3):
            with _hip.KernelTimer() as kt:
                opt.generator_step(X, Y)
            lrecs.append([(nm, d) for (nm, _, d) in kt.durations_ms() if nm in ('ptts_lstm_fwd', 'ptts_lstm_bwd')])
        fw = [d for r in lrecs for (nm, d) in r if nm == 'ptts_lstm_fwd']; bw = [d for r in lrecs for (nm, d) in r if nm == 'ptts_lstm_bwd']
        if fw and bw:
            t_f, t_b = sum(fw) / len(fw) * 1e-3, sum(bw) / len(bw) * 1e-3
            fl_step = 2.0 * B * H * 4 * H * 2            # both directions
            out['roofline_lstm'] = {'bound': 'mfma', 'kernel': 'lstm_fwd_step_pk_kernel / lstm_bwd_step_pk_kernel: {} + {} launches per generator step (one per time step, both directions each)'.format(T, T),
                                    'achieved': fl_step * T * 2 / (t_f + t_b) / 1e12, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                    'frac': fl_step * T * 2 / (t_f + t_b) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 'traffic': None,
                                    'flop_per_launch': fl_step, 'fwd_us_per_step': t_f / T * 1e6, 'bwd_us_per_step': t_b / T * 1e6,
                                    'fwd_chain_ms': t_f * 1e3, 'bwd_chain_ms': t_b * 1e3,
                                    'note': 'latency-bound: T sequential launches; timed alone on one stream (inside the loop the chain shares the chip with the critic step)'}
    except Exception as e:      # the leg must not take the headline down
        out['roofline_lstm'] = {'error': repr(e)}
    # HBM(+Infinity-Cache) bytes per launch from the separate rocprofv3 PMC passes of the same kernels
    # (tools/profile_round.sh -> tools/summarize_profiles.py -> profiles/<round>_traffic.json; FETCH_SIZE x2 + WRITE_SIZE)
    try:
        cands = sorted(f for f in os.listdir(os.path.join(ROOT, 'profiles')) if f.endswith('_traffic.json'))
        if cands and B == 64 and T == 400 and X.shape[2] == 601:
            tr = json.load(open(os.path.join(ROOT, 'profiles', cands[-1])))
            key = next((k for k in ('gemm_dma_kernel<0, 1>', 'gemm_dma_kernel<0>', 'gemm_f32_mfma_kernel<0, 0, 1, 0>') if k in tr), None)
            if conv1d_split:
                kname = 'gemm_bf16x1_kernel' if bool(getattr(opt.cfg, 'train_wgan_bf16_products', False)) else 'gemm_bf16x6_kernel'
                key = kname if kname in tr else None
            if 'roofline_conv1d' in out and key and (conv1d_split or conv1d_fwd) and not conv1d_freq:
                out['roofline_conv1d']['traffic'] = tr[key]['hbm_bytes_per_launch']
                out['roofline_conv1d']['traffic_source'] = 'profiles/' + cands[-1]
            ck = next((k for k in tr if k.startswith('c2m::fwd_ws_kernel<1, false')), None) or next((k for k in tr if k.startswith('c2m::fwd_kernel<1, 1, false')), None) or next((k for k in tr if k.startswith('conv2d_fwd_kernel<4, 4')), None)
            if 'roofline_conv2d' in out and ck and not bf16_stack:
                out['roofline_conv2d']['traffic_fwd_4to4_per_launch'] = tr[ck]['hbm_bytes_per_launch']
                out['roofline_conv2d']['traffic_kernel'] = ck
                dk = out.get('roofline', {}).get('dominant_kernel')
                if dk is not None:
                    # NOT measured by this run: the builder's separate rocprofv3 --pmc passes of the same kernel (a counter pass cannot run
                    # inside a plain bench); the forward variant of the kernel, [64,400,65,4] in and out
                    dk['traffic'] = tr[ck]['hbm_bytes_per_launch']
                    dk['traffic_source'] = 'profiles/' + cands[-1] + ' (' + ck + "): builder's rocprofv3 PMC run, copied -- not a measurement of this run"
                    stk = tr.get('critic_conv2d_stack_per_critic_step')
                    if stk:
                        out['roofline']['traffic'] = stk['hbm_bytes']
                        out['roofline']['traffic_source'] = 'profiles/' + cands[-1] + " (sum over the stack's kernels x their launches per critic step): builder's rocprofv3 PMC run, copied -- not a measurement of this run"
    except (OSError, ValueError, KeyError):
        pass
    return out


def build_optimizer(args, ctx, spec, nm, batch, errtype, gated=False, bf16=None, graph=None):
    """Generator + critic + optimiser at one geometry (random-init weights of the named architecture)."""
    import io, contextlib
    from percivaltts_amd import vocoders, modeltts_common, networks_critic, optimizertts_wgan
    cfg = make_cfg(args)
    cfg.train_batch_size = batch
    cfg.arch_critic_bf16 = (args.dtype == 'bf16') if bf16 is None else bool(bf16)
    cfg.train_wgan_bf16_products = bool(cfg.arch_critic_bf16)      # configs[2]: bf16 products in the GEMM-shaped layers as well
    if gated:       # BASELINE configs[4]: pGCNN2D spectral branch, time dilations 1,2,4,8,1,2,4,8, causal padding
        cfg.arch_gen_gated = True; cfg.arch_gen_dilations = [1, 2, 4, 8]; cfg.arch_gen_causal = True
    # (more than one rank: 'tune' as well since round 4 -- the replayed form is then the split graph, forward + backward captured, the
    # gradient all-reduce and Adam launched eagerly behind it; PTTS_DP_GRAPH=0 forces eager launches on every rank)
    multi = int(os.environ.get('WORLD_SIZE', '1')) > 1
    cfg.train_wgan_hipgraph = ((True if args.graph else False if args.no_graph else 'tune') if graph is None else graph) if (not multi or os.environ.get('PTTS_DP_GRAPH', '1') == '1') else False
    if graph is None:
        cfg.train_wgan_graph_critic = getattr(args, 'graph_critic', None)
        cfg.train_wgan_graph_generator = getattr(args, 'graph_generator', None)
    cfg.train_wgan_prune_dead_branches = not args.no_prune
    cfg.train_wgan_graph_streams = os.environ.get('PTTS_GRAPH_STREAMS', '0') == '1'      # (experiment) fork / join inside the capture
    cfg.train_wgan_parallel_streams = (not args.no_streams) and (cfg.train_wgan_hipgraph is not True or cfg.train_wgan_graph_streams)
    cfg.train_wgan_stack_real_fake = not args.no_stack
    cfg.train_wgan_reuse_ctx_conv = not args.no_ctx_reuse
    cfg.train_wgan_early_critic = not args.no_early_critic
    cfg.train_wgan_hoist_generator = (not args.no_hoist) and os.environ.get('PTTS_HOIST', '1') == '1'
    cfg.train_wgan_generator_lookahead = (not getattr(args, 'no_lookahead', False)) and os.environ.get('PTTS_LOOKAHEAD', '1') == '1'
    cfg.train_wgan_graph_frozen_planes = os.environ.get('PTTS_FROZEN_PLANES', '1') == '1'          # (A/B switch)
    cfg.train_wgan_fake_ahead = os.environ.get('PTTS_FAKE_AHEAD', '0') == '1'          # (A/B switch)
    cfg.train_wgan_hoist_side_backward = os.environ.get('PTTS_HOIST_SIDE_BWD', '1') == '1'      # (A/B switch)
    cfg.train_wgan_side_backward_first = os.environ.get('PTTS_SIDE_BWD_FIRST', '0') == '1'     # (A/B switch) the BLSTM's autograd node created last
    cfg.train_wgan_split_bf16 = not args.fp32_mfma
    cfg.train_wgan_ctx_stream = os.environ.get('PTTS_CTX_STREAM', '0') == '1'          # (A/B switch) the critic's context branch on a side stream
    cfg.train_sync_batchnorm = bool(getattr(args, 'sync_bn', False))
    voc = vocoders.VocoderPML(16000, 0.005, spec, nm)
    with contextlib.redirect_stdout(io.StringIO()):
        mod = modeltts_common.DCNNF0SpecNoiseFeatures(ctx, voc, cfg)
        crit = networks_critic.Critic(voc, ctx, cfg)
        opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype=errtype, critic=crit)
        opt.prepare()
    opt.generator_updates = 26           # steady state: critic_runs = 5 (optimizertts_wgan.py:225-228)
    return cfg, voc, mod, crit, opt


def percentiles(xs):
    xs = sorted(xs)
    if not xs:
        return None
    q = lambda f: xs[min(len(xs) - 1, int(round(f * (len(xs) - 1))))]
    return {'median': q(0.5), 'p10': q(0.1), 'p90': q(0.9), 'n': len(xs)}


def timed_loop(opt, batches, nsteps, warmup, dev, cycle=5):
    """W untimed steps, then EXACTLY `nsteps` steps between barrier + synchronize on both sides (wall clock, max over
    ranks).  Inside the timed region a HIP event is recorded on the launch stream after every `cycle` steps (one
    5 critic : 1 generator cycle of the steady-state schedule): their spacing gives the per-cycle distribution."""
    import torch
    from percivaltts_amd import parallel
    nbuf = len(batches)

    def run(n, start, events=None):
        for i in range(n):
            X, Y = batches[(start + i) % nbuf]
            # the next batch is named (as the training driver does from its prefetcher, optimizertts.train_oneparamset): a generator step's
            # forward is launched one batch ahead.  Not on the last step of a run: the timed region holds the work of its own steps only
            opt.device_step(start + i, X, Y, nxt=batches[(start + i + 1) % nbuf] if i + 1 < n else None)
            if events is not None and (i + 1) % cycle == 0:
                ev = torch.cuda.Event(enable_timing=True); ev.record(); events.append(ev)

    # Priming (part of the set-up, like the tuning runs of 'tune' before it; not counted as warm-up): two cycles of the steady-state schedule
    # with the look-ahead, so that whatever happens ONCE -- the allocator's growth to the look-ahead's working set, the first launch of a
    # kernel variant, the capture of the graph that takes its fake sample as an input -- has happened before a short warm-up (the
    # driver's 5 steps) hands over to the timed region.  A look-ahead left pending by it is dropped: the timed region, and the warm-up,
    # contain the work of their own steps only.
    def drop_pending():
        ahead = getattr(opt, '_ahead', None)
        if ahead is not None:
            opt._ahead = None
            opt._drop_ahead(ahead)
    for i in range(2 * cycle):
        X, Y = batches[i % nbuf]
        opt.device_step(i, X, Y, nxt=batches[(i + 1) % nbuf])
    drop_pending()
    run(warmup, 0)
    drop_pending()
    parallel.barrier(); torch.cuda.synchronize()
    events = []
    e0 = torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    e0.record()
    run(nsteps, 0, events)
    torch.cuda.synchronize(); parallel.barrier()
    dt = parallel.max_over_ranks(time.time() - t0, dev)
    marks = [e0] + events
    cyc = [marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1)]
    return dt, percentiles(cyc)


def main():
    args = parse()
    world_env = int(os.environ.get('WORLD_SIZE', '0'))
    if args.gpus > 1 and world_env == 0:
        # not under torchrun: start the ranks as child processes (nothing here has touched the GPU yet)
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(29500 + os.getpid() % 1000), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    from percivaltts_amd import parallel, backend_hip, ops

    world, rank = parallel.init()
    dev = backend_hip.device()
    if world > 1 and not args.side_legs:
        # the first real multi-GPU run should print its line quickly: no side legs (each builds optimisers of its own with their
        # own broadcasts / all-reduces) -- `--side-legs` brings them back
        args.no_variants = args.no_unreduced = args.no_host_leg = args.no_reference_shape = args.no_bf16_leg = args.no_gated_leg = True
    spec, nm = 65, 20
    cfg, voc, mod, crit, opt = build_optimizer(args, args.ctx, spec, nm, args.batch, args.errtype)
    par_streams = bool(cfg.train_wgan_parallel_streams)

    B, T = args.batch, args.frames
    nbuf = 3
    batches = [synthetic(B, T, args.ctx, voc.featuressize(), spec, 123 + 17 * rank + i, dev) for i in range(nbuf)]

    # ---- the headline loop ------------------------------------------------------------------------------------------
    # cfg.train_wgan_hipgraph = 'tune': the choice eager launches / hipGraph replay per step kind is made here, on the first batch
    # and before the warm-up (the training state is put back after the timing runs)
    hipgraph_choice = {'critic': bool(opt._use_graph(batches[0][0], 'critic', batches[0][1])),
                       'generator': bool(opt._use_graph(batches[0][0], 'generator', batches[0][1])),
                       'mode': cfg.train_wgan_hipgraph, 'pinned': {'critic': cfg.train_wgan_graph_critic, 'generator': cfg.train_wgan_graph_generator},
                       'form_with_more_than_one_rank': 'split graph (forward + backward captured; all-reduce + Adam eager behind it)' if world > 1 else None,
                       'tuning_ms': {k[0]: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in opt._graph_tuning.items()}}
    dt, cyc = timed_loop(opt, batches, args.steps, args.warmup, dev)
    # (a batch that trains both networks may run as ONE graph: decided on the first such batch, inside the warm-up)
    hipgraph_choice['batch_that_trains_both_as_one_graph'] = bool(opt._graph_choice.get(('batch', tuple(batches[0][0].shape), tuple(batches[0][1].shape)), False))
    hipgraph_choice['tuning_ms'] = {k[0]: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in opt._graph_tuning.items()}
    extra = {}
    if cyc:
        extra['cycle_ms'] = dict(cyc, what='HIP-event spacing of 5-step cycles (5 critic + 1 generator step) inside the timed region')

    short = max(6, min(args.steps, 60))      # the side loops: a bounded number of steps each
    # the same loop with the context Conv1D on the fp32 MFMA pipe (the variant kept selectable: --fp32-mfma)
    # (the variants change what a step launches: they run eagerly -- a captured graph would replay the headline's kernels)
    graph_mode = opt.cfg.train_wgan_hipgraph
    pins = (opt.cfg.train_wgan_graph_critic, opt.cfg.train_wgan_graph_generator)
    def unpinned_eager(on):
        opt.cfg.train_wgan_hipgraph = False if on else graph_mode
        opt.cfg.train_wgan_graph_critic, opt.cfg.train_wgan_graph_generator = (None, None) if on else pins
    if not args.fp32_mfma and not args.no_variants:
        opt.cfg.train_wgan_split_bf16 = False
        unpinned_eager(True)
        dtv, _ = timed_loop(opt, batches, short, 6, dev)
        unpinned_eager(False)
        opt.cfg.train_wgan_split_bf16 = True
        ops.conv1d_split(True)
        ops.dense_split(True)
        extra['variant_fp32_mfma_gemms'] = {
            'what': 'same loop, context Conv1D (forward, weight gradient) and Dense products on the fp32 matrix pipe, v_mfma_f32_32x32x2_f32 / 16x16x4_f32 (python bench.py --fp32-mfma)',
            'value': short * B * T * world / dtv, 'unit': 'frames/s', 'ms_per_step': dtv / short * 1e3, 'steps': short}
    # the same loop with every exact work reduction off: the TF graph's own amount and order of work
    if not args.no_unreduced and not (args.no_prune and args.no_stack and args.no_ctx_reuse and args.no_early_critic):
        saved = (opt._gen_spec, opt.cfg.train_wgan_stack_real_fake, opt.cfg.train_wgan_reuse_ctx_conv, opt.cfg.train_wgan_early_critic)
        hoist_saved = opt.cfg.train_wgan_hoist_generator
        opt._gen_spec = None
        opt.cfg.train_wgan_stack_real_fake = opt.cfg.train_wgan_reuse_ctx_conv = opt.cfg.train_wgan_early_critic = opt.cfg.train_wgan_hoist_generator = False
        unpinned_eager(True)
        dtu, _ = timed_loop(opt, batches, short, 6, dev)
        unpinned_eager(False)
        opt._gen_spec, opt.cfg.train_wgan_stack_real_fake, opt.cfg.train_wgan_reuse_ctx_conv, opt.cfg.train_wgan_early_critic = saved
        opt.cfg.train_wgan_hoist_generator = hoist_saved
        extra['all_exact_work_reductions_off'] = {
            'what': "--no-prune --no-stack --no-ctx-reuse --no-early-critic --no-hoist: G's f0/noise branches run in the critic step, critic(real) and "
                    'critic(fake) as two passes, the generator step recomputes its context Conv1D and waits for the BLSTM before the critic',
            'value': short * B * T * world / dtu, 'unit': 'frames/s', 'ms_per_step': dtu / short * 1e3, 'steps': short}

    # separate timings of the two step kinds (eager or graph as configured) and the roofline leg.  Every rank runs them
    # (the steps contain the gradient all-reduce: a collective only rank 0 entered would hang the job); rank 0 reports.
    X, Y = batches[0]
    def timeit(fn, n):
        fn(); torch.cuda.synchronize()
        t = time.time()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        return (time.time() - t) / n * 1e3
    graph_c, graph_g = bool(opt._use_graph(X, 'critic', Y)), bool(opt._use_graph(X, 'generator', Y))
    extra['critic_step_ms'] = timeit((lambda: opt._graphed('critic', X, Y)) if graph_c else (lambda: opt.critic_step(X, Y)), 10)
    extra['generator_step_ms'] = timeit((lambda: opt._graphed('generator', X, Y)) if graph_g else (lambda: opt.generator_step(X, Y)), 5)
    import ctypes
    from percivaltts_amd import _hip
    st3 = [ctypes.c_ulonglong(0) for _ in range(3)]
    _hip.lib().ptts_lstm_graph_stats(*[ctypes.byref(v) for v in st3])
    extra['lstm_graph'] = {'replays': st3[0].value, 'captures': st3[1].value, 'plain_launch_fallbacks': st3[2].value,
                           'what': "the BLSTM's 400-step recurrences (forward, backward) replayed as one hipGraph launch each"}
    if world > 1:
        # the two exchanges of a cycle, timed on their own (HIP events on the launch stream, max over ranks): all-reduce (sum) of
        # the flat fp32 gradient bucket of each network
        import torch.distributed as dist
        def ar_ms(t, n=10):
            parallel.allreduce_sum_(t); torch.cuda.synchronize(); parallel.barrier()
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            for _ in range(n): parallel.allreduce_sum_(t)
            e_.record(); torch.cuda.synchronize()
            return parallel.max_over_ranks(s_.elapsed_time(e_) / n, dev)
        gc, gg = torch.zeros_like(opt.critic_opti.flat.grad), torch.zeros_like(opt.gen_opti.flat.grad)
        extra['allreduce_ms'] = {'critic_grads': ar_ms(gc), 'generator_grads': ar_ms(gg), 'critic_bytes': gc.numel() * 4, 'generator_bytes': gg.numel() * 4,
                                 'backend': dist.get_backend(), 'what': 'all-reduce(sum) of one flat fp32 gradient bucket per network, max over ranks'}
        # what of the exchange + Adam is NOT hidden behind the next step: back-to-back steps with the update (on the communication
        # stream, overlapped) against the same steps without it (gradients only)
        def loop_ms(fn, n=10):
            fn(); opt.wait_updates(); torch.cuda.synchronize(); parallel.barrier()
            t = time.time()
            for _ in range(n): fn()
            opt.wait_updates(); torch.cuda.synchronize()
            return parallel.max_over_ranks((time.time() - t) / n * 1e3, dev)
        with_c = loop_ms((lambda: opt._graphed('critic', X, Y)) if graph_c else (lambda: opt.critic_step(X, Y)))
        saved_update = opt._update
        opt._update = lambda kind: None
        try:
            only_c = loop_ms((lambda: opt._graphed('critic', X, Y)) if graph_c else (lambda: opt._critic_grads(X, Y)))
        finally:
            opt._update = saved_update
        extra['allreduce_ms']['critic_update_exposed_ms_per_step'] = with_c - only_c
        extra['allreduce_ms']['critic_step_ms_with_update'] = with_c
        extra['allreduce_ms']['critic_step_ms_gradients_only'] = only_c
        extra['allreduce_ms']['update_form'] = 'asynchronous (communication stream: event -> all_reduce(async_op) -> work.wait() -> Adam)' if opt._async() else 'synchronous'
        extra['allreduce_ms']['critic_step_form'] = 'split hipGraph replay + eager update' if graph_c else 'eager launches'

    if not args.no_roofline:
        extra.update(roofline_leg(opt, X, Y, args))
        opt.cfg.train_wgan_parallel_streams = par_streams
        opt._model.kerasmodel.parallel_branches = par_streams
    if not args.no_host_leg:
        # PCIe-inclusive rate: the same steps fed from host numpy batches through the double-buffered prefetcher
        # (data.BatchPrefetcher: loader thread + copy stream), as the training driver feeds train_on_batch.  Never `value`.
        from percivaltts_amd import data
        pool = [(bx.cpu().numpy(), by.cpu().numpy()) for bx, by in batches]
        # (the first `fill` steps -- the loader thread's start, the pipeline's fill, the first look-ahead -- are not timed: a whole number of cycles)
        fill = 10
        nh = max(10, min(args.steps, 120)) // 5 * 5
        pf = data.BatchPrefetcher(lambda i: pool[i % nbuf], fill + nh, device=dev, depth=2)
        parallel.barrier(); torch.cuda.synchronize()
        th = time.time()
        from percivaltts_amd.optimizertts import _with_next
        for i, (hx, hy), nx in _with_next(pf):
            if i == fill:
                torch.cuda.synchronize(); parallel.barrier()
                th = time.time()
            opt.device_step(i, hx, hy, nxt=nx)
        torch.cuda.synchronize(); parallel.barrier()
        dth = parallel.max_over_ranks(time.time() - th, dev)
        pf.close()
        extra['pcie_inclusive'] = {'value': nh * B * T * world / dth, 'unit': 'frames/s', 'steps': nh,
                                   'ms_per_step': dth / nh * 1e3, 'host_bytes_per_step': int(sum(a.nbytes for a in pool[0])),
                                   'how': 'host numpy batches -> loader thread -> H2D on a copy stream, 2 batches ahead (data.BatchPrefetcher)'}
    if not args.no_reference_shape:
        # the reference's own training geometry (run.py:76,89,125-126): B=10 sentences of 400 frames, 425 context labels ->
        # 163 features (f0 1 + spec 129 + noise 33), same architecture; per rank
        rB, rT, rctx, rspec, rnm = 10, 400, 425, 129, 33
        # launch-bound at 4 000 frames per step: replayed as a hipGraph (cfg.train_wgan_hipgraph = 'auto': batches of <= 8192 frames)
        _, rvoc, _, _, ropt = build_optimizer(args, rctx, rspec, rnm, rB, args.errtype, graph='auto')
        rb = [synthetic(rB, rT, rctx, rvoc.featuressize(), rspec, 900 + 17 * rank + i, dev) for i in range(nbuf)]
        nr = max(10, min(args.steps, 100))
        dtr, rcyc = timed_loop(ropt, rb, nr, 10, dev)
        extra['reference_shape'] = {'workload': 'run.py geometry: [10,400,425] -> [10,400,163] (f0 1 + spec 129 + noise 33), per GPU',
                                    'value': nr * rB * rT * world / dtr, 'unit': 'frames/s', 'ms_per_step': dtr / nr * 1e3,
                                    'steps': nr, 'cycle_ms': rcyc, 'hipgraph': bool(ropt._use_graph(rb[0][0]))}
        del ropt, rb
    if args.dtype == 'f32' and not args.no_bf16_leg:
        # BASELINE configs[2]: same shapes, the critic's conv stack with bf16 maps (gradient penalty on, lambda = 10)
        _, bvoc, _, _, bopt = build_optimizer(args, args.ctx, spec, nm, B, args.errtype, bf16=True)
        nb = max(6, min(args.steps, 60))
        dtb, bcyc = timed_loop(bopt, batches, nb, 6, dev)
        leg = {'workload': 'BASELINE configs[2]: same shapes, bf16 maps between the 4->4 layers of the critic\'s Conv2D stack (and their gradients), '
                           'bf16 products with fp32 accumulation, fp32 master weights and weight gradients, gradient penalty lambda=10',
               'value': nb * B * T * world / dtb, 'unit': 'frames/s', 'ms_per_step': dtb / nb * 1e3, 'steps': nb, 'cycle_ms': bcyc, 'dtype': 'bf16'}
        if not args.no_roofline:
            rl = roofline_leg(bopt, batches[0][0], batches[0][1], args)
            if 'roofline_conv2d' in rl:
                leg['roofline_conv2d'] = rl['roofline_conv2d']
        extra['config2_bf16'] = leg
        del bopt
        torch.cuda.empty_cache()
    if not args.no_gated_leg:
        # BASELINE configs[4]: generator spectral branch from gated convolutions (networktts.py:128-134) with time dilations
        # 1,2,4,8,1,2,4,8 and causal padding, long context T = 2000; fp32, per GPU
        gB, gT = args.gated_batch, 2000
        _, gvoc, _, _, gopt = build_optimizer(args, args.ctx, spec, nm, gB, args.errtype, gated=True)
        gb = [synthetic(gB, gT, args.ctx, gvoc.featuressize(), spec, 500 + 17 * rank + i, dev) for i in range(2)]
        ng = 10
        dtg, gcyc = timed_loop(gopt, gb, ng, 5, dev)
        extra['config4_gated_dilated_causal_T2000'] = {
            'workload': 'BASELINE configs[4] shape per GPU: synthetic [{b},{t},{c}]->[{b},{t},{o}], generator spectral branch = 8 gated '
                        'Conv2D layers (dilations 1,2,4,8,1,2,4,8 in time, causal), 2D-conv critic, fp32'.format(b=gB, t=gT, c=args.ctx, o=gvoc.featuressize()),
            'value': ng * gB * gT * world / dtg, 'unit': 'frames/s', 'ms_per_step': dtg / ng * 1e3, 'steps': ng, 'cycle_ms': gcyc}
        del gopt, gb
        torch.cuda.empty_cache()
    parallel.barrier()
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        extra['cpu_baseline'] = cpu_baseline(args, (args.ctx, spec, nm))

    if rank == 0:
        frames = float(args.steps) * B * T * world
        res = {
            'metric': 'acoustic frames/sec per WGAN-GP critic+gen step', 'value': frames / dt, 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[{n}]: synthetic [{b},{t},{c}]->[{b},{t},{o}] (f0 1 + spec 65 + noise 20), DCNN generator + '
                                   '2D-conv critic, {d}, {e}, lambda=10, schedule 5 critic steps : 1 generator step; '
                                   'step = one train_on_batch'.format(n=2 if args.dtype == 'bf16' else 1, b=B, t=T, c=args.ctx, o=voc.featuressize(), e=args.errtype,
                                                                      d='bf16 (see config.bf16)' if args.dtype == 'bf16' else 'fp32'),
                       'per_gpu_batch': B, 'global_batch': B * world, 'frames_per_step_per_gpu': B * T,
                       'parallelism': 'dp{}'.format(world), 'hipgraph': hipgraph_choice,
                       'hip_streams': 3 if par_streams else 1,
                       'prune_dead_generator_branches_in_critic_step': bool(cfg.train_wgan_prune_dead_branches),
                       'stack_real_fake_critic_pass': bool(cfg.train_wgan_stack_real_fake),
                       'reuse_generator_ctx_conv_within_train_on_batch': bool(cfg.train_wgan_reuse_ctx_conv),
                       'generator_forward_hoisted_before_the_critic_step': bool(cfg.train_wgan_hoist_generator),
                       'fake_sample_one_batch_ahead_on_a_side_stream': bool(cfg.train_wgan_fake_ahead),
                       'generator_forward_one_batch_ahead': bool(cfg.train_wgan_hoist_generator and cfg.train_wgan_generator_lookahead),
                       'ctx_conv1d_forward_and_weight_gradient': (('frequency domain (DFT, per-frequency products, inverse DFT; correlation theorem for the weight gradient), every product a '
                                                                   if (ops._C1FFT.enabled and not cfg.train_wgan_bf16_products) else '') + 'bf16x6 split (bf16 MFMA, fp32 accumulate)'
                                                                  + (' -- ONE bf16 product (time domain)' if cfg.train_wgan_bf16_products else '')) if cfg.train_wgan_split_bf16 else 'fp32 MFMA',
                       'dense_products': ('forward / backward-data / weight gradients (>= {} frames, two stages): bf16x6 split (bf16 MFMA, fp32 accumulate); 1-wide heads: fp32'.format(2048)
                                          if (cfg.train_wgan_split_bf16 and ops._DenseSplit.enabled) else 'fp32 MFMA'),
                       'collective_backend': (__import__('torch').distributed.get_backend() if world > 1 else None),
                       'conv2d_stacks': ops.conv2d_path_description() if hasattr(ops, 'conv2d_path_description') else 'fp32 packed-FMA stencil',
                       'batchnorm_statistics': 'per rank (B={} each; SyncBN off)'.format(B) if not getattr(cfg, 'train_sync_batchnorm', False) else 'synchronised over ranks (SyncBN)',
                       'input_batches_rotating': nbuf,
                       'generator_params': mod.count_params(), 'critic_params': crit.model.count_params()},
        }
        res.update(extra)
        print(json.dumps(res))
    parallel.barrier()


if __name__ == '__main__':
    main()

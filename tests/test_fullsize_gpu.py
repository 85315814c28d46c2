"""Parity at BASELINE.json's full size (configs[1]: B=64, T=400, ctx=601, 86 outputs, H=256, 8 x Conv2D(4, 5x5)),
size-independent properties of the domain plus oracle checks on crops at B = 64, and (round 4, at the end of the file) the fp64
oracle against the whole networks at full width: predict, a critic step at B = 16 with every gradient, its loss parts at B = 64,
a generator step at B = 12.

 * the kernels at the real layer shapes against the fp64 oracle on crops (conv2d with its halo) / row samples (GEMM);
 * linearity of the convolution and of the GEMM at full size;
 * sample independence of the critic (no BatchNorm): a sub-batch gives the same values as the rows of the full batch;
 * critic(real) + critic(fake) stacked as one 2B pass == two separate passes, losses and gradients;
 * data parallelism by construction: the critic gradient of the batch is the mean of the gradients of its two halves
   (what the flat-bucket all-reduce averages), with the same interpolation weights.
fp32 kernels vs fp64 oracle: rtol 2e-4 (the north star's bar is 1e-3)."""
import io
import contextlib

import pytest
import torch

from oracle import percival_oracle as O

pytestmark = pytest.mark.gpu

B, T, CTX, SPEC, NM = 64, 400, 601, 65, 20


def close(got, want, rtol, atol, what):
    got = got.detach().double().cpu(); want = want.detach().double().cpu()
    assert got.shape == want.shape, '{}: {} vs {}'.format(what, tuple(got.shape), tuple(want.shape))
    err = (got - want).abs(); tol = atol + rtol * want.abs()
    if (err > tol).any():
        i = int(torch.argmax(err - tol))
        raise AssertionError('{}: {}/{} off, worst {:.3e} (got {:.6e}, want {:.6e})'.format(
            what, int((err > tol).sum()), err.numel(), float(err.flatten()[i]), float(got.flatten()[i]), float(want.flatten()[i])))


def rel_l2(a, b):
    a = a.detach().double(); b = b.detach().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


@pytest.fixture(scope='module')
def setup():
    import bench
    from percivaltts_amd import vocoders, modeltts_common, networks_critic, optimizertts_wgan, backend_hip

    class A: batch = B; frames = T; ctx = CTX
    cfg = bench.make_cfg(A)
    dev = backend_hip.device()
    voc = vocoders.VocoderPML(16000, 0.005, SPEC, NM)
    with contextlib.redirect_stdout(io.StringIO()):
        mod = modeltts_common.DCNNF0SpecNoiseFeatures(CTX, voc, cfg)
        crit = networks_critic.Critic(voc, CTX, cfg)
        opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype='WLSWGAN', critic=crit)
        opt.prepare()
    X, Y = bench.synthetic(B, T, CTX, voc.featuressize(), SPEC, 321, dev)
    return cfg, opt, crit, X, Y


def test_conv2d_full_size_against_oracle_crops_and_linearity():
    from percivaltts_amd import ops
    g = torch.Generator().manual_seed(1)
    x1 = torch.randn(B, T, SPEC, 4, generator=g).cuda()
    x2 = torch.randn(B, T, SPEC, 4, generator=g).cuda()
    w = (torch.randn(5, 5, 4, 4, generator=g) * 0.2).cuda()
    b = torch.randn(4, generator=g).cuda()
    y1 = ops.conv2d(ops.Lazy(x1, lrelu=True), w, b)
    # oracle on time crops with the halo (2 frames each side); first, middle (tile borders) and last rows
    for (bi, t0, n) in ((0, 0, 9), (17, 185, 30), (63, T - 7, 7)):
        lo, hi = max(0, t0 - 2), min(T, t0 + n + 2)
        ref = O.conv2d_nhwc(O.lrelu(x1[bi:bi + 1, lo:hi].double().cpu()), w.double().cpu(), b.double().cpu())
        close(y1[bi:bi + 1, t0:t0 + n], ref[:, t0 - lo:t0 - lo + n], 2e-4, 2e-5, 'conv2d crop b={} t0={}'.format(bi, t0))
    # linearity (no transform, no bias): conv(2.5 x1 - x2) == 2.5 conv(x1) - conv(x2)
    ya, yb = ops.conv2d(x1, w, None), ops.conv2d(x2, w, None)
    yc = ops.conv2d(2.5 * x1 - x2, w, None)
    assert rel_l2(yc, 2.5 * ya - yb) < 2e-6
    # the backward of the same layer: dx against the oracle on a crop, dw against a chunked fp64 reduction of crops
    x1r = x1.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    dy = torch.randn(B, T, SPEC, 4, generator=g).cuda()
    br_ = b.clone().requires_grad_(True)
    ops.conv2d(ops.Lazy(x1r, lrelu=True), wr, br_).backward(dy)
    bi, t0, n = 31, 203, 12
    lo, hi = t0 - 4, t0 + n + 4
    xc = x1[bi:bi + 1, lo:hi].double().cpu().requires_grad_(True)
    yc = O.conv2d_nhwc(O.lrelu(xc), w.double().cpu(), b.double().cpu())
    # rows within 2 of the crop border see a truncated receptive field: feed dy only where it is complete
    dyc = torch.zeros_like(yc); dyc[:, 2:-2] = dy[bi:bi + 1, lo + 2:hi - 2].double().cpu()
    yc.backward(dyc)
    close(x1r.grad[bi:bi + 1, t0:t0 + n], xc.grad[:, 4:4 + n], 2e-4, 2e-5, 'conv2d dx crop')
    # dw / db over ALL 1.66 M pixels (the per-workgroup partial sums of 64 x 22 workgroups and their reduction): an fp64
    # reduction over batch chunks of the oracle's own backward
    dw64 = torch.zeros(5, 5, 4, 4, dtype=torch.float64); db64 = torch.zeros(4, dtype=torch.float64)
    for b0 in range(0, B, 8):
        wq = w.double().cpu().requires_grad_(True); bq = b.double().cpu().requires_grad_(True)
        O.conv2d_nhwc(O.lrelu(x1[b0:b0 + 8].double().cpu()), wq, bq).backward(dy[b0:b0 + 8].double().cpu())
        dw64 += wq.grad; db64 += bq.grad
    scale = float(dw64.abs().mean())
    assert float((wr.grad.double().cpu() - dw64).abs().max()) < 1e-4 * scale, (float((wr.grad.double().cpu() - dw64).abs().max()), scale)
    close(br_.grad, db64, 2e-4, 1e-4 * float(db64.abs().mean()), 'conv2d db, full size')


def test_dense_split_products_full_size():
    """The DEFAULT Dense path at BASELINE size (csrc/dense.hip: bf16x6 split products against a weight of a flat parameter
    buffer; the test below exercises the fp32-MFMA kernel, which plain tensors select): forward with LeakyReLU + bias and
    backward-data with the output mask on rows at the 112-row workgroup borders against the fp64 oracle, linearity, and the
    two-stage weight gradient + bias gradient of all 25 600 frames against an fp64 product."""
    from percivaltts_amd import ops, layers, _hip
    g = torch.Generator().manual_seed(21)
    M, K, N = B * T, 256, 256
    A1 = torch.randn(M, K, generator=g).cuda(); A2 = torch.randn(M, K, generator=g).cuda()
    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.randn(K, N, generator=g) / 16)
    h = Holder(); layers.FlatParams(h, 'cuda'); W = h.w
    bias = torch.randn(N, generator=g).cuda()
    C1 = torch.empty(M, N, device='cuda'); C2 = torch.empty(M, N, device='cuda'); C3 = torch.empty(M, N, device='cuda')
    with _hip.KernelTimer() as kt:
        ops.gemm_raw(A1, W, C1, M, N, K, bias=bias, mode=ops.IN_LRELU)
    assert [r[0] for r in kt.records] == ['ptts_split3_dense_weight', 'ptts_dense_bf16x6'], [r[0] for r in kt.records]
    rows = torch.tensor([0, 1, 111, 112, 113, 12799, 25487, 25599])      # tile borders of the 112-row workgroups
    W64 = W.detach().double().cpu()
    close(C1[rows], O.lrelu(A1[rows].double().cpu()) @ W64 + bias.double().cpu(), 2e-4, 2e-4, 'dense rows')
    ops.gemm_raw(A1, W, C1, M, N, K); ops.gemm_raw(A2, W, C2, M, N, K); ops.gemm_raw(0.5 * A1 + 3 * A2, W, C3, M, N, K)
    assert rel_l2(C3, 0.5 * C1 + 3 * C2) < 2e-6
    # backward data: dX = (dY . W^T) * lrelu'(x), the mask in the epilogue
    dY = torch.randn(M, N, generator=g).cuda(); Xm = torch.randn(M, K, generator=g).cuda()
    dX = torch.empty(M, K, device='cuda')
    ops.gemm_raw(dY, W, dX, M, K, N, transB=1, ldb=N, alpha=0.3, out_mask=Xm)
    want = (dY[rows].double().cpu() @ W64.t()) * torch.where(Xm[rows].double().cpu() > 0, 1.0, 0.3)
    close(dX[rows], want, 2e-4, 2e-4, 'dense backward-data rows')
    # weight gradient over all frames (two stages: partial tiles, grouped reduce) and the bias gradient
    dW = torch.empty(K, N, device='cuda'); db = torch.empty(N, device='cuda')
    with _hip.KernelTimer() as kt:
        ops.gemm_raw(A1, dY, dW, K, N, M, transA=1, lda=K, rows_per_seg=M, mode=ops.IN_LRELU, alpha=0.3, colsum_b=db)
    assert 'ptts_dense_wgrad_bf16x6' in [r[0] for r in kt.records]
    ref = O.lrelu(A1.double().cpu()).t() @ dY.double().cpu()
    assert float((dW.double().cpu() - ref).abs().max()) < 3e-5 * float(ref.abs().mean()), float((dW.double().cpu() - ref).abs().max() / ref.abs().mean())
    close(db, dY.double().cpu().sum(0), 2e-4, 2e-3, 'dense db, full size')


@pytest.mark.parametrize('fwd', ['frequency', 'time'])
def test_context_conv1d_split_products_full_size(fwd):
    """The context Conv1D at BASELINE size (M = 25 600 frames, K = 21 x 601, N = 256; ops.conv1d on a weight of a flat parameter
    buffer) with direct oracle contact, for both forward paths: 'frequency' = the default since round 3 (ops._C1FFT: DFT, per-frequency
    products, inverse DFT as batched bf16x6 split products), 'time' = the bf16x6 split kernel of csrc/split.hip (conv1d_fft(False)).
    The forward on frames at utterance borders, at the 128-frame tile borders and at the stream-K segment borders (every output
    column), and the frame-major weight gradient wgrad_bf16x6_kernel<21> -- rows of dW over all 25 600 frames -- plus the bias
    gradient, against fp64 products of the fp32 operands.  Reference: networktts.py:116-120 (kl.Conv1D)."""
    from percivaltts_amd import ops, layers, _hip
    ops.conv1d_fft(fwd == 'frequency')
    try:
        _conv1d_full_size(fwd)
    finally:
        ops.conv1d_fft(None)


def _conv1d_full_size(fwd):
    from percivaltts_amd import ops, layers, _hip
    g = torch.Generator().manual_seed(31)
    KW, N = 21, 256
    x = torch.randn(B, T, CTX, generator=g).cuda()

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.randn(KW, CTX, N, generator=g) * 0.01)
            self.b = torch.nn.Parameter(torch.randn(N, generator=g) * 0.1)
    h = Holder(); layers.FlatParams(h, 'cuda'); w, b = h.w, h.b
    with _hip.KernelTimer() as kt:
        y = ops.conv1d(x, w, b)
    names = [r[0] for r in kt.records]
    if fwd == 'time':
        assert 'ptts_conv1d_bf16x6' in names and 'ptts_gemm' not in names, names
    else:
        assert names.count('ptts_dense_bf16x6_batched') == 3 and 'ptts_conv1d_bf16x6' not in names, names
    w64 = w.detach().double().cpu().reshape(KW * CTX, N); b64 = b.detach().double().cpu()
    xp = torch.zeros(B, T + KW - 1, CTX, dtype=torch.float64)
    xp[:, KW // 2:KW // 2 + T] = x.double().cpu()
    # frames: first / last of an utterance (zero padding), neighbours of the 128-frame tile borders (frame 25 599 = tile 199),
    # an utterance border inside a tile (399 | 400), the middle
    frames = [(0, 0), (0, 9), (0, 10), (0, 127), (0, 128), (0, 399), (1, 0), (1, 1), (17, 200), (31, 255), (31, 256), (63, 389), (63, 399)]
    scale = float(y.abs().mean())
    for (bi, t) in frames:
        want = xp[bi, t:t + KW].reshape(-1) @ w64 + b64
        err = float((y[bi, t].detach().double().cpu() - want).abs().max())
        assert err < 1e-4 * scale + 2e-5 * float(want.abs().max()), 'conv1d frame ({}, {}): {:.3e} (scale {:.3e})'.format(bi, t, err, scale)
    # weight gradient and bias gradient of all frames
    dy = torch.randn(B, T, N, generator=g).cuda()
    with _hip.KernelTimer() as kt:
        y.backward(dy)
    names = [r[0] for r in kt.records]
    assert ('ptts_conv1d_wgrad_bf16x6' if fwd == 'time' else 'ptts_conv1d_freq_wgrad_inverse') in names, names
    dy64 = dy.double().cpu().reshape(B * T, N)
    gw = w.grad.detach().double().cpu()
    gscale = float(gw.abs().mean())
    for (kw, ci) in ((0, 0), (0, 600), (3, 31), (10, 300), (10, 32), (20, 0), (20, 600), (7, 575), (7, 576)):      # taps at both ends, channel-block borders
        col = xp[:, kw:kw + T, ci].reshape(-1)
        want = col @ dy64
        err = float((gw[kw, ci] - want).abs().max())
        assert err < 2e-4 * gscale + 2e-5 * float(want.abs().max()), 'conv1d dW[{}, {}]: {:.3e} (scale {:.3e})'.format(kw, ci, err, gscale)
    close(b.grad, dy64.sum(0), 2e-4, 2e-3, 'conv1d db, full size')


def test_gemm_full_size_rows_and_linearity():
    from percivaltts_amd import ops
    g = torch.Generator().manual_seed(2)
    M, K, N = B * T, 256, 256
    A1 = torch.randn(M, K, generator=g).cuda(); A2 = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(K, N, generator=g) / 16).cuda(); bias = torch.randn(N, generator=g).cuda()
    C1 = torch.empty(M, N, device='cuda'); C2 = torch.empty(M, N, device='cuda'); C3 = torch.empty(M, N, device='cuda')
    ops.gemm_raw(A1, W, C1, M, N, K, bias=bias, mode=ops.IN_LRELU)
    rows = torch.tensor([0, 1, 111, 112, 113, 12799, 25487, 25599])      # tile borders of the 112-row workgroups
    want = O.lrelu(A1[rows].double().cpu()) @ W.double().cpu() + bias.double().cpu()
    close(C1[rows], want, 2e-4, 2e-4, 'dense rows')
    ops.gemm_raw(A1, W, C1, M, N, K); ops.gemm_raw(A2, W, C2, M, N, K); ops.gemm_raw(0.5 * A1 + 3 * A2, W, C3, M, N, K)
    assert rel_l2(C3, 0.5 * C1 + 3 * C2) < 2e-6
    # the context Conv1D as implicit GEMM over the zero-padded frame buffer, a few frames against the oracle
    KW = 21
    xp = torch.zeros(B, T + KW - 1, CTX, device='cuda'); x = torch.randn(B, T, CTX, generator=g).cuda()
    xp[:, KW // 2:KW // 2 + T] = x
    wc = (torch.randn(KW, CTX, N, generator=g) * 0.01).cuda()
    y = torch.empty(B, T, N, device='cuda')
    ops.gemm_raw(xp, wc, y, M, N, KW * CTX, lda=CTX, rows_per_seg=T, seg_stride=(T + KW - 1) * CTX, bias=bias)
    for (bi, t) in ((0, 0), (5, 9), (40, 399), (63, 200)):
        win = xp[bi, t:t + KW].double().cpu().reshape(-1)
        close(y[bi, t], win @ wc.double().cpu().reshape(-1, N) + bias.double().cpu(), 2e-4, 5e-4, 'conv1d frame')
    # weight gradient of the same product (deep K, split over workgroups): a sample of entries against fp64
    dyc = torch.randn(B, T, N, generator=g).cuda()
    dw = torch.empty(KW, CTX, N, device='cuda')
    ops.gemm_raw(xp, dyc, dw, KW * CTX, N, M, transA=1, lda=CTX, rows_per_seg=T, seg_stride=(T + KW - 1) * CTX)
    for (kw, ci) in ((0, 0), (10, 300), (20, 600)):
        col = xp[:, kw:kw + T, ci].double().cpu().reshape(-1)
        close(dw[kw, ci], col @ dyc.double().cpu().reshape(M, N), 2e-4, 2e-2, 'conv1d dW row')


def test_critic_rows_do_not_depend_on_the_batch(setup):
    cfg, opt, crit, X, Y = setup
    with torch.no_grad():
        full = crit.model(Y, X, training=False)
        sub = crit.model(Y[8:16].contiguous(), X[8:16].contiguous(), training=False)
    assert rel_l2(full[8:16], sub) < 5e-6


def _critic_grads(opt, X, Y, alpha, fake):
    opt.critic_opti.zero_grad()
    total, parts = opt.critic_loss(X, Y, alpha, training=True, fake=fake)
    total.backward()
    torch.cuda.synchronize()
    return total.detach().clone(), [p.detach().clone() for p in parts], opt.critic_opti.flat.grad.detach().clone()


def test_stacked_real_fake_equals_separate_passes(setup):
    cfg, opt, crit, X, Y = setup
    g = torch.Generator().manual_seed(3)
    alpha = torch.rand(B, generator=g).cuda()
    with torch.no_grad():
        fake = opt._fake_sample(X, True).detach()
    opt.cfg.train_wgan_stack_real_fake = True       # (the optimiser holds its own merged configuration object)
    t1, p1, g1 = _critic_grads(opt, X, Y, alpha, fake)
    opt.cfg.train_wgan_stack_real_fake = False
    t2, p2, g2 = _critic_grads(opt, X, Y, alpha, fake)
    opt.cfg.train_wgan_stack_real_fake = True
    for a, b, nm in zip(p1, p2, ('valid', 'fake', 'gp')):
        close(a, b, 1e-5, 1e-6, 'stacked vs separate: ' + nm)
    assert rel_l2(g1, g2) < 2e-5


def test_critic_fed_at_its_spectral_slice_equals_whole_samples(setup):
    """cfg.train_wgan_feed_spectra (default on, round 4): the critic is fed at its slice node -- real / fake spectra in the two halves of
    one [2B,T,spec] tensor, x^ interpolated between them, the two Wasserstein terms off the one stacked output, the shared context
    product added inside the first post-concat product's store -- against the reference's own data flow (86-column samples through
    the slice, networks_critic.py:57-59; concatenation, slice copies and the broadcast add as torch ops): same losses, same gradients
    (the penalty's norm over [T,spec] is the norm over [T,out]: the critic's gradient is zero in the columns it does not read)."""
    cfg, opt, crit, X, Y = setup
    g = torch.Generator().manual_seed(13)
    alpha = torch.rand(B, generator=g).cuda()
    with torch.no_grad():
        fake = opt._fake_sample(X, True).detach()
    assert fake.shape[-1] == SPEC
    res = []
    for feed in (True, False):
        opt.cfg.train_wgan_feed_spectra = feed
        res.append(_critic_grads(opt, X, Y, alpha, fake))
    opt.cfg.train_wgan_feed_spectra = True
    (t1, p1, g1), (t2, p2, g2) = res
    for a, b, nm in zip(p1, p2, ('valid', 'fake', 'gp')):
        close(a, b, 1e-5, 1e-6, 'fed at the slice vs whole samples: ' + nm)
    close(t1, t2, 1e-5, 1e-6, 'total')
    assert rel_l2(g1, g2) < 2e-5


def test_one_forward_launch_for_both_evaluations_equals_two(setup):
    """cfg.train_wgan_pair_forward (default on, round 4): the stacked real / fake batch (2B) and the interpolated sample (B) lie back to
    back in one buffer and walk the critic in lockstep -- a Conv2D / Dense layer's forward is ONE launch over 3B rows
    (ops.Conv2dPairFn / DensePairFn), its two outputs again back to back; the backward passes stay per evaluation.  Same kernels on
    the same rows: losses and gradients equal to the run-to-run spread of the atomics; a third fewer forward launches (asserted)."""
    from percivaltts_amd import _hip
    cfg, opt, crit, X, Y = setup
    g = torch.Generator().manual_seed(14)
    alpha = torch.rand(B, generator=g).cuda()
    with torch.no_grad():
        fake = opt._fake_sample(X, True).detach()
    res, counts = [], []
    for pair in (True, False):
        opt.cfg.train_wgan_pair_forward = pair
        with _hip.KernelTimer() as kt:
            res.append(_critic_grads(opt, X, Y, alpha, fake))
        names = [r[0] for r in kt.records]
        counts.append((names.count('ptts_conv2d_mfma_fwd'), names.count('ptts_conv2d_fwd'), names.count('ptts_dense_bf16x6') + names.count('ptts_dense_bf16x6_res')))
    opt.cfg.train_wgan_pair_forward = True
    (t1, p1, g1), (t2, p2, g2) = res
    assert counts[0][0] == counts[1][0] - 7 and counts[0][1] == counts[1][1] - 1 and counts[0][2] == counts[1][2] - 3, counts
    for a, b, nm in zip(p1, p2, ('valid', 'fake', 'gp')):
        close(a, b, 1e-6, 1e-7, 'one launch vs two: ' + nm)      # (the loss reductions add with fp32 atomics: last bits vary from run to run)
    assert rel_l2(g1, g2) < 2e-5


def test_batch_gradient_is_the_mean_of_the_shard_gradients(setup):
    """What data parallelism relies on: with per-sample interpolation weights fixed, the critic loss is a mean over the
    samples (no BatchNorm in the critic), so grad(batch) = (grad(first half) + grad(second half)) / 2."""
    cfg, opt, crit, X, Y = setup
    g = torch.Generator().manual_seed(4)
    alpha = torch.rand(B, generator=g).cuda()
    with torch.no_grad():
        fake = opt._fake_sample(X, True).detach()     # BatchNorm of the GENERATOR sees the full batch here, once
    h = B // 2
    t, _, gfull = _critic_grads(opt, X, Y, alpha, fake)
    ta, _, ga = _critic_grads(opt, X[:h].contiguous(), Y[:h].contiguous(), alpha[:h].contiguous(), fake[:h].contiguous())
    tb, _, gb = _critic_grads(opt, X[h:].contiguous(), Y[h:].contiguous(), alpha[h:].contiguous(), fake[h:].contiguous())
    close(t, 0.5 * (ta + tb), 2e-5, 1e-6, 'loss of the batch vs mean of the shard losses')
    assert rel_l2(gfull, 0.5 * (ga + gb)) < 5e-5


def test_grouped_weight_gradients_equal_per_layer_products(setup):
    """ops.deferred_weight_grads(): the Dense layers' weight (and bias) gradients of a whole backward pass, queued and run
    as one grouped launch that accumulates into the .grad buffers, against the per-layer products autograd accumulates."""
    from percivaltts_amd import ops
    cfg, opt, crit, X, Y = setup
    g = torch.Generator().manual_seed(6)
    alpha = torch.rand(B, generator=g).cuda()
    with torch.no_grad():
        fake = opt._fake_sample(X, True).detach()
    t1, p1, g1 = _critic_grads(opt, X, Y, alpha, fake)
    opt.critic_opti.zero_grad()
    from percivaltts_amd import _hip
    with _hip.KernelTimer() as kt:
        with ops.deferred_weight_grads():
            total, parts = opt.critic_loss(X, Y, alpha, training=True, fake=fake)
            assert not ops._Deferred.items
            total.backward()
    assert not ops._Deferred.items
    grouped = [tag[0] for (name, tag, _) in kt.durations_ms() if name == 'ptts_gemm_wgrad_grouped']
    split = [1 for (name, tag, _) in kt.durations_ms() if name == 'ptts_dense_wgrad_bf16x6_partials']     # the split kernel: one launch per product, grouped reduce
    assert sum(grouped) + sum(split) >= 8             # the critic's Dense layers, first and second order
    assert not any(name == 'ptts_gemm' and tag[3] == 1 and tag[1] > 4 and tag[5] == 0 for (name, tag, _) in kt.durations_ms())
    torch.cuda.synchronize()
    g2 = opt.critic_opti.flat.grad.detach().clone()
    close(total, t1, 1e-6, 1e-7, 'loss')
    assert rel_l2(g2, g1) < 2e-5
    # generator step: BatchNorm-fused Dense layers (scale/shift in the product's load transform)
    cps = opt.critic_opti.flat.params
    for p in cps: p.requires_grad_(False)
    try:
        opt.gen_opti.zero_grad()
        tot, _ = opt.generator_loss(X, Y, training=True)
        tot.backward(); torch.cuda.synchronize()
        ga = opt.gen_opti.flat.grad.detach().clone()
        opt.gen_opti.zero_grad()
        with ops.deferred_weight_grads():
            tot2, _ = opt.generator_loss(X, Y, training=True)
            tot2.backward()
        torch.cuda.synchronize()
        gb = opt.gen_opti.flat.grad.detach().clone()
    finally:
        for p in cps: p.requires_grad_(True)
    # two runs of the SAME (immediate) generator step already differ by ~1e-4 relative L2: the split products combine their
    # partial tiles with fp32 atomics and a last-bit difference can move a LeakyReLU mask (tools/wg_check.py)
    assert rel_l2(gb, ga) < 5e-4


def test_generator_step_reuses_the_context_conv_of_the_critic_step(setup):
    """device_step on a batch that also trains the generator: with ops._C1Cache the generator step takes the generator's
    context-Conv1D product from the critic step's fake sample; weights after the step must equal the recomputing path."""
    from percivaltts_amd import ops, _hip
    cfg, opt, crit, X, Y = setup

    def snapshot():
        return [t.detach().clone() for t in (opt.critic_opti.flat.flat, opt.critic_opti.m, opt.critic_opti.v, opt.critic_opti.step_count,
                                             opt.gen_opti.flat.flat, opt.gen_opti.m, opt.gen_opti.v, opt.gen_opti.step_count)]

    def restore(snap):
        for dst, src in zip((opt.critic_opti.flat.flat, opt.critic_opti.m, opt.critic_opti.v, opt.critic_opti.step_count,
                             opt.gen_opti.flat.flat, opt.gen_opti.m, opt.gen_opti.v, opt.gen_opti.step_count), snap):
            dst.copy_(src)
        opt.critic_opti.flat.epoch += 1; opt.gen_opti.flat.epoch += 1

    bn_state = [(k, t.detach().clone()) for k, t in opt._model.kerasmodel.weights() if 'moving' in k]
    snap = snapshot()
    torch.manual_seed(7)
    results = []
    opt.cfg.train_wgan_hoist_generator = False      # (the plain order: with the generator's forward hoisted -- the default, tested below -- there is nothing left to reuse)
    for reuse in (True, False):
        restore(snap)
        for (k, t), (_, t0) in zip([(k, t) for k, t in opt._model.kerasmodel.weights() if 'moving' in k], bn_state):
            t.copy_(t0)
        opt.cfg.train_wgan_reuse_ctx_conv = reuse
        torch.manual_seed(7)                       # same interpolation weights
        with _hip.KernelTimer() as kt:
            opt.device_step(0, X, Y)               # batchid 0: critic step + generator step
        torch.cuda.synchronize()
        nconv = sum(1 for (name, tag, _) in kt.durations_ms()      # context-Conv1D forward products (any of the three forms)
                    if (name == 'ptts_gemm' and tag[5] == 1 and tag[3] == 0) or name == 'ptts_conv1d_bf16x6' or
                    (name == 'ptts_dense_bf16x6_batched' and tag[0] == 'freq'))
        results.append((opt.gen_opti.flat.grad.detach().clone(), opt.critic_opti.flat.grad.detach().clone(), nconv))
    opt.cfg.train_wgan_reuse_ctx_conv = True
    opt.cfg.train_wgan_hoist_generator = True
    restore(snap)
    assert results[0][2] == results[1][2] - 1, 'one context-Conv1D forward fewer with the cache: {} vs {}'.format(results[0][2], results[1][2])
    # the gradients the two Adam steps consumed (still in the flat buffers).  Two runs of the same step already differ by
    # ~1e-4 relative L2 in the generator gradient (fp32 atomics in the split products, tools/wg_check.py); the weights
    # themselves are a poor yardstick here: Adam's first step moves every weight by lr*sign(g)
    assert rel_l2(results[0][1], results[1][1]) < 1e-4          # the critic's gradient does not depend on the cache
    assert rel_l2(results[0][0], results[1][0]) < 5e-4          # the generator's gradient


def test_hoisted_generator_forward_gives_the_same_train_step(setup):
    """cfg.train_wgan_hoist_generator (default on): on a batch that trains both networks the generator's forward -- it does not depend
    on the critic -- is launched before the critic step (its BLSTM chain then runs under that step), and the critic step takes its fake
    sample from it.  Same arithmetic on the same operands: both losses, both gradients and the BatchNorm moving statistics as with
    the plain order critic step -> generator step (reference optimizertts_wgan.py:225-240), within the run-to-run spread of a step."""
    from percivaltts_amd import ops, _hip
    cfg, opt, crit, X, Y = setup
    state = (opt.critic_opti.flat.flat, opt.critic_opti.m, opt.critic_opti.v, opt.critic_opti.step_count,
             opt.gen_opti.flat.flat, opt.gen_opti.m, opt.gen_opti.v, opt.gen_opti.step_count)
    snap = [t.detach().clone() for t in state]
    moving = [t for k, t in opt._model.kerasmodel.weights() if 'moving' in k]
    moving0 = [t.detach().clone() for t in moving]
    results = []
    branches0 = opt._model.kerasmodel.parallel_branches
    try:
        # hoisted with the BLSTM branch on its side stream (then its BACKWARD pass is hoisted as well: the branch is cut out of the
        # tape and run on its own, the cut's gradient joins the main backward pass as an extra root), hoisted on one stream, plain
        for hoist, branches in ((True, True), (True, False), (False, False)):
            for dst, src in zip(state, snap):
                dst.copy_(src)
            for dst, src in zip(moving, moving0):
                dst.copy_(src)
            opt.critic_opti.flat.epoch += 1; opt.gen_opti.flat.epoch += 1
            opt.cfg.train_wgan_hoist_generator = hoist
            opt.cfg.train_wgan_hoist_side_backward = branches        # (needs the branch on its side stream)
            opt._model.kerasmodel.parallel_branches = branches
            torch.manual_seed(5)                       # same interpolation weights
            with _hip.KernelTimer() as kt:
                lc, lg = opt.device_step(0, X, Y)      # batchid 0: critic step + generator step
            torch.cuda.synchronize()
            names = [r[0] for r in kt.records]
            results.append((float(lc), float(lg), opt.critic_opti.flat.grad.detach().clone(), opt.gen_opti.flat.grad.detach().clone(),
                            [t.detach().clone() for t in moving], names.count('ptts_lstm_fwd'), names.count('ptts_conv2d_fwd')))
    finally:
        opt.cfg.train_wgan_hoist_generator = True
        opt.cfg.train_wgan_hoist_side_backward = True
        opt._model.kerasmodel.parallel_branches = branches0
        for dst, src in zip(state, snap):
            dst.copy_(src)
        for dst, src in zip(moving, moving0):
            dst.copy_(src)
        opt.critic_opti.flat.epoch += 1; opt.gen_opti.flat.epoch += 1
    (lc0, lg0, gc0, gg0, mv0, nl0, nc0) = results[2]
    for what, (lc1, lg1, gc1, gg1, mv1, nl1, nc1) in zip(('hoisted, branch backward hoisted too', 'hoisted'), results[:2]):
        assert nl1 == 1 and nl0 == 1                      # the BLSTM forward runs once either way ...
        assert nc1 < nc0, (what, nc1, nc0)                # ... and the hoisted form does not evaluate G's spectral branch a second time for the fake sample
        assert abs(lc1 - lc0) <= 1e-4 * max(1.0, abs(lc0)) and abs(lg1 - lg0) <= 1e-4 * max(1.0, abs(lg0)), (what, lc1, lc0, lg1, lg0)
        assert rel_l2(gc1, gc0) < 3e-4, (what, rel_l2(gc1, gc0))
        assert rel_l2(gg1, gg0) < 1e-3, (what, rel_l2(gg1, gg0))
        for a, b in zip(mv1, mv0):
            close(a, b, 1e-5, 1e-6, 'BatchNorm moving statistics after the step (' + what + ')')


def test_generator_forward_one_batch_ahead_gives_the_same_two_batches(setup):
    """cfg.train_wgan_generator_lookahead (default on) with the next batch named (device_step(nxt=...), as the training driver does from
    its prefetcher): the forward of a generator step -- and its BLSTM branch's backward -- is launched in front of the critic step of the
    batch BEFORE it.  The generator's weights are not touched in between (reference optimizertts_wgan.py:225-240: the critic alone trains
    on that batch; its fake sample is drawn with the BatchNorm moving averages frozen), so two consecutive batches give the same losses,
    gradients, weights and moving statistics as the plain order; a look-ahead made for a batch that does not come is dropped."""
    from percivaltts_amd import _hip
    cfg, opt, crit, X, Y = setup
    X2, Y2 = X.flip(0).contiguous(), Y.flip(0).contiguous()
    state = (opt.critic_opti.flat.flat, opt.critic_opti.m, opt.critic_opti.v, opt.critic_opti.step_count,
             opt.gen_opti.flat.flat, opt.gen_opti.m, opt.gen_opti.v, opt.gen_opti.step_count)
    snap = [t.detach().clone() for t in state]
    moving = [t for k, t in opt._model.kerasmodel.weights() if 'moving' in k]
    moving0 = [t.detach().clone() for t in moving]
    gu0 = opt.generator_updates

    def restore():
        opt.wait_updates()
        for dst, src in zip(state, snap):
            dst.copy_(src)
        for dst, src in zip(moving, moving0):
            dst.copy_(src)
        opt.critic_opti.flat.epoch += 1; opt.gen_opti.flat.epoch += 1
        opt.generator_updates = 26                          # steady state: every fifth batch trains the generator (:225-228)
        opt._ahead = None

    results = {}
    try:
        # batch 4 (critic only) on (X2, Y2), batch 5 (critic + generator) on (X, Y)
        for what, named in (('ahead', (X, Y)), ('plain', None), ('ahead of another batch', (X2, Y2))):
            restore()
            torch.manual_seed(5)
            with _hip.KernelTimer() as k4:
                lc4, lg4 = opt.device_step(4, X2, Y2, nxt=named)
            with _hip.KernelTimer() as k5:
                lc5, lg5 = opt.device_step(5, X, Y)
            opt.wait_updates(); torch.cuda.synchronize()
            assert lg4 is None and lg5 is not None
            results[what] = (float(lc4), float(lc5), float(lg5), opt.critic_opti.flat.grad.detach().clone(), opt.gen_opti.flat.grad.detach().clone(),
                             opt.critic_opti.flat.flat.detach().clone(), opt.gen_opti.flat.flat.detach().clone(), [t.detach().clone() for t in moving],
                             [r[0] for r in k4.records].count('ptts_lstm_fwd'), [r[0] for r in k5.records].count('ptts_lstm_fwd'))
    finally:
        restore()
        opt.generator_updates = gu0
    ref = results['plain']
    assert (ref[8], ref[9]) == (0, 1)
    assert (results['ahead'][8], results['ahead'][9]) == (1, 0)                          # the recurrence ran one batch early, and once
    assert (results['ahead of another batch'][8], results['ahead of another batch'][9]) == (1, 1)      # dropped and done again on the batch that came
    for what in ('ahead', 'ahead of another batch'):
        r = results[what]
        for i in range(3):
            assert abs(r[i] - ref[i]) <= 1e-4 * max(1.0, abs(ref[i])), (what, i, r[i], ref[i])
        assert rel_l2(r[3], ref[3]) < 3e-4, (what, rel_l2(r[3], ref[3]))
        assert rel_l2(r[4], ref[4]) < 1e-3, (what, rel_l2(r[4], ref[4]))
        # both networks' weights after the two batches (Adam's first steps are sign-like -- an element whose gradient is noise moves by
        # +- the learning rate either way -- so: as a norm over the network)
        assert rel_l2(r[5], ref[5]) < 1e-3, (what, rel_l2(r[5], ref[5]))
        assert rel_l2(r[6], ref[6]) < 1e-3, (what, rel_l2(r[6], ref[6]))
        for a, b in zip(r[7], ref[7]):
            close(a, b, 1e-5, 1e-6, 'BatchNorm moving statistics after two batches (' + what + ')')


def test_batchnorm_statistics_taken_from_the_convolutions_sums(setup):
    """ops._BNStats (default on): the generator's 4 -> 4 Conv2D layers and its Dense layers sum what they store, and the BatchNormalization
    behind each takes its batch moments from those sums (reference networktts.py:59-63, 122-126: Dense / Conv2D -> BatchNormalization ->
    LeakyReLU).  Same tensors, same moments up to the order of the additions: the generator step gives the same loss, gradients and moving
    statistics as with a statistics pass per layer, and those passes are gone."""
    from percivaltts_amd import ops, _hip
    cfg, opt, crit, X, Y = setup
    state = (opt.gen_opti.flat.flat, opt.gen_opti.m, opt.gen_opti.v, opt.gen_opti.step_count)
    snap = [t.detach().clone() for t in state]
    moving = [t for k, t in opt._model.kerasmodel.weights() if 'moving' in k]
    moving0 = [t.detach().clone() for t in moving]
    res = []
    try:
        for on in (True, False):
            opt.wait_updates()
            for dst, src in zip(state, snap):
                dst.copy_(src)
            for dst, src in zip(moving, moving0):
                dst.copy_(src)
            opt.gen_opti.flat.epoch += 1
            ops.conv_bn_stats(on)
            with _hip.KernelTimer() as kt:
                lg = opt.generator_step(X, Y)
            opt.wait_updates(); torch.cuda.synchronize()
            names = [r[0] for r in kt.records]
            res.append((float(lg), opt.gen_opti.flat.grad.detach().clone(), [t.detach().clone() for t in moving],
                        names.count('ptts_bn_finalize_partials'), names.count('ptts_bn_batch_stats'), names.count('ptts_conv2d_mfma_fwd_stats'),
                        names.count('ptts_dense_bf16x6_stats'), names.count('ptts_colstats'),
                        names.count('ptts_dense_bf16x6_bwd_affine'), names.count('ptts_affine_act_bwd')))
    finally:
        ops.conv_bn_stats(None)
        opt.wait_updates()
        for dst, src in zip(state, snap):
            dst.copy_(src)
        for dst, src in zip(moving, moving0):
            dst.copy_(src)
        opt.gen_opti.flat.epoch += 1
    (l1, g1, m1, nf1, nb1, nc1, nd1, ns1, na1, nab1), (l0, g0, m0, nf0, nb0, nc0, nd0, ns0, na0, nab0) = res
    assert (nf0, nc0, nd0, na0) == (0, 0, 0, 0) and nb0 >= 8
    # backward: the Dense layers behind a BatchNormalization + LeakyReLU put that input's mask, its scale and the affine's two gradient
    # sums into their backward-data product's store: so many ptts_affine_act_bwd passes less
    assert na1 >= 3 and nab1 == nab0 - na1, (na1, nab1, nab0)
    # seven convolutions and the Dense layers in front of a BatchNormalization sum their own outputs: so many statistics passes less
    assert nc1 >= 7 and nd1 >= 3 and nf1 == nc1 + nd1 and nb1 == nb0 - nc1 and ns1 == ns0 - nd1, (nf1, nc1, nd1, nb1, nb0, ns1, ns0)
    assert abs(l1 - l0) <= 1e-5 * max(1.0, abs(l0)), (l1, l0)
    assert rel_l2(g1, g0) < 1e-3, rel_l2(g1, g0)
    for a, b in zip(m1, m0):
        close(a, b, 1e-5, 1e-6, 'BatchNorm moving statistics')


def test_replayed_critic_step_follows_the_generators_updates(setup):
    """cfg.train_wgan_graph_frozen_planes (default on): the critic step's hipGraph does not rebuild the FROZEN generator's frequency-domain
    kernel planes in every replay -- it reads the buffer the capture's warm-up left, and ops._C1FFT.refresh_planes rebuilds that buffer
    before a replay when the generator's weights have changed.  A stale buffer would be a silently wrong fake sample: the replay is
    compared with the eager step at the captured weights, after the generator's weights were changed, and after they changed again."""
    from percivaltts_amd import ops
    cfg, opt, crit, X, Y = setup
    state = (opt.critic_opti.flat.flat, opt.critic_opti.m, opt.critic_opti.v, opt.critic_opti.step_count, opt.gen_opti.flat.flat)
    snap = [t.detach().clone() for t in state]
    moving = [t for k, t in opt._model.kerasmodel.weights() if 'moving' in k]
    moving0 = [t.detach().clone() for t in moving]
    alpha = torch.rand(X.shape[0], generator=torch.Generator().manual_seed(7)).cuda()

    def load(gen_factor):
        opt.wait_updates()
        for dst, src in zip(state, snap):
            dst.copy_(src)
        for dst, src in zip(moving, moving0):
            dst.copy_(src)
        if gen_factor != 1.0:
            # "a generator update": every weight of G moves, each its own way (a common factor on a kernel would be undone by the
            # BatchNormalization behind it -- tools/neg_check_frozen.py: the test must fail when the refresh is switched off)
            gpert = torch.Generator(device='cuda').manual_seed(int(gen_factor * 1000))
            w = opt.gen_opti.flat.flat
            w.add_(torch.randn(w.shape, generator=gpert, device='cuda') * (abs(gen_factor - 1.0) * float(w.std())))
        opt.critic_opti.flat.epoch += 1; opt.gen_opti.flat.epoch += 1

    graphs0 = dict(opt._graphs)
    try:
        load(1.0)
        opt._graphed('critic', X, Y, alpha)                      # capture (its warm-up steps move the critic)
        key = [k for k in opt._graphs if k not in graphs0 and k[0] == 'critic' and not k[-1]]
        assert len(key) == 1
        fz = opt._graph_frozen[key[0]]
        assert len(fz['items']) >= 1                             # the generator owns a context kernel whose planes the graph reads
        for factor in (1.0, 1.05, 0.97):
            load(factor)
            lc_e = float(opt.critic_step(X, Y, alpha)); opt.wait_updates()
            ge = opt.critic_opti.flat.grad.detach().clone()
            load(factor)
            assert fz['epoch'] != opt.gen_opti.flat.epoch        # (the weights were reloaded: a refresh is due)
            lc_g = float(opt._graphed('critic', X, Y, alpha)); opt.wait_updates(); torch.cuda.synchronize()
            gg = opt.critic_opti.flat.grad.detach().clone()
            assert abs(lc_g - lc_e) <= 1e-4 * max(1.0, abs(lc_e)), (factor, lc_g, lc_e)
            assert rel_l2(gg, ge) < 3e-4, (factor, rel_l2(gg, ge))
            assert fz['epoch'] == opt.gen_opti.flat.epoch        # current after the replay's refresh
        # a second graph for batches of another length takes other planes for the same kernel (another segment geometry): the first
        # graph's buffer is then no longer the cache's, and must still follow the generator
        X2, Y2 = X[:, :200].contiguous(), Y[:, :200].contiguous()
        load(1.0)
        opt._graphed('critic', X2, Y2, alpha)
        for (Xa, Ya) in ((X, Y), (X2, Y2)):
            load(1.03)
            lc_e = float(opt.critic_step(Xa, Ya, alpha)); opt.wait_updates()
            ge = opt.critic_opti.flat.grad.detach().clone()
            load(1.03)
            lc_g = float(opt._graphed('critic', Xa, Ya, alpha)); opt.wait_updates(); torch.cuda.synchronize()
            gg = opt.critic_opti.flat.grad.detach().clone()
            assert abs(lc_g - lc_e) <= 1e-4 * max(1.0, abs(lc_e)), (tuple(Xa.shape), lc_g, lc_e)
            assert rel_l2(gg, ge) < 3e-4, (tuple(Xa.shape), rel_l2(gg, ge))
    finally:
        load(1.0)
        for k in [k for k in opt._graphs if k not in graphs0]:
            del opt._graphs[k]


def test_train_step_with_the_bf16x6_context_conv_matches_the_fp32_one(setup):
    """device_step at BASELINE configs[1] sizes with the context-Conv1D forward as a bf16x6 split product in the time domain
    (cfg.train_wgan_split_bf16, csrc/split.hip) and in the frequency domain (ops._C1FFT, the default) against the same step on
    the fp32 MFMA kernel: same losses, and gradients within the run-to-run spread of the fp32 path itself (fp32 atomics: ~1e-4
    relative L2)."""
    from percivaltts_amd import ops, _hip
    cfg, opt, crit, X, Y = setup
    state = (opt.critic_opti.flat.flat, opt.critic_opti.m, opt.critic_opti.v, opt.critic_opti.step_count,
             opt.gen_opti.flat.flat, opt.gen_opti.m, opt.gen_opti.v, opt.gen_opti.step_count)
    snap = [t.detach().clone() for t in state]
    moving = [t for k, t in opt._model.kerasmodel.weights() if 'moving' in k]
    moving0 = [t.detach().clone() for t in moving]
    results = []
    try:
        for split, fft in ((True, False), (False, False), (True, True)):
            ops.conv1d_fft(fft)
            for dst, src in zip(state, snap):
                dst.copy_(src)
            for dst, src in zip(moving, moving0):
                dst.copy_(src)
            opt.critic_opti.flat.epoch += 1; opt.gen_opti.flat.epoch += 1
            opt.cfg.train_wgan_split_bf16 = split
            torch.manual_seed(11)                      # same interpolation weights
            with _hip.KernelTimer() as kt:
                lc, lg = opt.device_step(0, X, Y)      # batchid 0: critic step + generator step
            torch.cuda.synchronize()
            names = [r[0] for r in kt.records]
            results.append((float(lc), float(lg), opt.critic_opti.flat.grad.detach().clone(),
                            opt.gen_opti.flat.grad.detach().clone(), names.count('ptts_conv1d_bf16x6'),
                            sum(1 for r in kt.records if r[0] == 'ptts_dense_bf16x6_batched' and r[1] and r[1][0] == 'freq')))
    finally:
        opt.cfg.train_wgan_split_bf16 = None
        ops.conv1d_split(None)
        ops.conv1d_fft(None)
        for dst, src in zip(state, snap):
            dst.copy_(src)
        for dst, src in zip(moving, moving0):
            dst.copy_(src)
        opt.critic_opti.flat.epoch += 1; opt.gen_opti.flat.epoch += 1
    (lc1, lg1, gc1, gg1, n1, f1), (lc0, lg0, gc0, gg0, n0, f0), (lc2, lg2, gc2, gg2, n2, f2) = results
    assert n1 == 3 and n0 == 0 and f1 == 0 and f0 == 0, (n1, n0, f1, f0)       # G and D context convs in the critic step, D's again in the generator step
    assert n2 == 0 and f2 == 3, (n2, f2)         # ... the same three in the frequency domain (the default, round 3)
    for (lc, lg, gc, gg, what) in ((lc1, lg1, gc1, gg1, 'time-domain split'), (lc2, lg2, gc2, gg2, 'frequency domain')):
        assert abs(lc - lc0) <= 1e-4 * max(1.0, abs(lc0)) and abs(lg - lg0) <= 1e-4 * max(1.0, abs(lg0)), (what, lc, lc0, lg, lg0)
        assert rel_l2(gc, gc0) < 3e-4, (what, rel_l2(gc, gc0))
        assert rel_l2(gg, gg0) < 1e-3, (what, rel_l2(gg, gg0))


def test_deterministic_mode_is_bit_reproducible_at_full_size(setup):
    """ops.deterministic(True) (PTTS_DETERMINISTIC=1): no reduction over workgroups uses fp32 atomics -- stream-K tiles, the
    grouped / frame-major / bf16x6 weight gradients, the split context Conv1D and the grouped conv2d reduction give way to
    fixed-order forms.  Two runs of the same critic step and of the same generator step at BASELINE configs[1] size then
    agree BIT FOR BIT (losses and every gradient), and they agree with the default (atomic) paths to 1e-4 relative L2 --
    the spread the atomics' summation order leaves, now measured against a fixed reference instead of against itself."""
    from percivaltts_amd import ops
    cfg, opt, crit, X, Y = setup
    g = torch.Generator().manual_seed(8)
    alpha = torch.rand(B, generator=g).cuda()
    with torch.no_grad():
        fake = opt._fake_sample(X, True).detach()

    def gen_grads():
        cps = opt.critic_opti.flat.params
        for p in cps: p.requires_grad_(False)
        try:
            opt.gen_opti.zero_grad()
            with ops.deferred_weight_grads():
                tot, _ = opt.generator_loss(X, Y, training=True)
                tot.backward()
            torch.cuda.synchronize()
            return tot.detach().clone(), opt.gen_opti.flat.grad.detach().clone()
        finally:
            for p in cps: p.requires_grad_(True)

    moving = [t for k, t in opt._model.kerasmodel.weights() if 'moving' in k]
    moving0 = [t.detach().clone() for t in moving]
    def restore():
        for dst, src in zip(moving, moving0): dst.copy_(src)

    def critic_grads():
        opt.critic_opti.zero_grad()
        with ops.deferred_weight_grads():
            total, _ = opt.critic_loss(X, Y, alpha, training=True, fake=fake)
            total.backward()
        torch.cuda.synchronize()
        return total.detach().clone(), opt.critic_opti.flat.grad.detach().clone()

    try:
        t_def, g_def = critic_grads()
        restore(); lg_def, gg_def = gen_grads()
        ops.deterministic(True)
        t1, g1 = critic_grads(); t2, g2 = critic_grads()
        restore(); lg1, gg1 = gen_grads()
        restore(); lg2, gg2 = gen_grads()
    finally:
        ops.deterministic(False)
        restore()
    assert torch.equal(t1, t2) and torch.equal(g1, g2), 'critic step not reproducible: rel L2 {:.3e}'.format(rel_l2(g2, g1))
    assert torch.equal(lg1, lg2) and torch.equal(gg1, gg2), 'generator step not reproducible: rel L2 {:.3e}'.format(rel_l2(gg2, gg1))
    close(t_def, t1, 1e-5, 1e-6, 'critic loss, default vs deterministic paths')
    close(lg_def, lg1, 1e-5, 1e-6, 'generator loss, default vs deterministic paths')
    assert rel_l2(g_def, g1) < 1e-4, rel_l2(g_def, g1)
    assert rel_l2(gg_def, gg1) < 5e-4, rel_l2(gg_def, gg1)


def test_bf16_storage_critic_at_full_size():
    """BASELINE configs[2] at its real size: the critic with bf16 maps in its Conv2D stack (cfg.arch_critic_bf16) against the
    same critic in fp32 on the same weights and batch.  Size-independent properties: the losses agree within the bf16 budget
    (stored values off by <= 2^-9 each, 7 bf16 layers: well inside 2e-2), the gradient of the whole critic agrees to a few
    per cent in relative L2, sample independence holds (a sub-batch gives the rows of the batch: the tiling does not couple
    samples), and the weight gradients are fp32."""
    import bench
    from percivaltts_amd import vocoders, modeltts_common, networks_critic, optimizertts_wgan, backend_hip, ops

    class A: batch = B; frames = T; ctx = CTX
    dev = backend_hip.device()
    voc = vocoders.VocoderPML(16000, 0.005, SPEC, NM)
    res = {}
    weights = None
    for name, bf in (('f32', False), ('bf16', True)):
        cfg = bench.make_cfg(A)
        cfg.arch_critic_bf16 = bf
        with contextlib.redirect_stdout(io.StringIO()):
            mod = modeltts_common.DCNNF0SpecNoiseFeatures(CTX, voc, cfg)
            crit = networks_critic.Critic(voc, CTX, cfg)
            if weights is None:
                weights = (mod.kerasmodel.get_weights(), crit.model.get_weights())
            else:
                mod.kerasmodel.set_weights(weights[0]); crit.model.set_weights(weights[1])
            opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype='WLSWGAN', critic=crit)
            opt.prepare()
        X, Y = bench.synthetic(B, T, CTX, voc.featuressize(), SPEC, 321, dev)
        g = torch.Generator().manual_seed(3)
        alpha = torch.rand(B, generator=g).cuda()
        with torch.no_grad():
            fake = opt._fake_sample(X, True).detach()
        opt.critic_opti.zero_grad()
        with ops.deferred_weight_grads():
            total, parts = opt.critic_loss(X, Y, alpha, training=True, fake=fake)
            total.backward()
        torch.cuda.synchronize()
        with torch.no_grad():
            full = crit.model(Y, X, training=False)
            sub = crit.model(Y[8:16].contiguous(), X[8:16].contiguous(), training=False)
        res[name] = (total.detach().clone(), [p.detach().clone() for p in parts], opt.critic_opti.flat.grad.detach().clone(), full, sub)
        assert opt.critic_opti.flat.grad.dtype == torch.float32
        del opt, mod, crit
    (t32, p32, g32, f32_, s32), (t16, p16, g16, f16, s16) = res['f32'], res['bf16']
    for a_, b_, nm in zip(p16, p32, ('valid', 'fake', 'gp')):
        close(a_, b_, 2e-2, 2e-3, 'bf16 vs fp32 critic: ' + nm)
    assert rel_l2(g16, g32) < 6e-2, rel_l2(g16, g32)
    assert rel_l2(f16, f32_) < 2e-2, rel_l2(f16, f32_)
    assert rel_l2(f16[8:16], s16) < 5e-6          # same roundings whatever the batch around a sample


def test_gated_dilated_causal_generator_at_T2000():
    """BASELINE configs[4] at its real length: the generator's spectral branch from gated convolutions (pGCNN2D,
    networktts.py:128-134) with time dilations 1,2,4,8,1,2,4,8 and causal padding (build extensions), T = 2000.
      * a gated layer at every dilation against the fp64 oracle on crops with their causal halo;
      * causality of the spectral branch in inference mode: frames before t0 - 10 (the context Conv1D looks 10 frames
        ahead) do not depend on the labels from t0 on;
      * one critic step and one generator step run and give finite losses and gradients.
    Batch 64 per GPU, as BASELINE configs[4] and the bench leg run it (round 2 tested B = 8)."""
    import bench
    from percivaltts_amd import vocoders, modeltts_common, networks_critic, optimizertts_wgan, backend_hip, ops, layers as kl
    Bq, Tq = 64, 2000

    class A: batch = Bq; frames = Tq; ctx = CTX
    cfg = bench.make_cfg(A)
    cfg.arch_gen_gated = True; cfg.arch_gen_dilations = [1, 2, 4, 8]; cfg.arch_gen_causal = True
    dev = backend_hip.device()
    voc = vocoders.VocoderPML(16000, 0.005, SPEC, NM)
    with contextlib.redirect_stdout(io.StringIO()):
        mod = modeltts_common.DCNNF0SpecNoiseFeatures(CTX, voc, cfg)
        crit = networks_critic.Critic(voc, CTX, cfg)
        opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype='WLSWGAN', critic=crit)
        opt.prepare()
    assert sum(isinstance(l, kl.GatedMultiply) for l in mod.kerasmodel.layers_list) == 8
    X, Y = bench.synthetic(Bq, Tq, CTX, voc.featuressize(), SPEC, 77, dev)

    # ---- a gated layer at each dilation, crops against the oracle
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, Tq, SPEC, 4, generator=g).cuda()
    wa = (torch.randn(5, 5, 4, 4, generator=g) * 0.2).cuda(); wb = (torch.randn(5, 5, 4, 4, generator=g) * 0.2).cuda()
    for dil in (1, 2, 4, 8):
        ya = ops.conv2d(ops.Lazy(x, lrelu=True), wa, None, dil_t=dil, pad_mode=ops.PAD_CAUSAL)
        yb = ops.conv2d(ops.Lazy(x, lrelu=True), wb, None, dil_t=dil, pad_mode=ops.PAD_CAUSAL)
        y = ops.gated_mul(ya, yb)
        for (bi, t0, n) in ((0, 0, 20), (1, 1000, 40), (1, Tq - 16, 16)):
            lo = max(0, t0 - 4 * dil)
            xc = O.lrelu(x[bi:bi + 1, lo:t0 + n].double().cpu())
            ref = O.pgcnn2d_product(xc, wa.double().cpu(), wb.double().cpu(), dil, True)
            if lo > 0:       # the crop's own zero padding is wrong for its first 4 dil rows: they are the halo
                ref = ref[:, t0 - lo:]
            close(y[bi:bi + 1, t0:t0 + n], ref, 2e-4, 2e-5, 'gated layer dil={} crop t0={}'.format(dil, t0))

    # ---- causality of the spectral branch (BatchNorm on its moving statistics)
    spec_model = kl.Model(inputs=mod.kerasmodel.inputs[0], outputs=mod.node_spec)
    t0 = 1200
    X2 = X.clone(); X2[:, t0:] = torch.rand_like(X2[:, t0:]) * 2 - 1
    with torch.no_grad():
        s1 = spec_model(X, training=False); s2 = spec_model(X2, training=False)
    # (the context Conv1D's split product combines partial tiles with fp32 atomics: equal to the last bits, not bit for bit)
    assert rel_l2(s2[:, :t0 - 10], s1[:, :t0 - 10]) < 1e-5, 'frames before t0 - 10 changed'
    assert rel_l2(s2[:, t0:], s1[:, t0:]) > 1e-2

    # ---- one critic step and one generator step at T = 2000
    lc = opt.critic_step(X, Y)
    lg = opt.generator_step(X, Y)
    torch.cuda.synchronize()
    assert torch.isfinite(lc) and torch.isfinite(lg), (float(lc), float(lg))
    assert torch.isfinite(opt.critic_opti.flat.grad).all() and torch.isfinite(opt.gen_opti.flat.grad).all()
    assert float(opt.gen_opti.flat.grad.abs().max()) > 0 and float(opt.critic_opti.flat.grad.abs().max()) > 0


# ---- round 4: the fp64 oracle against the BASELINE ARCHITECTURE end to end (H = 256, ctx = 601, L = 8, C = 4, spec 65, nm 20) ----
# The CPU oracle does a B = 16, T = 400 critic step with all gradients in seconds (fp64, host cores), so the whole networks at
# full width meet it: predict (modeltts.py:68-69), the critic loss parts and every weight gradient (optimizertts_wgan.py:115-154),
# the generator loss, gradients and BatchNorm moving statistics (:157-213).  Weights and the interpolation weights alpha are
# injected (the reference draws both from TF's RNG).
@pytest.fixture(scope='module')
def full_arch():
    import bench
    import percivaltts_amd
    from percivaltts_amd import vocoders, modeltts_common, networks_critic, optimizertts_wgan

    class A: batch = 16; frames = T; ctx = CTX
    cfg = bench.make_cfg(A)
    cfg.train_wgan_critic_LSWGANtransidx = 30.0
    voc = vocoders.VocoderPML(16000, 0.005, SPEC, NM)
    a = O.Arch(CTX, SPEC, NM, 256, 1, 21, 8, 4, 5, 5)
    # IDENTICAL inputs on both sides: weights (and, below, inputs and alpha) are fp32-representable values held in fp64 by the
    # oracle -- what is compared is the arithmetic, not the rounding of the operands on their way to the device
    gw = [w.float().double() for w in O.random_weights(O.generator_weight_shapes(a), seed=11)]
    cw = [w.float().double() for w in O.random_weights(O.critic_weight_shapes(a), seed=12)]
    with contextlib.redirect_stdout(io.StringIO()):
        mod = modeltts_common.DCNNF0SpecNoiseFeatures(CTX, voc, cfg)
        crit = networks_critic.Critic(voc, CTX, cfg)
        assert mod.count_params() == O.count_params(O.generator_weight_shapes(a)) == 4707471
        assert crit.model.count_params() == O.count_params(O.critic_weight_shapes(a)) == 3629941
        mod.kerasmodel.set_weights([w.numpy() for w in gw])
        crit.model.set_weights([w.numpy() for w in cw])
        opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype='WLSWGAN', critic=crit)
        opt.prepare()
    return cfg, mod, crit, opt, a, gw, cw


def _full_inputs(Bq, seed):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(Bq, T, CTX, generator=g, dtype=torch.float64) * 2 - 1
    Y = torch.randn(Bq, T, 1 + SPEC + NM, generator=g, dtype=torch.float64)
    Y[:, :, 1 + SPEC:] = torch.rand(Bq, T, NM, generator=g, dtype=torch.float64)
    al = torch.rand(Bq, generator=g, dtype=torch.float64)
    return X.float().double(), Y.float().double(), al.float().double()


def _reset_weights(mod, crit, opt, gw, cw):
    """The tests below share one pair of networks: put the injected weights (and BatchNorm moving statistics) back."""
    from percivaltts_amd import ops
    opt.wait_updates()
    mod.kerasmodel.set_weights([w.detach().numpy() for w in gw])
    crit.model.set_weights([w.detach().numpy() for w in cw])
    opt.gen_opti.flat.epoch += 1; opt.critic_opti.flat.epoch += 1
    ops.clear_caches()


def _f32(t):
    return t.to(torch.float32).cuda().contiguous()


def test_predict_at_baseline_architecture_against_oracle(full_arch):
    """ModelTTS.predict (reference modeltts.py:68-69) on [2,400,601] at H = 256: all 86 output columns within the north
    star's rtol 1e-3 of the fp64 oracle (BatchNorm on its moving statistics, BLSTM over 400 frames, the 8-layer conv stack)."""
    import numpy as np
    cfg, mod, crit, opt, a, gw, cw = full_arch
    _reset_weights(mod, crit, opt, gw, cw)
    X, _, _ = _full_inputs(2, 41)
    want = O.generator_forward([w.detach() for w in gw], a, X, training=False)
    got = torch.as_tensor(mod.predict(X.numpy().astype(np.float32)))
    assert tuple(got.shape) == (2, T, 86)
    scale = float(want.abs().mean())
    close(got, want, 1e-3, 1e-3 * scale, 'predict, all 86 columns')
    for name, lo, hi in (('f0', 0, 1), ('spec', 1, 1 + SPEC), ('noise', 1 + SPEC, 86)):
        assert rel_l2(got[:, :, lo:hi], want[:, :, lo:hi]) < 1e-4, (name, rel_l2(got[:, :, lo:hi], want[:, :, lo:hi]))


def _critic_step_device(opt, Xd, Yd, ald, part=None):
    """The device side of a critic step up to (not including) the update, as OptimizerTTSWGAN._critic_grads runs it, keeping the
    loss parts (part = 0 / 1 / 2: only that part is backpropagated); returns (total, parts, C-ABI call names)."""
    from percivaltts_amd import ops, _hip
    opt.critic_opti.zero_grad()
    with _hip.KernelTimer() as kt:
        with ops.deferred_weight_grads():
            total, parts = opt.critic_loss(Xd, Yd, ald, training=True)
            (total if part is None else parts[part]).backward()
    torch.cuda.synchronize()
    return total, parts, [r[0] for r in kt.records]


def test_critic_step_at_baseline_architecture_against_oracle(full_arch):
    """One critic step at B = 16, T = 400 (6 400 frames: the frequency-domain context Conv1D, the split Dense kernels, the two-stage
    weight gradients and the matrix-core Conv2D kernels are what runs -- asserted) against fp64 `critic_step_loss`
    (reference optimizertts_wgan.py:115-154) on identical fp32-representable weights, inputs and alpha: the three loss parts at 5e-4
    and EVERY weight-gradient tensor by relative L2, with no per-tensor escape.

    How the gradient bounds are set.  The critic loss is -mean D(y) + mean D(G(x)) + 10 gp, and for the CONTEXT branch (Conv1D +
    two Dense layers: it sees the same context frames in both evaluations and the penalty does not reach it) the two Wasserstein
    terms nearly cancel: |g_total| is 0.6 ... 1 % of |g_valid| + |g_fake| there (tools/fullarch_debug.py).  An error of 1e-5 of the
    parts is then 1e-3 of the total -- for ANY fp32 evaluation: the oracle itself run in fp32 is off by 4.7e-3 / 3.2e-3 / 2.5e-3 on
    those three kernels against fp64, the device by 4.8e-3 / 3.2e-3 / 2.5e-3.  So (1) each PART's gradient is checked against the
    oracle's gradient of that part, relative to its own norm -- the stringent check, nothing cancels inside a part; (2) the total's
    tensors are bounded relative to the sum of their parts' norms (the conditioning-aware form of "relative L2"), and over the whole
    network relative to the total.  The Conv1D kernel's gradient keeps a wider bound than the other tensors inside a part: its
    signal sum_t x[t+k] dz[t] is an incoherent sum over white-noise context frames, so ONE LeakyReLU mask that falls on the other side
    of zero in fp32 (1 of 1.6 M pre-activations, |z| < 1e-6 rms) already moves it by 0.7 / sqrt(1.6 M) = 5.5e-4 of its norm; two
    different weight-gradient kernels (frequency and time domain) fed the same dz agree to 8e-7."""
    cfg, mod, crit, opt, a, gw, cw = full_arch
    _reset_weights(mod, crit, opt, gw, cw)
    X, Y, al = _full_inputs(16, 42)
    cwr = [w.detach().clone().requires_grad_(True) for w in cw]
    total, parts = O.critic_step_loss(cwr, [w.detach() for w in gw], a, X, Y, al, gp_lambda=10.0)
    pg = [torch.autograd.grad(parts[k], cwr, retain_graph=True, allow_unused=True) for k in ('valid', 'fake', 'gp')]
    grads = torch.autograd.grad(total, cwr)
    Xd, Yd, ald = _f32(X), _f32(Y), _f32(al)
    params = opt.critic_opti.flat.params
    zero = lambda p: torch.zeros(tuple(p.shape), dtype=torch.float64)

    # (1) the three parts, each against its own gradient
    for ki, kname in enumerate(('valid', 'fake', 'gp')):
        _critic_step_device(opt, Xd, Yd, ald, part=ki)
        want = [g_ if g_ is not None else zero(p) for p, g_ in zip(params, pg[ki])]
        gmax = max(float(w_.norm()) for w_ in want)
        num = den = 0.0
        for p, w_ in zip(params, want):
            e = float((p.grad.detach().cpu().double() - w_).norm()); n = float(w_.norm())
            num += e * e; den += n * n
            bound = 3e-3 if tuple(w_.shape) == (21, CTX, 256) else 5e-4
            assert e <= bound * max(n, 1e-4 * gmax), 'gradient of the {} part, tensor {}: relative L2 error {:.3e}'.format(
                kname, tuple(w_.shape), e / max(n, 1e-300))
        assert num <= (1e-3 ** 2) * den, 'gradient of the {} part: relative L2 error {:.3e} over all tensors'.format(kname, (num / den) ** 0.5)
        print('critic step B=16, part {}: all-network gradient rel L2 {:.3e}'.format(kname, (num / den) ** 0.5))

    # (2) the step itself: loss parts, which kernels ran, the total gradient
    tot_d, (lv, lf, gp), names = _critic_step_device(opt, Xd, Yd, ald)
    close(lv, parts['valid'], 5e-4, 1e-5, 'L valid')
    close(lf, parts['fake'], 5e-4, 1e-5, 'L fake')
    close(gp, parts['gp'], 5e-4, 1e-5, 'gradient penalty')
    close(tot_d, total, 5e-4, 1e-5, 'critic loss')
    # the kernels of the BASELINE-size step ran, not their small-shape fallbacks
    for need in ('ptts_dense_bf16x6_batched', 'ptts_split3_frame_windows', 'ptts_conv1d_freq_wgrad_inverse',     # frequency-domain Conv1D
                 'ptts_dense_bf16x6', 'ptts_dense_wgrad_bf16x6_partials', 'ptts_dense_wgrad_reduce_grouped',      # split Dense + two-stage dW
                 'ptts_conv2d_mfma_fwd', 'ptts_conv2d_mfma_bwd_fused', 'ptts_conv2d_reduce_grouped'):                            # matrix-core Conv2D, fused backward
        assert need in names, '{} did not run in the B = 16 critic step: {}'.format(need, sorted(set(names)))
    num = den = 0.0
    worst = (0.0, None)
    for i, (p, g_) in enumerate(zip(params, grads)):
        e = float((p.grad.detach().cpu().double() - g_).norm()); n = float(g_.norm())
        num += e * e; den += n * n
        nparts = sum(float(pg[k][i].norm()) for k in range(3) if pg[k][i] is not None)
        r = e / max(nparts, 1e-300) if nparts > 0 else 0.0
        if r > worst[0]: worst = (r, tuple(g_.shape))
        # (the output bias: exactly zero on both sides -- the Wasserstein terms cancel it, the penalty does not see it)
        bound = 1e-3 if tuple(g_.shape) == (21, CTX, 256) else 3e-4       # (the Conv1D kernel: one flipped mask = 5.5e-4 of a part, see above)
        assert e <= bound * nparts, 'critic gradient {}: error {:.3e} of the parts\' norms ({:.3e} of its own)'.format(tuple(g_.shape), r, e / max(n, 1e-300))
    assert num <= (3e-3 ** 2) * den, 'critic gradients: relative L2 error {:.3e} over all tensors'.format((num / den) ** 0.5)
    print('critic step B=16: all-network gradient rel L2 {:.3e}; worst tensor against its parts {} {:.3e}'.format((num / den) ** 0.5, worst[1], worst[0]))


def test_critic_loss_parts_at_batch_64_against_oracle(full_arch):
    """The same at BASELINE configs[1]'s own batch, B = 64, T = 400: the three loss parts against the fp64 oracle (its loss alone
    is ~30 s of host time; the gradients are covered at B = 16 above and by the shard-mean property at B = 64)."""
    cfg, mod, crit, opt, a, gw, cw = full_arch
    _reset_weights(mod, crit, opt, gw, cw)
    X, Y, al = _full_inputs(64, 43)
    with torch.no_grad():
        gwd = [w.detach() for w in gw]; cwd = [w.detach() for w in cw]
        fake = O.generator_forward(gwd, a, X, training=True)
        valid = O.critic_forward(cwd, a, Y, X)
        fake_v = O.critic_forward(cwd, a, fake, X)
    x_hat = O.random_weighted_average(Y, fake, al).detach().requires_grad_(True)
    v_hat = O.critic_forward(cwd, a, x_hat, X)
    g = torch.autograd.grad(v_hat.sum(), x_hat)[0]
    gp_want = ((1 - torch.sqrt((g * g).sum(dim=(1, 2)))) ** 2).mean()
    tot_d, (lv, lf, gp), names = _critic_step_device(opt, _f32(X), _f32(Y), _f32(al))
    close(lv, -valid.mean(), 5e-4, 1e-5, 'L valid, B = 64')
    close(lf, fake_v.mean(), 5e-4, 1e-5, 'L fake, B = 64')
    close(gp, gp_want, 5e-4, 1e-5, 'gradient penalty, B = 64')
    close(tot_d, -valid.mean() + fake_v.mean() + 10.0 * gp_want, 5e-4, 1e-5, 'critic loss, B = 64')
    assert torch.isfinite(opt.critic_opti.flat.grad).all()


def test_generator_step_at_baseline_architecture_against_oracle(full_arch):
    """One generator step at B = 12, T = 400 (4 800 frames: above the 4 096 at which the frequency-domain Conv1D and the split
    weight gradients take over) against fp64 `generator_step_loss` (reference optimizertts_wgan.py:157-213): both loss terms, every
    trainable tensor's gradient by relative L2, and the BatchNorm moving statistics after the training forward."""
    from percivaltts_amd import ops
    cfg, mod, crit, opt, a, gw, cw = full_arch
    _reset_weights(mod, crit, opt, gw, cw)
    X, Y, _ = _full_inputs(12, 44)
    shapes = O.generator_weight_shapes(a)
    trainable, i = [], 0
    while i < len(shapes):      # everything except the BatchNorm moving statistics (3rd / 4th of each run of four equal 1-D shapes)
        if len(shapes[i]) == 1 and i + 3 < len(shapes) and all(shapes[i + k] == shapes[i] for k in range(4)):
            trainable += [i, i + 1]; i += 4
        else:
            trainable.append(i); i += 1
    gw_t = [w.detach().clone() for w in gw]
    for i in trainable: gw_t[i].requires_grad_(True)
    w_ls, ww = O.wls_weights(a.specsize, a.noisesize, 0, 0.25, 30.0)
    ltot, lparts = O.generator_step_loss([w.detach() for w in cw], gw_t, a, X, Y, 'WLSWGAN', torch.tensor(w_ls), ww, update_moving=True)
    ggrads = torch.autograd.grad(ltot, [gw_t[i] for i in trainable], allow_unused=True)
    opt.gen_opti.zero_grad()
    cps = opt.critic_opti.flat.params
    for p in cps: p.requires_grad_(False)
    try:
        with ops.deferred_weight_grads():
            ltot_d, (lw_d, lls_d) = opt.generator_loss(_f32(X), _f32(Y), training=True)
            ltot_d.backward()
    finally:
        for p in cps: p.requires_grad_(True)
    torch.cuda.synchronize()
    close(lw_d, lparts['wgan'], 5e-4, 1e-5, 'generator wgan term')
    close(lls_d, lparts['ls'], 5e-4, 1e-5, 'generator ls term')
    close(ltot_d, ltot, 5e-4, 1e-5, 'generator loss')
    num = den = 0.0
    worst = (0.0, None)
    gmax = max(float(g_.norm()) for g_ in ggrads if g_ is not None)
    for p, g_ in zip(opt.gen_opti.flat.params, ggrads):
        want = g_ if g_ is not None else torch.zeros(tuple(p.shape), dtype=torch.float64)
        e = float((p.grad.detach().cpu().double() - want).norm()); n = float(want.norm())
        num += e * e; den += n * n
        # (a bias in front of a BatchNorm has an exactly-zero gradient: bounded against the largest tensor instead of itself)
        r = e / max(n, 1e-6 * gmax)
        if r > worst[0]: worst = (r, tuple(want.shape))
        assert r <= 5e-3, 'generator gradient {}: relative L2 error {:.3e}'.format(tuple(want.shape), r)
    assert num <= (2e-3 ** 2) * den, 'generator gradients: relative L2 error {:.3e} over all tensors (worst tensor {} {:.3e})'.format(
        (num / den) ** 0.5, worst[1], worst[0])
    print('generator step B=12: all-network gradient rel L2 {:.3e}, worst tensor {} {:.3e}'.format((num / den) ** 0.5, worst[1], worst[0]))
    for (k, t), w in zip(mod.kerasmodel.weights(), gw_t):
        if 'moving' in k:
            close(t, w, 5e-4, 1e-5, k)
    _reset_weights(mod, crit, opt, gw, cw)

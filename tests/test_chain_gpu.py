"""GPU parity of the fused Conv2D-stack kernels (csrc/conv2d_chain.hip, BASELINE configs[2]) against the fp64 oracle with the
same roundings (oracle.critic_forward(..., bf16_stack='chain'): spectrum, kernels and every layer's post-activation output rounded
to bf16).  Reference: networks_critic.py:64-70 (the stack), optimizertts_wgan.py:53-68 (the gradient of a gradient).

Tolerances.  A layer fed with the kernel's OWN previous map differs from the oracle only where the fp32 sum lies within an
accumulation error of a bf16 rounding boundary: at most one bf16 ulp (2^-8 relative), checked pixel by pixel for every layer
(tile borders, utterance borders, the ragged last tile included).  End to end the flipped roundings propagate: relative L2
error <= 2^-7.  Gradient maps are rounded to bf16 inside the kernels and not in the oracle (straight-through): 2e-2 relative L2."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import percival_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    from percivaltts_amd import ops as _ops
    return _ops


def _weights(L, cin0, seed):
    g = torch.Generator().manual_seed(seed)
    ws, bs = [], []
    cin = cin0
    for _ in range(L):
        lim = (6.0 / (25 * cin + 25 * 4)) ** 0.5
        ws.append((torch.rand(5, 5, cin, 4, generator=g, dtype=torch.float64) * 2 - 1) * lim * 1.5)
        bs.append(0.1 * torch.randn(4, generator=g, dtype=torch.float64))
        cin = 4
    return ws, bs


def _bf(x):
    return x.to(torch.float32).to(torch.bfloat16).to(torch.float64)


def _oracle_stack(x0, ws, bs, alpha=0.3):
    """a_1 .. a_L of the chain in fp64 with its roundings (straight-through for autograd)."""
    h = O.bf16_st(x0.unsqueeze(-1))
    outs = []
    for w, b in zip(ws, bs):
        h = O.bf16_st(O.lrelu(O.conv2d_nhwc(h, O.bf16_st(w), b)))
        outs.append(h)
    return outs


CASES = [
    dict(B=2, T=50, F=65, L=8),       # BASELINE geometry: 17 bin groups (odd), two tiles, the second ragged
    dict(B=1, T=100, F=65, L=8),      # inner tiles with both neighbours
    dict(B=1, T=37, F=13, L=3),       # few bins, odd T
    dict(B=3, T=16, F=66, L=2),       # even F (no pad bin), one short tile per utterance
    dict(B=2, T=33, F=24, L=1),       # a single layer
    dict(B=1, T=1, F=65, L=8),        # one frame (update_validation_cost runs whole utterances of any length)
]


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'B{B}T{T}F{F}L{L}'.format(**c))
def test_chain_forward_layer_by_layer(ops, case):
    from percivaltts_amd import _hip
    B, T, F, L = case['B'], case['T'], case['F'], case['L']
    ws, bs = _weights(L, 1, 3)
    g = torch.Generator().manual_seed(7)
    x0 = torch.randn(B, T, F, generator=g, dtype=torch.float64)
    wd = [w.float().cuda().contiguous() for w in ws]
    bd = [b.float().cuda().contiguous() for b in bs]
    xd = x0.float().cuda().contiguous()
    tab = ops._C2C.table(wd, bd)
    FP = (F + 1) & ~1
    maps = torch.full((max(L - 1, 1), B, T, FP, 4), float('nan'), dtype=torch.bfloat16, device='cuda')
    a_last = torch.full((B, T, F, 4), float('nan'), dtype=torch.bfloat16, device='cuda')
    _hip.call('ptts_conv2d_chain_fwd', _hip.ptr(xd), xd.stride(1), _hip.ptr(tab), _hip.ptr(maps), _hip.ptr(a_last), B, T, F, L, 0.3, _hip.stream())
    torch.cuda.synchronize()
    got = [maps[l][:, :, :F].double().cpu() for l in range(L - 1)] + [a_last.double().cpu()]
    if FP != F and L > 1:
        assert float(maps[:L - 1, :, :, F:].float().abs().max()) == 0.0, 'the pad bin of the internal maps must be zero'
    # (i) every layer against the oracle's layer applied to the kernel's own previous map: one bf16 ulp
    prev = _bf(x0).unsqueeze(-1)
    for l in range(L):
        z = O.conv2d_nhwc(prev, _bf(ws[l]), bs[l])
        ref = _bf(O.lrelu(z))
        d = (got[l] - ref).abs()
        assert torch.isfinite(got[l]).all(), 'layer {}: non-finite output'.format(l + 1)
        tol = 2.0 ** -7 * ref.abs() + 4e-6 * (1.0 + float(ref.abs().max()))      # + the fp32 accumulation error where the sum cancels
        # a sum that differs in the last fp32 bits can land on the other side of a rounding boundary: one ulp = 2^-8 .. 2^-7 relative
        bad = d > tol
        assert int(bad.sum()) == 0, 'layer {}: {} of {} values off by more than one bf16 ulp (max {:.3e})'.format(
            l + 1, int(bad.sum()), d.numel(), float(d.max()))
        assert float((d > 0).double().mean()) < 0.02, 'layer {}: {:.2%} of the values differ from the oracle'.format(l + 1, float((d > 0).double().mean()))
        prev = got[l]
    # (ii) end to end against the oracle's own chain
    refs = _oracle_stack(x0, ws, bs)
    num = float(((got[-1] - refs[-1]) ** 2).sum()); den = float((refs[-1] ** 2).sum())
    assert num <= (2.0 ** -7) ** 2 * den, 'a_L: relative L2 error {:.3e}'.format((num / max(den, 1e-300)) ** 0.5)


def _rel(a, b):
    return float(((a - b) ** 2).sum() ** 0.5 / max(float((b ** 2).sum() ** 0.5), 1e-300))


def _convT(d, w, like):
    """Gradient of conv2d_nhwc(x, w) w.r.t. x against d (the transposed convolution), fp64."""
    x = torch.zeros_like(like, requires_grad=True)
    return torch.autograd.grad(O.conv2d_nhwc(x, w), x, d)[0]


def _conv_dw(h, d, w_like):
    w = torch.zeros_like(w_like, requires_grad=True)
    return torch.autograd.grad(O.conv2d_nhwc(h, w), w, d)[0]


def _device_maps(ops, xd, wd, bd):
    """a_1 .. a_L as the forward kernel stores them (fp64 copies on the host)."""
    from percivaltts_amd import _hip
    B, T, F = xd.shape
    L = len(wd)
    FP = (F + 1) & ~1
    tab = ops._C2C.table(wd, bd)
    maps = torch.zeros((max(L - 1, 1), B, T, FP, 4), dtype=torch.bfloat16, device='cuda')
    a_last = torch.zeros((B, T, F, 4), dtype=torch.bfloat16, device='cuda')
    _hip.call('ptts_conv2d_chain_fwd', _hip.ptr(xd), xd.stride(1), _hip.ptr(tab), _hip.ptr(maps), _hip.ptr(a_last), B, T, F, L, 0.3, _hip.stream())
    torch.cuda.synchronize()
    return [maps[l][:, :, :F].double().cpu() for l in range(L - 1)] + [a_last.double().cpu()]


def _manual_chain(x0, ws, bs, R, S, maps=None, alpha=0.3):
    """The kernels' arithmetic restated step by step in fp64 WITH their roundings -- also those of the gradient maps, which the
    straight-through autograd oracle does not round: d_l, gamma_l, u_l are bf16 in the LDS / HBM; the bias gradients sum the
    unrounded products; d_last and u_0 arrive rounded to bf16.  `maps`: the activation maps a_1 .. a_L to use (the forward
    kernel's own: a rounding that fell the other way in the forward would otherwise be charged to the backward)."""
    L = len(ws)
    wq = [_bf(w) for w in ws]
    a = [_bf(x0).unsqueeze(-1)]
    for l in range(L):
        a.append(maps[l] if maps is not None else _bf(O.lrelu(O.conv2d_nhwc(a[-1], wq[l], bs[l]))))
    mask = lambda t: torch.where(t > 0, torch.ones_like(t), torch.full_like(t, alpha))
    dw, db, d = [None] * L, [None] * L, [None] * (L + 1)
    v = _bf(R) * mask(a[L])
    for l in range(L, 0, -1):
        db[l - 1] = v.sum(dim=(0, 1, 2))
        d[l] = _bf(v)
        dw[l - 1] = _conv_dw(a[l - 1], d[l], ws[l - 1])
        back = _convT(d[l], wq[l - 1], a[l - 1])
        if l > 1:
            v = back * mask(a[l - 1])
    g0 = back[..., 0]
    u = _bf(S).unsqueeze(-1)
    dw2 = [None] * L
    for l in range(1, L + 1):
        dw2[l - 1] = _conv_dw(u, d[l], ws[l - 1])
        v = O.conv2d_nhwc(u, wq[l - 1]) * mask(a[l])
        u = _bf(v)
    return dw, db, g0, dw2, v


@pytest.mark.parametrize('case', CASES[:5], ids=lambda c: 'B{B}T{T}F{F}L{L}'.format(**c))
def test_chain_first_and_second_order_gradients(ops, case):
    """Through ops.conv2d_chain (the autograd Functions the critic uses): dW_l, db_l of a linear functional of a_L; the
    backward-data pass g0 = d(sum R.a_L)/dx0; and the second-order sweep -- gradients of sum(S . g0) w.r.t. every kernel and
    w.r.t. R.  Against (i) the kernels' arithmetic restated in fp64 with ALL their roundings on the forward kernel's own maps
    (_manual_chain: what is left is fp32 summation order, 2e-3) and (ii) the oracle's fp64 autograd of the rounded forward, which does not round the gradient
    maps (2^-9 per stored gradient value, accumulated over up to 8 layers: 2e-2 / 3e-2 relative L2 on the kernel gradients and
    on g0; the bias gradients are sums of thousands of signed terms that cancel, so their straight-through comparison is
    relative to the sum of magnitudes)."""
    B, T, F, L = case['B'], case['T'], case['F'], case['L']
    ws, bs = _weights(L, 1, 5)
    g = torch.Generator().manual_seed(11)
    x0 = torch.randn(B, T, F, generator=g, dtype=torch.float64)
    R = torch.randn(B, T, F, 4, generator=g, dtype=torch.float64)
    S = torch.randn(B, T, F, generator=g, dtype=torch.float64)

    # ---- oracle (straight-through autograd)
    wo = [w.clone().requires_grad_(True) for w in ws]
    bo = [b.clone().requires_grad_(True) for b in bs]
    xo = x0.clone().requires_grad_(True)
    Ro = R.clone().requires_grad_(True)
    aL = _oracle_stack(xo, wo, bo)[-1]
    l1 = (aL * Ro).sum()
    g1 = torch.autograd.grad(l1, wo + bo, retain_graph=True)
    g0o = torch.autograd.grad(l1, xo, create_graph=True)[0]
    l2 = (g0o * S).sum()
    g2 = torch.autograd.grad(l2, wo + [Ro])

    # ---- device
    wd = [w.float().cuda().requires_grad_(True) for w in ws]
    bd = [b.float().cuda().requires_grad_(True) for b in bs]
    xd = x0.float().cuda().requires_grad_(True)
    with torch.no_grad():
        dev_maps = _device_maps(ops, xd.detach(), [w.detach() for w in wd], [b.detach() for b in bd])
    m_dw, m_db, m_g0, m_dw2, m_out = _manual_chain(x0, ws, bs, R, S, dev_maps)
    Rd = R.float().cuda().requires_grad_(True)
    a = ops.conv2d_chain(xd, wd, bd, 0.3)
    assert a.dtype == torch.bfloat16 and tuple(a.shape) == (B, T, F, 4)
    l1d = (a.float() * Rd).sum()
    g1d = torch.autograd.grad(l1d, wd + bd, retain_graph=True)
    g0d = torch.autograd.grad(l1d, xd, create_graph=True)[0]
    l2d = (g0d * S.float().cuda()).sum()
    g2d = torch.autograd.grad(l2d, wd + [Rd])
    torch.cuda.synchronize()
    cpu = lambda t: t.detach().double().cpu()

    errs = {}
    for l in range(L):
        errs['dW{} same roundings'.format(l + 1)] = (_rel(cpu(g1d[l]), m_dw[l]), 2e-3)
        errs['db{} same roundings'.format(l + 1)] = (_rel(cpu(g1d[L + l]), m_db[l]), 2e-3)
        errs['dW{} autograd'.format(l + 1)] = (_rel(cpu(g1d[l]), g1[l]), 2e-2)
        errs['second-order dW{} same roundings'.format(l + 1)] = (_rel(cpu(g2d[l]), m_dw2[l]), 2e-3)
        errs['second-order dW{} autograd'.format(l + 1)] = (_rel(cpu(g2d[l]), g2[l]), 3e-2)
    errs['g0 same roundings'] = (_rel(cpu(g0d), m_g0), 2e-3)
    errs['g0 autograd'] = (_rel(cpu(g0d), g0o.detach()), 2e-2)
    errs['second-order d/dR same roundings'] = (_rel(cpu(g2d[L]), _bf(m_out)), 2e-3)
    errs['second-order d/dR autograd'] = (_rel(cpu(g2d[L]), g2[L]), 3e-2)
    for gd in list(g1d) + list(g2d) + [g0d]:
        assert torch.isfinite(gd).all()
    bad = ['{}: {:.3e} > {:.0e}'.format(k, e, t) for k, (e, t) in errs.items() if not e < t]
    assert not bad, '; '.join(bad) + ' || all: ' + ', '.join('{} {:.1e}'.format(k, e) for k, (e, t) in errs.items())


def test_chain_linearity_and_tile_independence_at_full_size(ops):
    """BASELINE size ([64,400,65], 8 layers), where the oracle needs minutes: size-independent properties.  (i) A tile's
    result does not depend on what else is in the batch: the maps of utterance 5 alone equal those inside the batch, bit for
    bit.  (ii) The backward-data pass is linear in d_last: g0(R1 + R2) = g0(R1) + g0(R2) up to the bf16 roundings of the
    gradient maps.  (iii) Oracle on a crop: frames 180..230 of one utterance (tile borders at 192 and 224) against the fp64
    chain run on frames 150..260 (the halo of 16 rows per side covered)."""
    B, T, F, L = 64, 400, 65, 8
    ws, bs = _weights(L, 1, 9)
    g = torch.Generator().manual_seed(13)
    x0 = torch.randn(B, T, F, generator=g, dtype=torch.float32)
    wd = [w.float().cuda() for w in ws]
    bd = [b.float().cuda() for b in bs]
    xd = x0.cuda()
    with torch.no_grad():
        a = ops.conv2d_chain(xd, wd, bd, 0.3)
        a5 = ops.conv2d_chain(xd[5:6].contiguous(), wd, bd, 0.3)
    assert torch.equal(a[5:6], a5), 'a tile depends on its neighbours in the batch'
    crop = _oracle_stack(x0[5:6, 150:260].double(), ws, bs)[-1][:, 30:80]
    got = a[5:6, 180:230].double().cpu()
    assert _rel(got, crop) < 2.0 ** -7, 'crop of a_L against the oracle: {:.3e}'.format(_rel(got, crop))
    R1 = torch.randn(B, T, F, 4, generator=g).cuda()
    R2 = torch.randn(B, T, F, 4, generator=g).cuda()
    xr = xd.clone().requires_grad_(True)
    ar = ops.conv2d_chain(xr, wd, bd, 0.3)
    with ops.input_grad_only():
        g12 = torch.autograd.grad(ar, xr, grad_outputs=(R1 + R2).to(ar.dtype), retain_graph=True)[0]
        g1 = torch.autograd.grad(ar, xr, grad_outputs=R1.to(ar.dtype), retain_graph=True)[0]
        g2 = torch.autograd.grad(ar, xr, grad_outputs=R2.to(ar.dtype))[0]
    assert _rel((g1 + g2).double().cpu(), g12.double().cpu()) < 2e-2

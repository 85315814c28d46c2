"""GPU parity of every HIP op (through the C ABI) against the CPU oracle, on seeded inputs.
fp32 kernels vs fp64 oracle: rtol 1e-4 / atol 1e-5 unless a test states otherwise
(the north star's bar is 1e-3 rtol)."""
import math

import pytest
import torch

from oracle import percival_oracle as O

pytestmark = pytest.mark.gpu

RT, AT = 1e-4, 1e-5


def dev(t, grad=False):
    return t.detach().to(torch.float32).cuda().contiguous().requires_grad_(grad)


def ref(t, grad=False):
    return t.detach().to(torch.float64).cpu().requires_grad_(grad)


def close(got, want, rtol=RT, atol=AT, what=''):
    got = got.detach().cpu().to(torch.float64)
    want = want.detach().to(torch.float64)
    assert got.shape == want.shape, '{}: shape {} vs {}'.format(what, tuple(got.shape), tuple(want.shape))
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    if bad.any():
        i = int(torch.argmax(err - tol))
        raise AssertionError('{}: {} / {} mismatches, worst |err|={:.3e} at flat {} (got {:.6e}, want {:.6e}), max|want|={:.3e}'.format(
            what, int(bad.sum()), bad.numel(), float(err.flatten()[i]), i, float(got.flatten()[i]),
            float(want.flatten()[i]), float(want.abs().max())))


@pytest.fixture(scope='module')
def ops():
    from percivaltts_amd import ops as _ops
    return _ops


def gen(seed):
    return torch.Generator().manual_seed(seed)


def test_library_identity():
    from percivaltts_amd import _hip
    assert _hip.lib().ptts_device_arch() == b'gfx950'
    assert torch.cuda.is_available()


CONV_CASES = [
    # B, T, F, Cin, Cout, KT, KF, dil, causal
    (2, 12, 9, 1, 4, 5, 5, 1, False),
    (2, 12, 9, 4, 4, 5, 5, 1, False),
    (2, 12, 9, 4, 1, 5, 5, 1, False),
    (2, 12, 9, 1, 2, 3, 3, 1, False),
    (2, 12, 9, 2, 2, 3, 3, 1, False),
    (2, 12, 9, 2, 1, 3, 3, 1, False),
    (3, 100, 65, 4, 4, 5, 5, 1, False),     # several time tiles, real frequency width
    (1, 37, 129, 4, 4, 5, 5, 1, False),     # reference-shaped F
    (2, 50, 65, 4, 4, 5, 5, 2, True),       # dilated causal (config 5 extension)
    (2, 10, 7, 3, 5, 3, 3, 1, False),       # generic fallback kernels
    (1, 3, 2, 1, 4, 5, 5, 1, False),        # image smaller than the kernel
]


@pytest.mark.parametrize('case', CONV_CASES)
@pytest.mark.parametrize('mode', ['none', 'lrelu', 'affine'])
def test_conv2d_forward_backward(ops, case, mode):
    B, T, F, Cin, Cout, KT, KF, dil, causal = case
    g = gen(1)
    x = torch.randn(B, T, F, Cin, generator=g, dtype=torch.float64)
    w = torch.randn(KT, KF, Cin, Cout, generator=g, dtype=torch.float64) * 0.3
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    sc = torch.rand(Cin, generator=g, dtype=torch.float64) + 0.5
    sh = torch.randn(Cin, generator=g, dtype=torch.float64) * 0.3
    dy = torch.randn(B, T, F, Cout, generator=g, dtype=torch.float64)

    xr, wr, br, scr, shr = ref(x, True), ref(w, True), ref(b, True), ref(sc, True), ref(sh, True)
    if mode == 'none':
        a = xr
    elif mode == 'lrelu':
        a = O.lrelu(xr)
    else:
        a = O.lrelu(xr * scr + shr)
    yr = O.conv2d_nhwc(a, wr, br, dil_t=dil, causal=causal)
    yr.backward(dy)

    xd, wd, bd, scd, shd = dev(x, True), dev(w, True), dev(b, True), dev(sc, True), dev(sh, True)
    if mode == 'none':
        v = xd
    elif mode == 'lrelu':
        v = ops.Lazy(xd, lrelu=True)
    else:
        v = ops.Lazy(xd, scd, shd, lrelu=True)
    yd = ops.conv2d(v, wd, bd, dil_t=dil, pad_mode=ops.PAD_CAUSAL if causal else ops.PAD_SAME)
    close(yd, yr, what='y')
    yd.backward(dev(dy))
    close(xd.grad, xr.grad, what='dx')
    close(wd.grad, wr.grad, rtol=2e-4, atol=1e-4, what='dw')
    close(bd.grad, br.grad, rtol=2e-4, atol=1e-4, what='db')
    if mode == 'affine':
        # sums over B T F pixels of terms of either sign (a channel's sum may all but cancel: -0.59 beside 665 at case 6): the absolute
        # tolerance grows with the number of terms -- 5e-8 per pixel is a third of what fp32 products of O(1) terms allow
        at = max(1e-4, 5e-8 * B * T * F)
        close(scd.grad, scr.grad, rtol=2e-4, atol=at, what='dscale')
        close(shd.grad, shr.grad, rtol=2e-4, atol=at, what='dshift')


def test_conv2d_mfma_tables_are_the_split_of_the_toeplitz_blocks(ops):
    """csrc/conv2d_mfma.hip: the operand tables of a 5x5 4->4 kernel.  Per kernel row the MFMA A operand is the banded block
    A[(so,co)][(j,ci)] = w[kt][j-so][ci][co] (forward) resp. w[4-kt][4-(j-so)][ci][co] with the channel roles swapped
    (backward data); lane (li, lg) holds m = li -> (so, oc), k = 8 lg + e -> (j = 2 lg + e // 4, ic = e % 4), i.e. two
    consecutive taps kf = 2 lg - so, + 1 of the (kt, oc) row.  The table stores every row once, zero-padded to the taps -3..7:
    tab[kt][plane][oc][kf + 3][ic].  Checked here: the planes are oracle.np_split3_bf16 of that, bit for bit, and the
    fragment every lane reads out of it (16 bytes at slot 2 lg - so + 3) is the Toeplitz block.  The same device function
    (split3) splits the activations on their way into the LDS, where they cannot be read back."""
    from percivaltts_amd.ops import call, ptr, stream
    g = gen(51)
    w = (torch.randn(5, 5, 4, 4, generator=g) * torch.exp(2 * torch.randn(5, 5, 4, 4, generator=g))).float()
    nb = ops._hip.lib().ptts_conv2d_mfma_table_bytes(5)
    assert nb == 5 * 3 * 4 * 11 * 4 * 2
    tf = torch.zeros(nb, dtype=torch.uint8, device='cuda'); tb = torch.zeros(nb, dtype=torch.uint8, device='cuda')
    call('ptts_conv2d_mfma_tables', ptr(w.cuda()), ptr(tf), ptr(tb), 5, 5, 4, 4, 3, stream())
    wn = w.numpy()
    for tab, transposed in ((tf, False), (tb, True)):
        got = tab.view(torch.bfloat16).float().cpu().numpy().reshape(5, 3, 4, 11, 4)          # [kt][plane][oc][slot][ic]
        rows = O.np.zeros((5, 4, 11, 4), dtype=O.np.float32)
        for kt in range(5):
            for oc in range(4):
                for kf in range(5):
                    for ic in range(4):
                        rows[kt, oc, kf + 3, ic] = wn[4 - kt, 4 - kf, oc, ic] if transposed else wn[kt, kf, ic, oc]
        planes = O.np_split3_bf16(rows)
        for p in range(3):
            assert (got[:, p] == planes[p]).all(), 'plane {} of the {} table'.format(p, 'transposed' if transposed else 'forward')
        assert (got.sum(1) == rows).all()
        # the fragment of lane (li, lg): elements e = 0..7 at slot 2 lg - so + 3 + e // 4
        for kt in range(5):
            for lane in range(64):
                li, lg = lane & 15, lane >> 4
                so, oc = li >> 2, li & 3
                frag = got.sum(1)[kt, oc].reshape(-1)[(2 * lg - so + 3) * 4:(2 * lg - so + 3) * 4 + 8]
                for e in range(8):
                    j, ic = 2 * lg + e // 4, e % 4
                    kf = j - so
                    want = 0.0
                    if 0 <= kf < 5:
                        want = wn[4 - kt, 4 - kf, oc, ic] if transposed else wn[kt, kf, ic, oc]
                    assert frag[e] == want, (kt, lane, e)


@pytest.mark.parametrize('case', [
    # B, T, F, dil, causal
    (3, 100, 65, 1, False),        # the critic's layer: one block of 17 bin groups, several time tiles, a ragged last tile
    (2, 16, 65, 1, False),         # exactly one tile per utterance
    (1, 1, 65, 1, False),          # a single frame
    (2, 37, 130, 1, False),        # two blocks of bin groups, F not a multiple of 4
    (1, 50, 129, 1, True),         # reference-shaped F, causal
    (2, 40, 7, 2, False), (2, 70, 65, 4, True), (1, 90, 33, 8, False),     # every dilation instance
])
def test_conv2d_mfma_against_the_oracle_and_the_fp32_stencil(ops, case):
    """The matrix-core kernels of the 4->4 5x5 layers (bf16x6 split, csrc/conv2d_mfma.hip) in every role -- forward with
    LeakyReLU on load, with a BatchNorm affine, backward data + weight/bias gradient, and the masked (second-order)
    sweeps -- against the fp64 oracle at the tolerance of the fp32 stencil they replace (rtol 1e-4 / atol 1e-5; 2e-4 /
    1e-4 for the reductions), and A/B against that stencil (PTTS_CONV2D_MFMA=0).  The kernels are asserted to have run."""
    B, T, F, dil, causal = case
    g = gen(52)
    x = torch.randn(B, T, F, 4, generator=g, dtype=torch.float64)
    w = torch.randn(5, 5, 4, 4, generator=g, dtype=torch.float64) * 0.3
    b = torch.randn(4, generator=g, dtype=torch.float64)
    sc = torch.rand(4, generator=g, dtype=torch.float64) + 0.5
    sh = torch.randn(4, generator=g, dtype=torch.float64) * 0.3
    dy = torch.randn(B, T, F, 4, generator=g, dtype=torch.float64)
    msk = torch.randn(B, T, F, 4, generator=g, dtype=torch.float64)
    pm = ops.PAD_CAUSAL if causal else ops.PAD_SAME

    xr, wr, br = ref(x, True), ref(w, True), ref(b, True)
    yr = O.conv2d_nhwc(O.lrelu(xr), wr, br, dil_t=dil, causal=causal)
    yr.backward(dy)
    ya = O.conv2d_nhwc(O.lrelu(ref(x) * ref(sc) + ref(sh)), ref(w), ref(b), dil_t=dil, causal=causal)
    dmask = torch.where(ref(msk) > 0, 1.0, 0.3)
    ym = O.conv2d_nhwc(ref(x) * dmask, ref(w), None, dil_t=dil, causal=causal)
    wm = ref(w, True)
    O.conv2d_nhwc(ref(x) * dmask, wm, None, dil_t=dil, causal=causal).backward(dy)

    def run(on):
        ops.conv2d_mfma(on)
        try:
            with ops._hip.KernelTimer() as kt:
                xd, wd, bd = dev(x, True), dev(w, True), dev(b, True)
                yd = ops.conv2d(ops.Lazy(xd, lrelu=True), wd, bd, dil_t=dil, pad_mode=pm)
                yd.backward(dev(dy))
                yad = ops.conv2d(ops.Lazy(dev(x), dev(sc), dev(sh), lrelu=True), dev(w), dev(b), dil_t=dil, pad_mode=pm)
                ymd = ops._conv2d_fwd_raw(dev(x), dev(w), None, None, None, dev(msk), ops.IN_MASKMUL, 0.3, dil, pm)
                _, dwm, _, _, _ = ops._conv2d_bwd_raw(dev(dy), dev(x), dev(w), None, None, dev(msk), ops.IN_MASKMUL, 0.3, dil, pm,
                                                      False, True, False, False)
            return (yd, xd.grad, wd.grad, bd.grad, yad, ymd, dwm), [r[0] for r in kt.records]
        finally:
            ops.conv2d_mfma(None)

    res_m, names_m = run(True)
    res_s, names_s = run(False)
    if dil == 1 and not causal:
        # dx + dW + dbias of the layer are ONE launch (c2m::bwd_ws_kernel, kind 1)
        assert names_m.count('ptts_conv2d_mfma_fwd') == 3 and names_m.count('ptts_conv2d_mfma_wgrad_partials') == 1 and \
            names_m.count('ptts_conv2d_mfma_bwd_fused') == 1, names_m
    else:
        assert names_m.count('ptts_conv2d_mfma_fwd') == 4 and names_m.count('ptts_conv2d_mfma_wgrad_partials') == 2, names_m
    assert 'ptts_conv2d_fwd' not in names_m and not any(n.startswith('ptts_conv2d_mfma') for n in names_s)
    wants = (yr, xr.grad, wr.grad, br.grad, ya, ym, wm.grad)
    tols = ((RT, AT), (RT, AT), (2e-4, 1e-4), (2e-4, 1e-4), (RT, AT), (RT, AT), (2e-4, 1e-4))
    for nm, got_m, got_s, want, (rt, at) in zip(('y', 'dx', 'dw', 'db', 'y (affine)', 'y (maskmul)', 'dw (maskmul)'), res_m, res_s, wants, tols):
        close(got_m, want, rtol=rt, atol=at, what=nm + ' [matrix cores]')
        close(got_s, want, rtol=rt, atol=at, what=nm + ' [fp32 stencil]')


@pytest.mark.parametrize('case', [(3, 100, 65), (2, 16, 65), (1, 1, 65), (2, 37, 130), (5, 400, 65), (1, 50, 129), (4, 7, 33), (300, 17, 3), (17, 270, 65), (70, 59, 68), (64, 400, 65)])
def test_conv2d_mfma_wave_specialised_form_is_bit_identical(ops, case):
    """The default dilation-1 fp32 forward kernel (c2m::fwd_ws_kernel: eight waves, four stage, four multiply) against the
    four-wave form it replaced (c2m::fwd_kernel, forced by bit 16 of ptts_conv2d_mfma_debug): same tiles, passes and arithmetic,
    so forward (LeakyReLU and BatchNorm-affine inputs), masked forward and backward data (with the output mask) are equal bit for bit."""
    B, T, F = case
    g = gen(77)
    x = torch.randn(B, T, F, 4, generator=g).cuda()
    w = (torch.randn(5, 5, 4, 4, generator=g) * 0.3).cuda()
    b = torch.randn(4, generator=g).cuda()
    sc = (torch.rand(4, generator=g) + 0.5).cuda()
    sh = (torch.randn(4, generator=g) * 0.3).cuda()
    dy = torch.randn(B, T, F, 4, generator=g).cuda()
    msk = torch.randn(B, T, F, 4, generator=g).cuda()
    lib = ops._hip.lib()

    def run():
        y = ops._conv2d_fwd_raw(x, w, b, None, None, None, ops.IN_LRELU, 0.3, 1, ops.PAD_SAME)
        ya = ops._conv2d_fwd_raw(x, w, b, sc, sh, None, ops.IN_LRELU, 0.3, 1, ops.PAD_SAME)
        ym = ops._conv2d_fwd_raw(x, w, None, None, None, msk, ops.IN_MASKMUL, 0.3, 1, ops.PAD_SAME)
        dx, _, _, _, _ = ops._conv2d_bwd_raw(dy, x, w, None, None, None, ops.IN_LRELU, 0.3, 1, ops.PAD_SAME, True, False, False, False)
        torch.cuda.synchronize()
        return y, ya, ym, dx
    res_ws = run()
    lib.ptts_conv2d_mfma_debug(1 << 16, None)
    try:
        res_4 = run()
    finally:
        lib.ptts_conv2d_mfma_debug(0, None)
    for nm, a, c in zip(('y', 'y (affine)', 'y (maskmul)', 'dx'), res_ws, res_4):
        assert torch.equal(a, c), nm


@pytest.mark.parametrize('case', [(3, 100, 65), (2, 16, 65), (1, 1, 65), (2, 37, 130), (5, 400, 65), (1, 50, 129), (4, 7, 33), (300, 17, 3), (2, 33, 71), (17, 270, 65), (70, 59, 68), (64, 400, 65)])
def test_conv2d_mfma_fused_backward_launches(ops, case):
    """The fused backward launches (c2m::bwd_ws_kernel, round 4): dx + dW + dbias of a layer (kind 1) and the second-order sweep's
    masked forward + dW (kind 2), each ONE launch sharing the staging of dy / u, against the separate launches they replace
    (ops.conv2d_fused(False)): dx and cot_dy bit for bit (same tiles, passes and MFMA order), the weight and bias gradients --
    partial sums grouped per 256 instead of 512 workgroups -- to fp32 summation accuracy, and against an fp64 oracle.  Shapes:
    ragged T and F, F > 68 (two bin blocks), F = 3 (one group: the zeroed half of an odd piece's last pair), many small
    utterances, and BASELINE's [64,400,65]."""
    B, T, F = case
    g = gen(79)
    x = torch.randn(B, T, F, 4, generator=g).cuda()
    u = torch.randn(B, T, F, 4, generator=g).cuda()
    dy = torch.randn(B, T, F, 4, generator=g).cuda()
    w = (torch.randn(5, 5, 4, 4, generator=g) * 0.3).cuda()

    def run(fused):
        ops.conv2d_fused(fused)
        try:
            with ops._hip.KernelTimer() as kt:
                dx, dw, db, _, _ = ops._conv2d_bwd_raw(dy, x, w, None, None, None, ops.IN_LRELU, 0.3, 1, ops.PAD_SAME, True, True, True, False)
                # second order through the autograd Function that runs it in the optimiser: backward of the backward-data pass
                dyr, wr = dy.clone().requires_grad_(True), w.clone().requires_grad_(True)
                gx = ops.Conv2dBwdDataFn.apply(dyr, x, wr, ops.IN_LRELU, 0.3, 1, ops.PAD_SAME, None, 3)
                gx.backward(u)
            torch.cuda.synchronize()
            return (dx, dw, db, dyr.grad, wr.grad), [r[0] for r in kt.records]
        finally:
            ops.conv2d_fused(None)
    (dx1, dw1, db1, c1, cw1), n1 = run(True)
    (dx0, dw0, db0, c0, cw0), n0 = run(False)
    assert n1.count('ptts_conv2d_mfma_bwd_fused') == 2 and 'ptts_conv2d_mfma_wgrad_partials' not in n1, n1
    assert 'ptts_conv2d_mfma_bwd_fused' not in n0 and n0.count('ptts_conv2d_mfma_wgrad_partials') == 2, n0
    assert torch.equal(dx1, dx0) and torch.equal(c1, c0)
    for nm, a, b in (('dw', dw1, dw0), ('db', db1, db0), ('cot_w', cw1, cw0)):
        assert float((a - b).norm()) <= 3e-6 * float(b.norm()) + 1e-6, (nm, float((a - b).norm() / b.norm()))
    # fp64 oracle (chunked over the batch)
    dw64 = torch.zeros(5, 5, 4, 4, dtype=torch.float64); db64 = torch.zeros(4, dtype=torch.float64); cw64 = torch.zeros(5, 5, 4, 4, dtype=torch.float64)
    step = max(1, min(B, (1 << 19) // (T * F)))
    for b0 in range(0, B, step):
        sl = slice(b0, b0 + step)
        xq = x[sl].double().cpu().requires_grad_(True); wq = w.double().cpu().requires_grad_(True); bq = torch.zeros(4, dtype=torch.float64, requires_grad=True)
        O.conv2d_nhwc(O.lrelu(xq), wq, bq).backward(dy[sl].double().cpu())
        dw64 += wq.grad; db64 += bq.grad
        close(dx1[sl], xq.grad, rtol=RT, atol=AT, what='dx (fused)')
        m = torch.where(x[sl].double().cpu() > 0, 1.0, 0.3)
        wq2 = w.double().cpu().requires_grad_(True)
        O.conv2d_nhwc(u[sl].double().cpu() * m, wq2, None).backward(dy[sl].double().cpu())
        cw64 += wq2.grad
        close(c1[sl], O.conv2d_nhwc(u[sl].double().cpu() * m, w.double().cpu(), None), rtol=RT, atol=AT, what='cot_dy (fused)')
    for nm, a, b in (('dw', dw1, dw64), ('db', db1, db64), ('cot_w', cw1, cw64)):
        assert float((a.double().cpu() - b).abs().max()) <= 2e-5 * float(b.abs().mean()) + 1e-5, (nm, float((a.double().cpu() - b).abs().max() / b.abs().mean()))


def test_conv2d_mfma_hand_off_timeout_is_reported_not_silent(ops):
    """The wave-specialised forward hands its plane buffers over by LDS counters with BOUNDED polls (csrc/conv2d_mfma.hip).  A poll
    that runs out must not pass silently: the wave stores its code into the sticky device status word, the C ABI returns PTTS_EDEVICE
    from the next call on, and the word stays clear in normal operation.  The time-out is forced once through the debug hook
    (bit 7 of ptts_conv2d_mfma_debug: the multiplying waves wait, briefly, for a count that never comes)."""
    _hip = ops._hip
    g = gen(78)
    x = torch.randn(4, 100, 65, 4, generator=g).cuda()
    w = (torch.randn(5, 5, 4, 4, generator=g) * 0.3).cuda()
    lib = _hip.lib()
    _hip.clear_status()
    y0 = ops._conv2d_fwd_raw(x, w, None, None, None, None, ops.IN_LRELU, 0.3, 1, ops.PAD_SAME)
    torch.cuda.synchronize()
    _hip.check_status()                                   # a normal launch leaves the word clear
    lib.ptts_conv2d_mfma_debug(128, None)
    try:
        ops._conv2d_fwd_raw(x, w, None, None, None, None, ops.IN_LRELU, 0.3, 1, ops.PAD_SAME)     # this launch times out on the device
        torch.cuda.synchronize()
    finally:
        lib.ptts_conv2d_mfma_debug(0, None)
    with pytest.raises(_hip.HipLibraryError, match='hand-off'):
        _hip.check_status()
    with pytest.raises(_hip.HipLibraryError, match='hand-off'):      # sticky: the next launch of the family is refused
        ops._conv2d_fwd_raw(x, w, None, None, None, None, ops.IN_LRELU, 0.3, 1, ops.PAD_SAME)
    _hip.clear_status()
    y1 = ops._conv2d_fwd_raw(x, w, None, None, None, None, ops.IN_LRELU, 0.3, 1, ops.PAD_SAME)
    torch.cuda.synchronize()
    _hip.check_status()
    assert torch.equal(y0, y1)


@pytest.mark.parametrize('in16,out16', [(False, True), (True, True), (True, False)])
def test_conv2d_bf16_storage_layer(ops, in16, out16):
    """One 4 -> 4 channel 5x5 layer of the bf16-storage path (ops.conv2d(..., bf16=...), csrc/conv2d_mfma.hip with one
    plane): input map stored as fp32 or bf16, result stored as bf16 or fp32.  Forward against the fp64 oracle with the
    same roundings -- activation to bf16 after the LeakyReLU, the kernel's bf16 copy, exact products, the result rounded
    once when stored as bf16: at most one bf16 ulp (2^-8) apart where the sums differ in order.  Backward: dx has the
    input's storage type, dw / db are fp32; against the oracle's fp64 backward of the same rounded forward, at the bf16
    budget (2^-8 per stored value)."""
    B, T, F = 2, 40, 65
    g = gen(71)
    x = torch.randn(B, T, F, 4, generator=g, dtype=torch.float64)
    if in16:
        x = x.to(torch.bfloat16).double()          # a map that already lies in HBM as bf16
    w = torch.randn(5, 5, 4, 4, generator=g, dtype=torch.float64) * 0.2
    b = torch.randn(4, generator=g, dtype=torch.float64)
    dy = torch.randn(B, T, F, 4, generator=g, dtype=torch.float64)
    xr, wr, br = ref(x, True), ref(w, True), ref(b, True)
    yr = O.conv2d_nhwc(O.bf16_st(O.lrelu(xr)), O.bf16_st(wr), br)
    if out16:
        yr = O.bf16_st(yr)
        dy = dy.to(torch.bfloat16).double()
    yr.backward(dy)
    xd = x.to(torch.bfloat16 if in16 else torch.float32).cuda().requires_grad_(True)
    wd, bd = dev(w, True), dev(b, True)
    yd = ops.conv2d(ops.Lazy(xd, lrelu=True), wd, bd, bf16='out16' if out16 else 'out32')
    assert yd.dtype == (torch.bfloat16 if out16 else torch.float32)
    ulp = 2.0 ** -8
    err = (yd.double().cpu() - yr.detach()).abs()
    assert float((err / (yr.detach().abs() + 1e-2)).max()) < (1.1 * ulp if out16 else 1e-4), float((err / (yr.detach().abs() + 1e-2)).max())
    yd.backward(dy.to(yd.dtype).cuda())
    assert xd.grad.dtype == xd.dtype and wd.grad.dtype == torch.float32 and bd.grad.dtype == torch.float32
    gx = float((xd.grad.double().cpu() - xr.grad).norm() / xr.grad.norm())
    gwt = float((wd.grad.double().cpu() - wr.grad).norm() / wr.grad.norm())
    assert gx < 2.0 ** -8 and gwt < 2.0 ** -8, (gx, gwt)       # dx: rounded dy planes and (bf16 x) a rounded store
    close(bd.grad, br.grad, rtol=2e-4, atol=1e-3, what='db')     # db sums the raw dy values: fp32 accuracy


@pytest.mark.parametrize('case', [(2, 12, 9, 2, 3, 3), (2, 40, 65, 4, 5, 5)])
def test_conv2d_stack_second_order(ops, case):
    """The gradient-penalty pattern: differentiate ||d(sum v)/dx||^2 w.r.t. the kernels of a
    (conv -> lrelu -> conv -> lrelu -> conv) stack (optimizertts_wgan.py:53-68)."""
    B, T, F, C, KT, KF = case
    g = gen(2)
    x = torch.randn(B, T, F, 1, generator=g, dtype=torch.float64)
    ws = [torch.randn(KT, KF, 1, C, generator=g, dtype=torch.float64) * 0.4,
          torch.randn(KT, KF, C, C, generator=g, dtype=torch.float64) * 0.3,
          torch.randn(KT, KF, C, 1, generator=g, dtype=torch.float64) * 0.3]
    bs = [torch.randn(s.shape[-1], generator=g, dtype=torch.float64) * 0.2 for s in ws]

    xr = ref(x, True); wr = [ref(w, True) for w in ws]; br = [ref(b, True) for b in bs]
    h = O.conv2d_nhwc(xr, wr[0], br[0])
    h = O.conv2d_nhwc(O.lrelu(h), wr[1], br[1])
    v = O.conv2d_nhwc(O.lrelu(h), wr[2], br[2])
    gr = torch.autograd.grad(v.sum(), xr, create_graph=True)[0]
    lossr = (gr * gr).sum() + v.mean()
    lossr.backward()

    xd = dev(x, True); wd = [dev(w, True) for w in ws]; bd = [dev(b, True) for b in bs]
    h = ops.conv2d(xd, wd[0], bd[0])
    h = ops.conv2d(ops.Lazy(h, lrelu=True), wd[1], bd[1])
    vd = ops.conv2d(ops.Lazy(h, lrelu=True), wd[2], bd[2])
    with ops.input_grad_only():
        gd = torch.autograd.grad(vd, xd, grad_outputs=torch.ones_like(vd), create_graph=True)[0]
    close(gd, gr, what='g')
    lossd = (gd * gd).sum() + vd.mean()
    lossd.backward()
    for i in range(3):
        close(wd[i].grad, wr[i].grad, rtol=3e-4, atol=1e-4, what='dw%d' % i)
        close(bd[i].grad, br[i].grad, rtol=3e-4, atol=1e-4, what='db%d' % i)


GEMM_CASES = [
    # M, N, K, transA, transB
    (128, 128, 16, 0, 0), (130, 70, 33, 0, 0), (257, 129, 100, 0, 1), (64, 300, 77, 1, 0), (50, 40, 30, 1, 1),
    (256, 256, 4096, 1, 0),      # split-K path (weight-gradient shape)
    (1000, 1, 256, 0, 0), (256, 1, 1000, 1, 0),
]


@pytest.mark.parametrize('case', GEMM_CASES)
def test_gemm(ops, case):
    M, N, K, ta, tb = case
    g = gen(3)
    A = torch.randn(M, K, generator=g, dtype=torch.float64)
    Bm = torch.randn(K, N, generator=g, dtype=torch.float64)
    bias = torch.randn(N, generator=g, dtype=torch.float64)
    want = A @ Bm + bias
    As = dev(A.t() if ta else A)
    Bs = dev(Bm.t() if tb else Bm)
    C = torch.empty(M, N, dtype=torch.float32, device='cuda')
    ops.gemm_raw(As, Bs, C, M, N, K, transA=ta, transB=tb, bias=dev(bias))
    close(C, want, rtol=2e-4, atol=2e-4 * math.sqrt(K), what='C')
    # accumulate
    ops.gemm_raw(As, Bs, C, M, N, K, transA=ta, transB=tb, accumulate=1)
    close(C, 2 * want - bias, rtol=2e-4, atol=4e-4 * math.sqrt(K), what='C+=')


@pytest.mark.parametrize('case', [(300, 200, 5000), (256, 256, 25600), (70, 260, 900)])
def test_gemm_weight_gradient_with_fused_bias_gradient(ops, case):
    """dW = a^T.dy and db = column sums of dy from ONE product (colsum_b): split and unsplit, ragged tiles."""
    M, N, K = case
    g = gen(18)
    A = torch.randn(K, M, generator=g, dtype=torch.float64)
    D = torch.randn(K, N, generator=g, dtype=torch.float64) + 0.3
    W = torch.empty(M, N, dtype=torch.float32, device='cuda')
    db = torch.full((N,), 5.0, dtype=torch.float32, device='cuda')      # must be overwritten, not added to
    ops.gemm_raw(dev(A), dev(D), W, M, N, K, transA=1, lda=M, rows_per_seg=K, mode=ops.IN_LRELU, colsum_b=db)
    close(W, O.lrelu(A).t() @ D, rtol=2e-4, atol=2e-4 * math.sqrt(K), what='dW')
    close(db, D.sum(0), rtol=2e-4, atol=2e-4 * math.sqrt(K), what='db')


def test_gemm_transforms(ops):
    g = gen(4)
    M, N, K = 200, 96, 50
    A = torch.randn(M, K, generator=g, dtype=torch.float64)
    Bm = torch.randn(K, N, generator=g, dtype=torch.float64)
    sc = torch.rand(K, generator=g, dtype=torch.float64) + 0.5
    sh = torch.randn(K, generator=g, dtype=torch.float64)
    msk = torch.randn(M, K, generator=g, dtype=torch.float64)
    C = torch.empty(M, N, dtype=torch.float32, device='cuda')
    ops.gemm_raw(dev(A), dev(Bm), C, M, N, K, mode=ops.IN_LRELU, scale=dev(sc), shift=dev(sh))
    close(C, O.lrelu(A * sc + sh) @ Bm, rtol=2e-4, atol=1e-3, what='lrelu-affine')
    ops.gemm_raw(dev(A), dev(Bm), C, M, N, K, mode=ops.IN_MASKMUL, mask_src=dev(msk))
    close(C, (A * torch.where(msk > 0, 1.0, 0.3)) @ Bm, rtol=2e-4, atol=1e-3, what='maskmul')
    # transA with the transform on the stored column (weight gradient of a fused layer)
    D = torch.randn(M, N, generator=g, dtype=torch.float64)
    W = torch.empty(K, N, dtype=torch.float32, device='cuda')
    ops.gemm_raw(dev(A), dev(D), W, K, N, M, transA=1, lda=K, rows_per_seg=M, mode=ops.IN_LRELU, scale=dev(sc), shift=dev(sh))
    close(W, O.lrelu(A * sc + sh).t() @ D, rtol=2e-4, atol=1e-3, what='transA-lrelu')


def test_gemm_thin_shapes(ops):
    """The thin-product kernels (thin.hip): N<=4 heads, K<=4 outer products with the output mask, and 1-2 weighted
    column sums, each with the fused transforms and strided C/mask views the Dense layers use."""
    g = gen(14)
    M, K = 3000, 256
    A = torch.randn(M, K, generator=g, dtype=torch.float64)
    msk = torch.randn(M, K, generator=g, dtype=torch.float64)
    sc = torch.rand(K, generator=g, dtype=torch.float64) + 0.5
    sh = torch.randn(K, generator=g, dtype=torch.float64) * 0.3
    for N in (1, 2, 3, 4):
        Bm = torch.randn(K, N, generator=g, dtype=torch.float64)
        bias = torch.randn(N, generator=g, dtype=torch.float64)
        for tb in (0, 1):
            C = torch.empty(M, N, dtype=torch.float32, device='cuda')
            ops.gemm_raw(dev(A), dev(Bm.t() if tb else Bm), C, M, N, K, transB=tb, bias=dev(bias), mode=ops.IN_LRELU,
                         scale=dev(sc), shift=dev(sh))
            close(C, O.lrelu(A * sc + sh) @ Bm + bias, rtol=2e-4, atol=2e-3, what='gemv N=%d tb=%d' % (N, tb))
        C = torch.empty(M, N, dtype=torch.float32, device='cuda')
        ops.gemm_raw(dev(A), dev(Bm), C, M, N, K, mode=ops.IN_MASKMUL, mask_src=dev(msk))
        close(C, (A * torch.where(msk > 0, 1.0, 0.3)) @ Bm, rtol=2e-4, atol=2e-3, what='gemv maskmul N=%d' % N)
    # N=4 remainder written into columns 256.. of a 260-wide C with the output mask laid out alike
    Bm = torch.randn(K, 4, generator=g, dtype=torch.float64)
    om = torch.randn(M, 260, generator=g, dtype=torch.float64)
    Cw = torch.zeros(M, 260, dtype=torch.float32, device='cuda')
    omd = dev(om)
    ops.gemm_raw(dev(A), dev(Bm), Cw[:, 256:], M, 4, K, ldc=260, out_mask=omd[:, 256:])
    close(Cw[:, 256:], (A @ Bm) * torch.where(om[:, 256:] > 0, 1.0, 0.3), rtol=2e-4, atol=2e-3, what='gemv strided mask')
    assert float(Cw[:, :256].abs().max()) == 0.0
    # K<=4 outer products
    for Kt in (1, 2, 3, 4):
        At = torch.randn(M, Kt, generator=g, dtype=torch.float64)
        W = torch.randn(K, Kt, generator=g, dtype=torch.float64)     # used transposed: C = At @ W^T
        C = torch.empty(M, K, dtype=torch.float32, device='cuda')
        ops.gemm_raw(dev(At), dev(W), C, M, K, Kt, transB=1, out_mask=dev(msk))
        close(C, (At @ W.t()) * torch.where(msk > 0, 1.0, 0.3), rtol=2e-4, atol=1e-4, what='thinK %d' % Kt)
        ops.gemm_raw(dev(At), dev(W.t()), C, M, K, Kt, accumulate=1)
        close(C, (At @ W.t()) * torch.where(msk > 0, 1.0, 0.3) + At @ W.t(), rtol=2e-4, atol=2e-4, what='thinK acc %d' % Kt)
    # weighted column sums (weight gradient of a 1-2 wide head)
    for N in (1, 2):
        D = torch.randn(M, N, generator=g, dtype=torch.float64)
        W = torch.empty(K, N, dtype=torch.float32, device='cuda')
        ops.gemm_raw(dev(A), dev(D), W, K, N, M, transA=1, lda=K, rows_per_seg=M, mode=ops.IN_LRELU, scale=dev(sc), shift=dev(sh))
        close(W, O.lrelu(A * sc + sh).t() @ D, rtol=2e-4, atol=5e-3, what='wcol lrelu N=%d' % N)
        ops.gemm_raw(dev(A), dev(D), W, K, N, M, transA=1, lda=K, rows_per_seg=M, mode=ops.IN_MASKMUL, mask_src=dev(msk), accumulate=1)
        close(W, O.lrelu(A * sc + sh).t() @ D + (A * torch.where(msk > 0, 1.0, 0.3)).t() @ D, rtol=2e-4, atol=1e-2,
              what='wcol maskmul acc N=%d' % N)
    # ... at the critic's head (25 600 frames) and with a column count that is no multiple of 256 (the four-column kernel's tail lanes)
    for M2, K2 in ((25600, 256), (5003, 260), (2048, 516)):
        A2 = torch.randn(M2, K2, generator=g, dtype=torch.float64)
        D2 = torch.randn(M2, 1, generator=g, dtype=torch.float64)
        W2 = torch.empty(K2, 1, dtype=torch.float32, device='cuda')
        ops.gemm_raw(dev(A2), dev(D2), W2, K2, 1, M2, transA=1, lda=K2, rows_per_seg=M2, mode=ops.IN_LRELU)
        ref = O.lrelu(A2).t() @ D2
        close(W2, ref, rtol=2e-4, atol=2e-4 * float(ref.abs().max()), what='wcol at [%d, %d]' % (M2, K2))


@pytest.mark.parametrize('mode', ['none', 'lrelu', 'affine'])
@pytest.mark.parametrize('shape', [(2, 8, 11, 5), (3, 50, 300, 130)])
def test_dense_forward_backward(ops, mode, shape):
    B, T, K, N = shape
    g = gen(5)
    x = torch.randn(B, T, K, generator=g, dtype=torch.float64)
    w = torch.randn(K, N, generator=g, dtype=torch.float64) / math.sqrt(K)
    b = torch.randn(N, generator=g, dtype=torch.float64)
    sc = torch.rand(K, generator=g, dtype=torch.float64) + 0.5
    sh = torch.randn(K, generator=g, dtype=torch.float64) * 0.3
    dy = torch.randn(B, T, N, generator=g, dtype=torch.float64)
    xr, wr, br, scr, shr = ref(x, True), ref(w, True), ref(b, True), ref(sc, True), ref(sh, True)
    a = xr if mode == 'none' else (O.lrelu(xr) if mode == 'lrelu' else O.lrelu(xr * scr + shr))
    yr = O.dense(a, wr, br)
    yr.backward(dy)
    xd, wd, bd, scd, shd = dev(x, True), dev(w, True), dev(b, True), dev(sc, True), dev(sh, True)
    v = xd if mode == 'none' else (ops.Lazy(xd, lrelu=True) if mode == 'lrelu' else ops.Lazy(xd, scd, shd, lrelu=True))
    yd = ops.dense(v, wd, bd)
    close(yd, yr, rtol=2e-4, atol=1e-4, what='y')
    yd.backward(dev(dy))
    close(xd.grad, xr.grad, rtol=2e-4, atol=1e-4, what='dx')
    close(wd.grad, wr.grad, rtol=2e-4, atol=2e-4, what='dw')
    close(bd.grad, br.grad, rtol=2e-4, atol=2e-4, what='db')
    if mode == 'affine':
        close(scd.grad, scr.grad, rtol=2e-4, atol=2e-4, what='dscale')
        close(shd.grad, shr.grad, rtol=2e-4, atol=2e-4, what='dshift')


def test_dense_second_order(ops):
    g = gen(6)
    B, T, K, H = 2, 9, 7, 12
    x = torch.randn(B, T, K, generator=g, dtype=torch.float64)
    w1 = torch.randn(K, H, generator=g, dtype=torch.float64) * 0.5
    w2 = torch.randn(H, H, generator=g, dtype=torch.float64) * 0.5
    w3 = torch.randn(H, 1, generator=g, dtype=torch.float64) * 0.5
    b1 = torch.randn(H, generator=g, dtype=torch.float64) * 0.2
    xr = ref(x, True); w1r, w2r, w3r, b1r = ref(w1, True), ref(w2, True), ref(w3, True), ref(b1, True)
    v = O.dense(O.lrelu(O.dense(O.lrelu(O.dense(xr, w1r, b1r)), w2r)), w3r)
    gr = torch.autograd.grad(v.sum(), xr, create_graph=True)[0]
    ((gr * gr).sum()).backward()
    xd = dev(x, True); w1d, w2d, w3d, b1d = dev(w1, True), dev(w2, True), dev(w3, True), dev(b1, True)
    h = ops.dense(xd, w1d, b1d)
    h = ops.dense(ops.Lazy(h, lrelu=True), w2d)
    vd = ops.dense(ops.Lazy(h, lrelu=True), w3d)
    with ops.input_grad_only():
        gd = torch.autograd.grad(vd, xd, grad_outputs=torch.ones_like(vd), create_graph=True)[0]
    close(gd, gr, rtol=2e-4, atol=1e-4, what='g')
    ((gd * gd).sum()).backward()
    close(w1d.grad, w1r.grad, rtol=3e-4, atol=1e-4, what='dw1')
    close(w2d.grad, w2r.grad, rtol=3e-4, atol=1e-4, what='dw2')
    close(w3d.grad, w3r.grad, rtol=3e-4, atol=1e-4, what='dw3')
    assert b1d.grad is None or float(b1d.grad.abs().max()) == 0.0


@pytest.mark.parametrize('case', [(2, 30, 5, 6, 3), (2, 30, 5, 6, 4), (3, 60, 37, 20, 21)])
def test_conv1d(ops, case):
    B, T, Cin, N, KW = case
    g = gen(7)
    x = torch.randn(B, T, Cin, generator=g, dtype=torch.float64)
    w = torch.randn(KW, Cin, N, generator=g, dtype=torch.float64) * 0.2
    b = torch.randn(N, generator=g, dtype=torch.float64)
    dy = torch.randn(B, T, N, generator=g, dtype=torch.float64)
    xr, wr, br = ref(x, True), ref(w, True), ref(b, True)
    yr = O.conv1d_ntc(O.lrelu(xr), wr, br)
    yr.backward(dy)
    xd, wd, bd = dev(x, True), dev(w, True), dev(b, True)
    yd = ops.conv1d(ops.Lazy(xd, lrelu=True), wd, bd)
    close(yd, yr, rtol=2e-4, atol=2e-4, what='y')
    yd.backward(dev(dy))
    close(xd.grad, xr.grad, rtol=2e-4, atol=2e-4, what='dx')
    close(wd.grad, wr.grad, rtol=2e-4, atol=3e-4, what='dw')
    close(bd.grad, br.grad, rtol=2e-4, atol=3e-4, what='db')


@pytest.mark.parametrize('case', [
    # B, T, Cin, N, KW, bias, nonzero dW in the buffer beforehand
    (8, 520, 150, 96, 3, True), (8, 520, 150, 96, 5, False), (8, 520, 150, 96, 21, True),
    (5, 830, 70, 32, 21, True),           # one 32-output tile, ragged 64-channel tile, qsteps not divisible by the split
    (11, 400, 601, 256, 21, True),        # the context Conv1D of BASELINE configs[1]: 10 channel tiles, 8 output tiles
])
def test_conv1d_weight_gradient_frame_major(ops, case):
    """The default weight-gradient path of the context Conv1D for B*T >= 4096 (csrc/conv1d_wgrad.hip: ptts_transpose_frames +
    ptts_conv1d_wgrad_t, exact fp32 over frame-major operands) against the fp64 oracle on the FULL dW and db; ragged channel
    counts (c0 > 0 tiles with a partial last tile), all three KW instantiations, with and without a bias, and an A/B run
    of the stream-K product it replaced (PTTS_WGRAD_T=0).  The kernel is asserted to have run."""
    B, T, Cin, N, KW, has_b = case
    assert B * T >= 4096
    g = gen(41)
    x = torch.randn(B, T, Cin, generator=g, dtype=torch.float64).float()
    w = (torch.randn(KW, Cin, N, generator=g, dtype=torch.float64) * (1.0 / (KW * Cin) ** 0.5)).float()
    b = torch.randn(N, generator=g, dtype=torch.float64).float() if has_b else None
    dy = torch.randn(B, T, N, generator=g, dtype=torch.float64).float()
    wr = w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if has_b else None
    O.conv1d_ntc(x.double(), wr, br).backward(dy.double())

    def run(enabled):
        old = ops._C1WgradT.enabled
        ops._C1WgradT.enabled = enabled
        ops._C1WgradT.clear()
        try:
            wd = w.cuda().requires_grad_(True)
            bd = b.cuda().requires_grad_(True) if has_b else None
            with ops._hip.KernelTimer() as kt:
                ops.conv1d(x.cuda(), wd, bd).backward(dy.cuda())
            return wd.grad.cpu().double(), (bd.grad.cpu().double() if has_b else None), [r[0] for r in kt.records]
        finally:
            ops._C1WgradT.enabled = old
            ops._C1WgradT.clear()

    ops.conv1d_split(False)
    try:
        dw_t, db_t, names_t = run(True)
        dw_k, db_k, names_k = run(False)
    finally:
        ops.conv1d_split(None)
    assert 'ptts_conv1d_wgrad_t' in names_t and 'ptts_transpose_frames' in names_t, names_t
    assert 'ptts_conv1d_wgrad_t' not in names_k
    gs = wr.grad.abs().mean()
    e_t = ((dw_t - wr.grad).abs().max() / gs).item()
    e_k = ((dw_k - wr.grad).abs().max() / gs).item()
    # tolerance: fp32 accumulation over B*T <= 4400 frames, the same bound as the stream-K product
    assert e_t < 3e-5, (e_t, e_k)
    assert e_t < 4 * max(e_k, 2e-6), (e_t, e_k)
    if has_b:
        close(db_t, br.grad, rtol=2e-4, atol=3e-4, what='db (frame-major kernel)')
        close(db_k, br.grad, rtol=2e-4, atol=3e-4, what='db (stream-K product)')


def test_split3_planes_are_exact(ops):
    """The three bf16 planes of the split pass add up to the fp32 operand EXACTLY (3 x 8 significant bits = 24), sit in
    the documented 32-channel-block layout, and are zero in the time padding and in the channels C..Cp-1."""
    from percivaltts_amd.ops import call, ptr, stream
    g = gen(31)
    B, T, C, KW, N = 3, 37, 45, 5, 128
    x = (torch.randn(B, T, C, generator=g) * torch.exp(4 * torch.randn(B, T, C, generator=g))).cuda()     # wide dynamic range
    w = (torch.randn(KW, C, N, generator=g) * 0.1).cuda()
    pl, pr, Cp = 2, 2, 64
    xp = torch.full((3, Cp // 32, B, T + pl + pr, 32), 7.0, dtype=torch.bfloat16, device='cuda')
    wp = torch.full((3, Cp // 32, N, KW, 32), 7.0, dtype=torch.bfloat16, device='cuda')
    call('ptts_split3_frames', ptr(x), ptr(xp[0]), ptr(xp[1]), ptr(xp[2]), B, T, C, pl, pr, Cp, stream())
    call('ptts_split3_weight_t', ptr(w), ptr(wp[0]), ptr(wp[1]), ptr(wp[2]), KW, C, N, Cp, stream())
    xs = (xp[0].float() + xp[1].float() + xp[2].float()).permute(1, 2, 0, 3).reshape(B, T + pl + pr, Cp)
    assert torch.equal(xs[:, pl:pl + T, :C], x)
    assert (xs[:, :pl] == 0).all() and (xs[:, pl + T:] == 0).all() and (xs[..., C:] == 0).all()
    ws = (wp[0].float() + wp[1].float() + wp[2].float()).permute(1, 2, 0, 3).reshape(N, KW, Cp)
    assert torch.equal(ws[..., :C].permute(1, 2, 0), w)
    assert (ws[..., C:] == 0).all()
    # each plane is what rounding the running remainder to bf16 gives
    assert torch.equal(xp[0].permute(1, 2, 0, 3).reshape(B, T + pl + pr, Cp)[:, pl:pl + T, :C], x.to(torch.bfloat16))
    # bit for bit the oracle's restatement of the split (oracle.np_split3_bf16), all three planes, both layouts
    o1, o2, o3 = O.np_split3_bf16(x.cpu().numpy())
    for plane, ref_ in zip(xp, (o1, o2, o3)):
        got = plane.float().permute(1, 2, 0, 3).reshape(B, T + pl + pr, Cp)[:, pl:pl + T, :C].cpu().numpy()
        assert (got == ref_).all()
    q1, q2, q3 = O.np_split3_bf16(w.cpu().numpy())
    for plane, ref_ in zip(wp, (q1, q2, q3)):
        got = plane.float().permute(1, 2, 0, 3).reshape(N, KW, Cp)[..., :C].permute(1, 2, 0).cpu().numpy()
        assert (got == ref_).all()
    # frame-major planes of the weight-gradient kernel
    Tp = T + pl + pr
    Pp = ops._C1Split.plane_len(B, T, pl + pr + 1)
    xt, Crows = ops._C1Split.transposed(x, B, T, C, pl, Tp, Pp)
    for plane, ref_ in zip(xt, (o1, o2, o3)):
        full = plane.float()[:C, :B * Tp].reshape(C, B, Tp).permute(1, 2, 0)
        assert (full[:, pl:pl + T].cpu().numpy() == ref_).all()
        assert (full[:, :pl] == 0).all() and (full[:, pl + T:] == 0).all()
        assert (plane.float()[C:] == 0).all() and (plane.float()[:, B * Tp:] == 0).all()


@pytest.mark.parametrize('case', [(2, 130, 37, 128, 5), (3, 129, 70, 256, 21), (1, 50, 33, 128, 3), (2, 300, 601, 256, 21)])
def test_conv1d_bf16x6_split_product(ops, case):
    """Context Conv1D forward as a bf16x6 split product (csrc/split.hip) against the fp64 oracle: the error must be of the
    size of the fp32 MFMA kernel's own, and far below what a plain bf16 product would give (2^-9 relative).  The cases
    put an utterance boundary inside a 128-frame tile, an M edge tile, a single short utterance, and the real K."""
    B, T, Cin, N, KW = case
    g = gen(32)
    x = torch.randn(B, T, Cin, generator=g, dtype=torch.float64)
    w = torch.randn(KW, Cin, N, generator=g, dtype=torch.float64) * (1.0 / (KW * Cin) ** 0.5)
    b = torch.randn(N, generator=g, dtype=torch.float64)
    x32, w32, b32 = x.float(), w.float(), b.float()          # the fp32 operands both kernels see
    yr = O.conv1d_ntc(x32.double(), w32.double(), b32.double())
    dy = torch.randn(B, T, N, generator=g, dtype=torch.float64).float()
    # weight / bias gradient of the oracle for the same fp32 operands
    wr = w32.double().requires_grad_(True); br = b32.double().requires_grad_(True)
    O.conv1d_ntc(x32.double(), wr, br).backward(dy.double())
    xd = x32.cuda()

    def run():
        wd, bd = w32.cuda().requires_grad_(True), b32.cuda().requires_grad_(True)
        with ops._hip.KernelTimer() as kt:
            y = ops.conv1d(xd, wd, bd)
            y.backward(dy.cuda())
        return y.detach().cpu().double(), wd.grad.cpu().double(), bd.grad.cpu().double(), [r[0] for r in kt.records]

    ops.conv1d_split(False)
    y_f32, dw_f32, db_f32, _ = run()
    ops.conv1d_split(True)
    try:
        y_split, dw_split, db_split, names = run()
    finally:
        ops.conv1d_split(None)          # back to the default (on)
    assert 'ptts_conv1d_bf16x6' in names, 'the split forward kernel did not run'
    assert ('ptts_conv1d_wgrad_bf16x6' in names) == (KW in (3, 5, 21)), names
    scale = yr.abs().mean()
    e_f32 = ((y_f32 - yr).abs().max() / scale).item()
    e_split = ((y_split - yr).abs().max() / scale).item()
    assert e_split < 3e-5, (e_split, e_f32)                 # tolerance: fp32 rounding of a K <= 12 621 accumulation
    assert e_split < 4 * max(e_f32, 2e-6), (e_split, e_f32)
    if Cin < 100:       # the oracle's own restatement of the six-product arithmetic (numpy loops: small cases only)
        six = torch.from_numpy(O.np_conv1d_same_bf16x6(x32.numpy(), w32.numpy(), b32.numpy()))
        assert ((y_split - six).abs().max() / scale).item() < 1e-5      # fp32 accumulation order only
    # weight gradient: a reduction over B*T frames
    gs = wr.grad.abs().mean()
    g_f32 = ((dw_f32 - wr.grad).abs().max() / gs).item()
    g_split = ((dw_split - wr.grad).abs().max() / gs).item()
    assert g_split < 3e-5, (g_split, g_f32)
    assert g_split < 4 * max(g_f32, 2e-6), (g_split, g_f32)
    close(db_split, br.grad, rtol=2e-4, atol=3e-4, what='db')


@pytest.mark.parametrize('case', [(3, 129, 70, 256, 21), (2, 300, 601, 256, 21), (2, 130, 37, 128, 3)])
def test_conv1d_bf16_products(ops, case):
    """ops.bf16_products(True) (BASELINE configs[2]; build extension): the context Conv1D forward (gemm_bf16x1_kernel: three taps
    per k-step) and its weight gradient (wgrad_bf16x6_kernel<KW, 1>) as ONE product of the operands' bf16 roundings with fp32
    accumulation.  Oracle: the fp64 products of the ROUNDED operands (what is left is fp32 summation order: 2e-5 of the mean
    magnitude, the bound of the fp32 kernels); and, as a sanity bound, within 2^-7 of the unrounded product's scale.  KW = 3 has
    no three-taps-per-step kernel (KW >= 6): its forward stays on the six-product kernel, its weight gradient is one product."""
    B, T, Cin, N, KW = case
    g = gen(33)
    x = torch.randn(B, T, Cin, generator=g).float()
    w = (torch.randn(KW, Cin, N, generator=g) * (1.0 / (KW * Cin) ** 0.5)).float()
    b = torch.randn(N, generator=g).float()
    dy = torch.randn(B, T, N, generator=g).float()
    bf = lambda t: t.to(torch.bfloat16).double()
    one_fwd = KW >= 6
    yr = O.conv1d_ntc(bf(x) if one_fwd else x.double(), bf(w) if one_fwd else w.double(), b.double())
    wr = bf(w).requires_grad_(True)
    O.conv1d_ntc(bf(x), wr, None).backward(bf(dy))          # dW = corr(bf16(x), bf16(dy))
    ops.bf16_products(True)
    try:
        wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
        with ops._hip.KernelTimer() as kt:
            y = ops.conv1d(x.cuda(), wd, bd)
            y.backward(dy.cuda())
        torch.cuda.synchronize()
    finally:
        ops.bf16_products(False)
    names = [r[0] for r in kt.records]
    assert 'ptts_conv1d_bf16x6' in names and 'ptts_conv1d_wgrad_bf16x6' in names, names
    scale = float(yr.abs().mean())
    e = float((y.detach().double().cpu() - yr).abs().max()) / scale
    assert e < 3e-5, 'forward vs the product of the rounded operands: {:.3e}'.format(e)
    exact = O.conv1d_ntc(x.double(), w.double(), b.double())
    assert float((y.detach().double().cpu() - exact).abs().max()) < 2.0 ** -5 * float(exact.abs().mean()) * 4
    gs = float(wr.grad.abs().mean())
    ge = float((wd.grad.double().cpu() - wr.grad).abs().max()) / gs
    assert ge < 3e-5, 'weight gradient vs the product of the rounded operands: {:.3e}'.format(ge)
    close(bd.grad, dy.double().sum((0, 1)), rtol=2e-4, atol=3e-4, what='db')


def test_conv1d_frequency_domain_bf16_products(ops):
    """BASELINE configs[2] with the context Conv1D in the frequency domain (round 4): ops.bf16_products(True) and >= 4096 frames select
    ops._C1FFT with its two big products -- the per-frequency product X^_f H^_f of the forward, the per-frequency correlation of the
    weight gradient -- as ONE bf16 product of the operands' roundings (fp32 accumulation); DFT and inverse DFT stay six-product.
    Bound: the bf16 budget of one rounding per operand of a product (2^-8 relative each): relative L2 <= 1e-2 and no entry off by more
    than 2^-5 of the mean magnitude against the exact fp64 convolution -- what the time-domain one-product kernels are held to -- and
    the asserted kernels ran with planes_count = 1."""
    g = gen(35)
    B, T, Cin, N, KW = 12, 400, 601, 256, 21
    x = (torch.rand(B, T, Cin, generator=g) * 2 - 1).float()
    w = (torch.randn(KW, Cin, N, generator=g) * (1.0 / (KW * Cin) ** 0.5)).float()
    b = torch.randn(N, generator=g).float()
    dy = torch.randn(B, T, N, generator=g).float()
    wr = w.double().requires_grad_(True)
    yr = O.conv1d_ntc(x.double(), wr, b.double())
    yr.backward(dy.double())
    ops.bf16_products(True)
    try:
        wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
        with ops._hip.KernelTimer() as kt:
            y = ops.conv1d(x.cuda(), wd, bd)
            y.backward(dy.cuda())
        torch.cuda.synchronize()
        # the same layer through the time-domain one-product kernels: the error budget must be of the same class
        fft_saved = ops._C1FFT.bf16_one_product
        ops._C1FFT.bf16_one_product = False
        ops.clear_caches()
        wt = w.cuda().requires_grad_(True)
        yt = ops.conv1d(x.cuda(), wt, b.cuda())
        yt.backward(dy.cuda())
        torch.cuda.synchronize()
        ops._C1FFT.bf16_one_product = fft_saved
    finally:
        ops.bf16_products(False)
    tags = [(r[0], r[1]) for r in kt.records]
    assert any(n == 'ptts_dense_bf16x6_batched' and t and t[0] == 'freq' for n, t in tags) and any(n == 'ptts_dense_bf16x6_batched' and t and t[0] == 'corr' for n, t in tags), tags
    assert not any(n in ('ptts_conv1d_bf16x6', 'ptts_conv1d_wgrad_bf16x6') for n, _ in tags)
    def rl2(a, ref): return float((a.detach().double().cpu() - ref).norm() / ref.norm())
    e_f, e_t = rl2(y, yr.detach()), rl2(yt, yr.detach())
    g_f, g_t = rl2(wd.grad, wr.grad), rl2(wt.grad, wr.grad)
    assert e_f <= 1e-2 and g_f <= 1e-2, (e_f, g_f)
    assert e_f <= 2.0 * e_t + 1e-4 and g_f <= 2.0 * g_t + 1e-4, 'frequency-domain one-product error {:.2e} / {:.2e} against the time-domain kernels\' {:.2e} / {:.2e}'.format(e_f, g_f, e_t, g_t)
    assert float((y.detach().double().cpu() - yr.detach()).abs().max()) < 2.0 ** -5 * float(yr.abs().mean())
    close(bd.grad, dy.double().sum((0, 1)), rtol=2e-4, atol=3e-4, what='db')


def test_dense_bf16_products(ops):
    """ops.bf16_products(True): Dense forward (pending LeakyReLU applied, then ONE rounding to bf16), backward-data with the
    output mask, and the two-stage weight gradient as single products of bf16 roundings (csrc/dense.hip with one plane) against
    the fp64 products of the rounded operands; the weight lies in a flat parameter buffer, as in the model."""
    from percivaltts_amd import layers
    g = gen(34)
    M, K, N = 4096, 256, 256
    A = torch.randn(M, K, generator=g).float(); dY = torch.randn(M, N, generator=g).float()

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.randn(K, N, generator=g) / 16)
    h = Holder(); layers.FlatParams(h, 'cuda'); W = h.w
    bias = torch.randn(N, generator=g).float()
    bf = lambda t: t.to(torch.bfloat16).double()
    W64 = bf(W.detach().cpu())
    a_act = bf(torch.maximum(A, A * 0.3))          # the kernel's fp32 LeakyReLU, max(x, 0.3f x), then the one rounding to bf16
    ops.dense_split(True); ops.deterministic(False)
    ops.bf16_products(True)
    try:
        Ad, dYd = A.cuda(), dY.cuda()
        C1 = torch.empty(M, N, device='cuda'); dX = torch.empty(M, K, device='cuda'); dW = torch.empty(K, N, device='cuda'); db = torch.empty(N, device='cuda')
        with ops._hip.KernelTimer() as kt:
            ops.gemm_raw(Ad, W, C1, M, N, K, bias=bias.cuda(), mode=ops.IN_LRELU)
            ops.gemm_raw(dYd, W, dX, M, K, N, transB=1, ldb=N, alpha=0.3, out_mask=Ad)
            ops.gemm_raw(Ad, dYd, dW, K, N, M, transA=1, lda=K, rows_per_seg=M, mode=ops.IN_LRELU, alpha=0.3, colsum_b=db)
        torch.cuda.synchronize()
    finally:
        ops.bf16_products(False); ops.dense_split(None)
    names = [r[0] for r in kt.records]
    assert names.count('ptts_dense_bf16x6') == 2 and any(n.startswith('ptts_dense_wgrad_bf16x6') for n in names), ' '.join(names)
    want = a_act @ W64 + bias.double()
    assert float((C1.double().cpu() - want).abs().max()) < 3e-5 * float(want.abs().mean()) * 4, 'dense forward'
    want = (bf(dY) @ W64.t()) * torch.where(A.double() > 0, 1.0, 0.3)
    assert float((dX.double().cpu() - want).abs().max()) < 3e-5 * float(want.abs().mean()) * 4, 'dense backward data'
    want = a_act.t() @ bf(dY)
    assert float((dW.double().cpu() - want).abs().max()) < 3e-5 * float(want.abs().mean()) * 4, 'dense weight gradient'
    close(db, dY.double().sum(0), rtol=2e-4, atol=2e-3, what='dense db')


@pytest.mark.parametrize('shape', [(2, 12, 9, 4), (3, 50, 65, 4), (1, 7, 5, 3)])
def test_gated_product(ops, shape):
    """y = a * sigmoid(b) and its backward (ptts_gated_mul_fwd/bwd: the Multiply + sigmoid of the reference's gated
    convolution, networktts.py:128-134) against torch fp64; the last shape has an element count that is no multiple of 4."""
    g = gen(61)
    a = torch.randn(*shape, generator=g, dtype=torch.float64) * 2
    b = torch.randn(*shape, generator=g, dtype=torch.float64) * 3
    dy = torch.randn(*shape, generator=g, dtype=torch.float64)
    ar, br = ref(a, True), ref(b, True)
    yr = ar * torch.sigmoid(br)
    yr.backward(dy)
    ad, bd = dev(a, True), dev(b, True)
    yd = ops.gated_mul(ad, bd)
    close(yd, yr, what='y')
    yd.backward(dev(dy))
    close(ad.grad, ar.grad, what='da')
    close(bd.grad, br.grad, what='db')


@pytest.mark.parametrize('shape', [(4, 6, 3), (2, 50, 256), (2, 10, 9, 4), (3, 7, 300)])
def test_batchnorm_train_and_infer(ops, shape):
    g = gen(8)
    C = shape[-1]
    z = torch.randn(*shape, generator=g, dtype=torch.float64) * 2 + 0.7
    gamma = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    beta = torch.randn(C, generator=g, dtype=torch.float64)
    dy = torch.randn(*shape, generator=g, dtype=torch.float64)
    fused4d = len(shape) == 4
    zr, gr_, br = ref(z, True), ref(gamma, True), ref(beta, True)
    mm, mv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    yr = O.lrelu(O.BN(gr_, br, mm, mv)(zr, True, update=True, unbiased_moving=fused4d))
    yr.backward(dy)
    zd, gd, bd = dev(z, True), dev(gamma, True), dev(beta, True)
    mmd, mvd = dev(torch.zeros(C)), dev(torch.ones(C))
    # (z is handed through the node: the consumer's gradient and the statistics' share of dz are added in one pass)
    zt, sc, sh = ops.batchnorm_affine(zd, gd, bd, mmd, mvd, True, True, fused4d)
    yd = ops.Lazy(zt, sc, sh, lrelu=True).tensor()
    close(yd, yr, what='y')
    yd.backward(dev(dy))
    close(zd.grad, zr.grad, rtol=3e-4, atol=3e-5, what='dz')
    close(gd.grad, gr_.grad, rtol=3e-4, atol=1e-4, what='dgamma')
    close(bd.grad, br.grad, rtol=3e-4, atol=1e-4, what='dbeta')
    close(mmd, mm, what='moving_mean')
    close(mvd, mv, what='moving_var')
    # inference mode uses the moving statistics
    _, sc2, sh2 = ops.batchnorm_affine(zd.detach(), gd.detach(), bd.detach(), mmd, mvd, False)
    yi = ops.Lazy(zd.detach(), sc2, sh2, lrelu=True).tensor()
    close(yi, O.lrelu(O.BN(ref(gamma), ref(beta), mm, mv)(ref(z), False)), what='infer')


@pytest.mark.parametrize('act', [None, 'lrelu', 'sigmoid', 'tanh'])
def test_affine_act(ops, act):
    g = gen(9)
    x = torch.randn(3, 11, 20, generator=g, dtype=torch.float64)
    sc = torch.rand(20, generator=g, dtype=torch.float64) + 0.5
    sh = torch.randn(20, generator=g, dtype=torch.float64)
    dy = torch.randn(3, 11, 20, generator=g, dtype=torch.float64)
    f = {None: lambda t: t, 'lrelu': O.lrelu, 'sigmoid': torch.sigmoid, 'tanh': torch.tanh}[act]
    xr, scr, shr = ref(x, True), ref(sc, True), ref(sh, True)
    yr = f(xr * scr + shr)
    yr.backward(dy)
    xd, scd, shd = dev(x, True), dev(sc, True), dev(sh, True)
    yd = ops.affine_act(xd, scd, shd, act)
    close(yd, yr, what='y')
    yd.backward(dev(dy))
    close(xd.grad, xr.grad, what='dx')
    close(scd.grad, scr.grad, rtol=2e-4, atol=1e-4, what='dscale')
    close(shd.grad, shr.grad, rtol=2e-4, atol=1e-4, what='dshift')
    # no affine, odd channel count (scalar path)
    x2 = torch.randn(5, 7, generator=g, dtype=torch.float64)
    close(ops.affine_act(dev(x2), None, None, act), f(x2), what='y-noaffine')


@pytest.mark.parametrize('C', [4, 8, 32, 128, 256, 260, 1024, 86, 2044])
def test_colsums_modes_wide(ops, C):
    """Column sums / sums of squares at the layer widths of the networks, vectorised and scalar paths, all transforms."""
    g = gen(15)
    rows = 5003
    x = torch.randn(rows, C, generator=g, dtype=torch.float64)
    m = torch.randn(rows, C, generator=g, dtype=torch.float64)
    sc = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    sh = torch.randn(C, generator=g, dtype=torch.float64) * 0.3
    for mode, want in ((ops.IN_NONE, x), (ops.IN_LRELU, O.lrelu(x * sc + sh)), (ops.IN_MASKMUL, x * torch.where(m > 0, 1.0, 0.3))):
        kw = {}
        if mode == ops.IN_LRELU:
            kw = dict(scale=dev(sc), shift=dev(sh))
        elif mode == ops.IN_MASKMUL:
            kw = dict(mask_src=dev(m))
        got = ops.colsums(dev(x), mode=mode, **kw)
        close(got[:C], want.sum(0), rtol=1e-5, atol=2e-3, what='sum mode %d' % mode)
        close(got[C:], (want * want).sum(0), rtol=1e-5, atol=2e-3, what='sumsq mode %d' % mode)


@pytest.mark.parametrize('act', [None, 'lrelu', 'sigmoid', 'tanh'])
def test_affine_act_bwd_wide(ops, act):
    g = gen(16)
    rows, C = 3001, 256
    x = torch.randn(rows, C, generator=g, dtype=torch.float64)
    sc = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    sh = torch.randn(C, generator=g, dtype=torch.float64)
    dy = torch.randn(rows, C, generator=g, dtype=torch.float64)
    f = {None: lambda t: t, 'lrelu': O.lrelu, 'sigmoid': torch.sigmoid, 'tanh': torch.tanh}[act]
    xr, scr, shr = ref(x, True), ref(sc, True), ref(sh, True)
    yr = f(xr * scr + shr)
    yr.backward(dy)
    xd, scd, shd = dev(x, True), dev(sc, True), dev(sh, True)
    yd = ops.affine_act(xd, scd, shd, act)
    close(yd, yr, what='y')
    yd.backward(dev(dy))
    close(xd.grad, xr.grad, what='dx')
    close(scd.grad, scr.grad, rtol=2e-4, atol=2e-3, what='dscale')
    close(shd.grad, shr.grad, rtol=2e-4, atol=2e-3, what='dshift')


@pytest.mark.parametrize('case', [(3, 7, 5, 4), (16, 20, 24, 64), (5, 33, 10, 70), (20, 9, 12, 32), (64, 6, 40, 256)])
def test_blstm(ops, case):
    B, T, In, H = case
    g = gen(10)
    x = torch.randn(B, T, In, generator=g, dtype=torch.float64)
    W = torch.randn(In, 8 * H, generator=g, dtype=torch.float64) / math.sqrt(In)
    U = torch.randn(2, H, 4 * H, generator=g, dtype=torch.float64) / math.sqrt(H)
    b = torch.randn(8 * H, generator=g, dtype=torch.float64) * 0.2
    dh = torch.randn(B, T, 2 * H, generator=g, dtype=torch.float64)
    xr, Wr, Ur, br = ref(x, True), ref(W, True), ref(U, True), ref(b, True)
    hr = O.blstm(xr, Wr, Ur, br)
    hr.backward(dh)
    xd, Wd, Ud, bd = dev(x, True), dev(W, True), dev(U, True), dev(b, True)
    hd = ops.lstm(xd, Wd, Ud, bd)
    close(hd, hr, rtol=2e-4, atol=2e-5, what='h')
    hd.backward(dev(dh))
    close(xd.grad, xr.grad, rtol=3e-4, atol=1e-4, what='dx')
    close(Wd.grad, Wr.grad, rtol=3e-4, atol=2e-4, what='dW')
    close(Ud.grad, Ur.grad, rtol=3e-4, atol=2e-4, what='dU')
    close(bd.grad, br.grad, rtol=3e-4, atol=2e-4, what='db')


def test_single_direction_lstm_reverse(ops):
    g = gen(11)
    B, T, In, H = 2, 6, 3, 5
    x = torch.randn(B, T, In, generator=g, dtype=torch.float64)
    W = torch.randn(In, 4 * H, generator=g, dtype=torch.float64)
    U = torch.randn(1, H, 4 * H, generator=g, dtype=torch.float64) * 0.5
    b = torch.randn(4 * H, generator=g, dtype=torch.float64) * 0.2
    for rev in (False, True):
        close(ops.lstm(dev(x), dev(W), dev(U), dev(b), reverse=rev), O.lstm_keras(x, W, U[0], b, reverse=rev),
              rtol=2e-4, atol=2e-5, what='h rev=%s' % rev)


def test_gp_and_losses(ops):
    g = gen(12)
    B, T, D = 5, 40, 86
    real = torch.randn(B, T, D, generator=g, dtype=torch.float64)
    fake = torch.randn(B, T, D, generator=g, dtype=torch.float64)
    al = torch.rand(B, generator=g, dtype=torch.float64)
    close(ops.gp_interpolate(dev(real), dev(fake), dev(al)), O.random_weighted_average(real, fake, al), what='x_hat')
    gg = torch.randn(B, T, D, generator=g, dtype=torch.float64) * 0.05
    gr = ref(gg, True)
    n = torch.sqrt((gr * gr).sum(dim=(1, 2)))
    pr = ((1 - n) ** 2).mean()
    (3.0 * pr).backward()
    gd = dev(gg, True)
    pd = ops.grad_penalty(gd)
    close(pd, pr, what='penalty')
    (3.0 * pd).backward()
    close(gd.grad, gr.grad, what='dpenalty/dg')
    v = torch.randn(B, T, 1, generator=g, dtype=torch.float64)
    vr = ref(v, True); (O.wasserstein_loss(-1.0, vr) * 2).backward()
    vd = dev(v, True); wl = ops.wasserstein(vd, -1.0); (wl * 2).backward()
    close(wl, O.wasserstein_loss(-1.0, v), what='wasserstein')
    close(vd.grad, vr.grad, what='dwasserstein')
    w = torch.rand(D, generator=g, dtype=torch.float64)
    yh = ref(fake, True); (O.specweighted_lse_loss(real, yh, w) * 0.7).backward()
    yd = dev(fake, True); l = ops.wlse(yd, dev(real), dev(w)); (l * 0.7).backward()
    close(l, O.specweighted_lse_loss(real, fake, w), what='wlse')
    close(yd.grad, yh.grad, what='dwlse')
    close(ops.wlse(dev(fake), dev(real)), ((real - fake) ** 2).mean(), what='lse')


def test_adam_keras_and_clip(ops):
    g = gen(13)
    n = 10007
    p = torch.randn(n, generator=g, dtype=torch.float64)
    m, v = torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    pd, md, vd = dev(p), dev(m), dev(v)
    step = torch.zeros((), dtype=torch.int32, device='cuda')
    for t in (1, 2, 3):
        gr = torch.randn(n, generator=g, dtype=torch.float64)
        O.adam_keras([p], [gr], [m], [v], t, 1e-3, 0.5, 0.9, 1e-7)
        ops.adam_keras_step_(pd, dev(gr), md, vd, step, 1e-3, 0.5, 0.9, 1e-7)
    assert int(step.item()) == 3
    close(pd, p, rtol=1e-5, atol=1e-6, what='adam p')
    close(md, m, rtol=1e-5, atol=1e-7, what='adam m')
    close(vd, v, rtol=1e-5, atol=1e-7, what='adam v')
    # gscale = 1/world (data parallel): same as averaging the gradient first
    p2, m2, v2 = dev(p), dev(m), dev(v)
    p3, m3, v3 = dev(p), dev(m), dev(v)
    s2 = torch.zeros((), dtype=torch.int32, device='cuda'); s3 = torch.zeros((), dtype=torch.int32, device='cuda')
    gr = torch.randn(n, generator=g, dtype=torch.float64)
    ops.adam_keras_step_(p2, dev(gr * 4), m2, v2, s2, 1e-3, 0.5, 0.9, 1e-7, gscale=0.25)
    ops.adam_keras_step_(p3, dev(gr), m3, v3, s3, 1e-3, 0.5, 0.9, 1e-7)
    close(p2, p3.cpu(), rtol=1e-6, atol=1e-7, what='gscale')
    ops.weight_clip_(pd, -0.01, 0.01)
    close(pd, p.clamp(-0.01, 0.01), rtol=1e-6, atol=1e-7, what='clip')
    assert float(pd.abs().max()) <= 0.01 + 1e-9


def test_cpu_tensor_is_refused(ops):
    from percivaltts_amd._hip import HipLibraryError
    with pytest.raises(HipLibraryError):
        ops.gp_interpolate(torch.zeros(2, 3, 4), torch.zeros(2, 3, 4), torch.zeros(2))


@pytest.mark.parametrize('case', [(2048, 256, 256, 0), (1500, 260, 256, 1), (1024, 64, 100, 0), (1100, 2048, 256, 0), (1024, 20, 256, 0),
                                  (1200, 256, 2048, 1), (1111, 65, 256, 0), (1030, 70, 64, 1)])
def test_dense_bf16x6_planes_and_products(ops, case):
    """csrc/dense.hip: (i) the fragment-ordered weight planes BIT FOR BIT against oracle.np_split3_bf16; (ii) forward
    (LeakyReLU / BatchNorm-affine on load, bias), backward-data (W^T, output mask) and the masked forward of the second-order
    sweep against the fp64 oracle at the tolerance of the fp32-MFMA kernel, and A/B against that kernel.  Cases: the critic's
    shape, a 260-wide output (256 + thin remainder), ragged K, the LSTM projection width, a narrow head, a deep K, and widths that
    are no multiples of 4 (the 65-bin spectral head: element-wise stores)."""
    import ctypes
    M, N, K, transposed = case
    g = gen(77)
    A = torch.randn(M, K, generator=g, dtype=torch.float64).float()
    W = (torch.randn(K, N, generator=g, dtype=torch.float64) / K ** 0.5).float()          # Keras layout [in, out]
    b = torch.randn(N, generator=g, dtype=torch.float64).float()
    lib = ops._hip.lib()
    # ---- (i) planes of B = W (transposed == 0: stored [K][N]) or of B[k][n] = Wt[n][k] with Wt = W^T stored [N][K]
    Wd = W.cuda()
    src = Wd.t().contiguous() if transposed else Wd
    nb = lib.ptts_dense_planes_bytes(N, K)
    planes = torch.zeros(nb, dtype=torch.uint8, device='cuda')
    ops.call('ptts_split3_dense_weight', ops.ptr(src), src.shape[1], K, N, transposed, ops.ptr(planes), ops.stream())
    NT, KS = -(-N // 256) * 16, -(-K // 32)
    np = O.np
    got = planes.view(torch.bfloat16).float().view(3, NT, KS, 64, 8).cpu().numpy()
    want = np.zeros((3, NT * 16, KS * 32), dtype=np.float32)
    for p, pl in enumerate(O.np_split3_bf16(W.numpy())):                                  # planes of W[k][n]
        want[p, :N, :K] = pl.T
    # fragment order: [nt][ks][lane][e] <- n = 16 nt + (lane & 15), k = 32 ks + 8 (lane >> 4) + e
    w4 = want.reshape(3, NT, 16, KS, 4, 8)                                               # [p][nt][li][ks][lg][e]
    ref_frag = w4.transpose(0, 1, 3, 4, 2, 5).reshape(3, NT, KS, 64, 8)                   # lane = lg * 16 + li
    assert (got == ref_frag).all(), 'planes differ from np_split3_bf16 in {} entries'.format(int((got != ref_frag).sum()))
    assert (got.sum(0) == ref_frag.sum(0)).all()
    # ---- (ii) the products through ops.gemm_raw (the routing the layers use) against fp64 and against the fp32 kernel
    from percivaltts_amd import layers
    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(src.clone())
    h = Holder(); flat = layers.FlatParams(h, 'cuda')
    w = h.w
    Ad, bd = A.cuda(), b.cuda()
    scale = (torch.rand(K, generator=g, dtype=torch.float64) + 0.5).float(); shift = torch.randn(K, generator=g, dtype=torch.float64).float()
    msk = torch.randn(M, K, generator=g, dtype=torch.float64).float()
    omask = torch.randn(M, N, generator=g, dtype=torch.float64).float()
    ldb = src.shape[1]
    def run(split, **kw):
        ops.dense_split(split)
        try:
            C = torch.empty(M, N, dtype=torch.float32, device='cuda')
            with ops._hip.KernelTimer() as kt:
                ops.gemm_raw(Ad, w, C, M, N, K, transB=transposed, ldb=ldb, **kw)
            return C.cpu().double(), [r[0] for r in kt.records]
        finally:
            ops.dense_split(None)
    A64, W64 = A.double(), W.double()
    variants = [
        ('plain+bias', dict(bias=bd), A64 @ W64 + b.double()),
        ('lrelu', dict(mode=ops.IN_LRELU, alpha=0.3), O.lrelu(A64) @ W64),
        ('affine+lrelu', dict(mode=ops.IN_LRELU, scale=scale.cuda(), shift=shift.cuda(), alpha=0.3, bias=bd),
         O.lrelu(A64 * scale.double() + shift.double()) @ W64 + b.double()),
        ('maskmul', dict(mode=ops.IN_MASKMUL, mask_src=msk.cuda(), alpha=0.3), (A64 * torch.where(msk.double() > 0, 1.0, 0.3)) @ W64),
        ('out_mask', dict(out_mask=omask.cuda(), alpha=0.3), (A64 @ W64) * torch.where(omask.double() > 0, 1.0, 0.3)),
    ]
    for name, kw, ref64 in variants:
        y_split, names = run(True, **kw)
        y_f32, names32 = run(False, **kw)
        assert 'ptts_dense_bf16x6' in names and 'ptts_dense_bf16x6' not in names32, (name, names, names32)
        sc = ref64.abs().mean()
        e_split = ((y_split - ref64).abs().max() / sc).item(); e_f32 = ((y_f32 - ref64).abs().max() / sc).item()
        # fp32 accumulation of K terms: a relative error of about sqrt(K) 2^-24 per output, its maximum over 10^5..10^6 outputs
        # at about 5 sigma -> 8 sqrt(K) 2^-24 of mean |y| (the fp32 kernel's k-interleaved sums do better than that at deep K)
        assert e_split < 3e-5 and e_split < max(4 * max(e_f32, 2e-6), 8 * 2.0 ** -24 * K ** 0.5), (name, e_split, e_f32)
    # accumulate
    ops.dense_split(True)
    try:
        C = torch.ones(M, N, dtype=torch.float32, device='cuda')
        ops.gemm_raw(Ad, w, C, M, N, K, transB=transposed, ldb=ldb, accumulate=1)
    finally:
        ops.dense_split(None)
    close(C, A64 @ W64 + 1.0, rtol=2e-4, atol=2e-4, what='accumulate')


@pytest.mark.parametrize('case', [(256, 256, 4096), (260, 256, 3001), (256, 2048, 2100), (64, 100, 2500), (516, 32, 2048)])
def test_dense_weight_gradient_bf16x6(ops, case):
    """csrc/dense.hip, dW[Kin,N] = T(A)^T . dY and db = column sums of dY as a bf16x6 split product (both operands split in
    the kernel, read transposed out of the LDS; fp32 atomics over the workgroups that share a tile) against the fp64 oracle
    at the tolerance of the fp32-MFMA kernel and A/B against it: plain, LeakyReLU, BatchNorm-affine + LeakyReLU and
    gradient-penalty mask on A; a 260-wide input (three column tiles, the last with 4 columns), a ragged frame count, the LSTM
    projection width, widths that are no multiples of 16."""
    Kin, N, M = case
    g = gen(78)
    A = torch.randn(M, Kin, generator=g, dtype=torch.float64).float()
    dY = torch.randn(M, N, generator=g, dtype=torch.float64).float()
    scale = (torch.rand(Kin, generator=g, dtype=torch.float64) + 0.5).float(); shift = torch.randn(Kin, generator=g, dtype=torch.float64).float()
    msk = torch.randn(M, Kin, generator=g, dtype=torch.float64).float()
    Ad, dYd = A.cuda(), dY.cuda()
    A64, dY64 = A.double(), dY.double()
    def run(split, **kw):
        ops.dense_split(split)
        min_n, ops._DenseSplit.wgrad_min_n = ops._DenseSplit.wgrad_min_n, 16      # (the default routes only wide products here)
        try:
            C = torch.full((Kin, N), 7.0, dtype=torch.float32, device='cuda')
            db = torch.full((N,), 7.0, dtype=torch.float32, device='cuda')
            with ops._hip.KernelTimer() as kt:
                ops.gemm_raw(Ad, dYd, C, Kin, N, M, transA=1, lda=Kin, rows_per_seg=M, colsum_b=db, **kw)
            return C.cpu().double(), db.cpu().double(), [r[0] for r in kt.records]
        finally:
            ops.dense_split(None)
            ops._DenseSplit.wgrad_min_n = min_n
    variants = [
        ('plain', dict(), A64),
        ('lrelu', dict(mode=ops.IN_LRELU, alpha=0.3), O.lrelu(A64)),
        ('affine+lrelu', dict(mode=ops.IN_LRELU, scale=scale.cuda(), shift=shift.cuda(), alpha=0.3), O.lrelu(A64 * scale.double() + shift.double())),
        ('maskmul', dict(mode=ops.IN_MASKMUL, mask_src=msk.cuda(), alpha=0.3), A64 * torch.where(msk.double() > 0, 1.0, 0.3)),
    ]
    for name, kw, TA in variants:
        ref = TA.t() @ dY64
        dw_s, db_s, names = run(True, **kw)
        dw_f, db_f, names32 = run(False, **kw)
        assert 'ptts_dense_wgrad_bf16x6' in names and 'ptts_dense_wgrad_bf16x6' not in names32, (name, names, names32)
        sc = ref.abs().mean()
        e_s = ((dw_s - ref).abs().max() / sc).item(); e_f = ((dw_f - ref).abs().max() / sc).item()
        # fp32 accumulation of M terms (see test_dense_bf16x6_planes_and_products)
        assert e_s < 3e-5 and e_s < max(4 * max(e_f, 2e-6), 8 * 2.0 ** -24 * M ** 0.5), (name, e_s, e_f)
        close(db_s, dY64.sum(0), rtol=2e-4, atol=2e-3, what='db ' + name)


def test_persistent_lstm_forward_equals_the_per_step_launches(ops, monkeypatch):
    """csrc/lstm.hip, lstm_fwd_persistent_kernel (one launch for all T steps, h handed from workgroup to workgroup as
    data-tagged sc1 granules; off by default because it measured slower than the per-step launches): same arithmetic in
    the same order, so h, c and the gates must be BIT-identical to the per-step kernels -- both directions, a batch that is no
    multiple of the 16-sample slice, and against the fp64 oracle."""
    g = gen(13)
    for (B, T, In, H) in ((64, 9, 12, 256), (20, 33, 8, 256)):
        x = torch.randn(B, T, In, generator=g, dtype=torch.float64)
        W = torch.randn(In, 8 * H, generator=g, dtype=torch.float64) / math.sqrt(In)
        U = torch.randn(2, H, 4 * H, generator=g, dtype=torch.float64) / math.sqrt(H)
        b = torch.randn(8 * H, generator=g, dtype=torch.float64) * 0.2
        monkeypatch.delenv('PTTS_LSTM_PERSISTENT', raising=False)
        with ops._hip.KernelTimer() as kt:
            h_steps = ops.lstm(dev(x), dev(W), dev(U), dev(b))
        monkeypatch.setenv('PTTS_LSTM_PERSISTENT', '1')
        h_pers = ops.lstm(dev(x), dev(W), dev(U), dev(b))
        monkeypatch.delenv('PTTS_LSTM_PERSISTENT', raising=False)
        torch.cuda.synchronize()
        assert torch.equal(h_pers, h_steps), float((h_pers - h_steps).abs().max())
        close(h_pers, O.blstm(x, W, U, b), rtol=2e-4, atol=2e-5, what='h')


def test_lstm_recurrence_graph_replay_equals_the_plain_launches(ops):
    """csrc/lstm.hip, lstm_graph_run (ptts_set_lstm_graph(1)): the T step launches of the forward and of the backward recurrence
    captured once per (pointers, shape) tuple and replayed as one hipGraph launch.  Same kernels in the same order: h and every
    gradient BIT-identical to the plain launches; a second call on the same buffers is a replay (counters), a call on other
    buffers captures anew; ptts_lstm_graph_clear drops the cache."""
    import ctypes
    lib = ops._hip.lib()
    g = gen(31)
    B, T, In, H = 20, 24, 8, 256
    x = torch.randn(B, T, In, generator=g, dtype=torch.float64)
    W = torch.randn(In, 8 * H, generator=g, dtype=torch.float64) / math.sqrt(In)
    U = torch.randn(2, H, 4 * H, generator=g, dtype=torch.float64) / math.sqrt(H)
    b = torch.randn(8 * H, generator=g, dtype=torch.float64) * 0.2
    dy = dev(torch.randn(B, T, 2 * H, generator=g, dtype=torch.float64))

    def run():
        xs = [dev(t).requires_grad_(True) for t in (x, W, U, b)]
        h = ops.lstm(*xs)
        h.backward(dy)
        torch.cuda.synchronize()
        return [h.detach().clone()] + [t.grad.detach().clone() for t in xs]

    def stats():
        v = [ctypes.c_ulonglong(0) for _ in range(3)]
        lib.ptts_lstm_graph_stats(*[ctypes.byref(q) for q in v])
        return [q.value for q in v]

    lib.ptts_set_lstm_graph(0)
    ops.deterministic(True)          # the products around the recurrence (dx, dW, dU) without order-dependent atomics
    plain = run()
    try:
        lib.ptts_lstm_graph_clear()
        lib.ptts_set_lstm_graph(1)
        s0 = stats()
        first = run()
        s1 = stats()
        again = run()           # the allocator hands the same blocks back: replays, or new captures -- identical results either way
        s2 = stats()
    finally:
        lib.ptts_set_lstm_graph(0)
        lib.ptts_lstm_graph_clear()
        ops.deterministic(False)
    assert s1[1] - s0[1] == 2 and s1[2] == s0[2], (s0, s1)              # forward + backward chains captured
    assert (s2[0] - s1[0]) + (s2[1] - s1[1]) == 2, (s1, s2)
    for name, a, c, p in zip(('h', 'dx', 'dW', 'dU', 'db'), first, again, plain):
        assert torch.equal(a, p) and torch.equal(c, p), name


def test_bn_batch_stats_one_launch_equals_the_three_launch_path(ops):
    """csrc/elementwise.hip, bn_stats_fused_kernel (the conv stacks' C = 4 maps): partial sums, the finishing workgroup's fixed-order
    reduction and bn_finalize's arithmetic in ONE launch.  Against ptts_colstats + ptts_bn_finalize on the same map: affine, saved mean /
    rstd and the moving statistics to fp32 rounding (the two reduction trees differ in the last bits of the fp64 sums); against the
    fp64 oracle; twice the same bits (the result does not depend on which workgroup finishes last); the counter is left zero."""
    import ctypes
    lib = ops._hip.lib()
    g = gen(5)
    for (B, T, F, C) in ((64, 400, 65, 4), (3, 37, 13, 4), (2, 50, 9, 16), (1, 1, 5, 8)):
        z = dev(torch.randn(B, T, F, C, generator=g, dtype=torch.float64) * 1.7 + 0.4)
        gamma = dev(torch.rand(C, generator=g, dtype=torch.float64) + 0.5)
        beta = dev(torch.randn(C, generator=g, dtype=torch.float64))
        rows = B * T * F
        assert lib.ptts_bn_batch_stats_supported(rows, C) == 1
        outs = []
        for fused in (True, False, True):
            mm, mv = dev(torch.full((C,), 0.25, dtype=torch.float64)), dev(torch.full((C,), 1.5, dtype=torch.float64))
            scale, shift, mean, rstd = [torch.empty(C, dtype=torch.float32, device='cuda') for _ in range(4)]
            ws = torch.empty(lib.ptts_colstats_workspace_bytes(rows, C), dtype=torch.uint8, device='cuda')
            if fused:
                cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
                ops.call('ptts_bn_batch_stats', ops.ptr(z), rows, C, ops.ptr(gamma), ops.ptr(beta), ops.ptr(mm), ops.ptr(mv), 1e-3, 0.99, 1, 0,
                         ops.ptr(scale), ops.ptr(shift), ops.ptr(mean), ops.ptr(rstd), ops.ptr(ws), ws.numel(), ops.ptr(cnt), ops.stream())
                torch.cuda.synchronize()
                assert int(cnt.item()) == 0
            else:
                sums = torch.empty(2 * C, dtype=torch.float64, device='cuda')
                ops.call('ptts_colstats', ops.ptr(z), rows, C, 0, None, None, None, 0.3, ops.ptr(sums), ops.ptr(ws), ws.numel(), ops.stream())
                ops.call('ptts_bn_finalize', ops.ptr(sums), rows, ops.ptr(gamma), ops.ptr(beta), ops.ptr(mm), ops.ptr(mv), 1e-3, 0.99, 1, 1, 0, C,
                         ops.ptr(scale), ops.ptr(shift), ops.ptr(mean), ops.ptr(rstd), ops.stream())
            torch.cuda.synchronize()
            outs.append([t.clone() for t in (scale, shift, mean, rstd, mm, mv)])
        for a, b in zip(outs[0], outs[2]):
            assert torch.equal(a, b), 'the fused launch is not reproducible'
        for name, a, b in zip(('scale', 'shift', 'mean', 'rstd', 'moving_mean', 'moving_var'), outs[0], outs[1]):
            close(a, b.double().cpu(), rtol=2e-6, atol=1e-6, what=name + ' fused vs three launches')
        z64 = z.double().cpu().reshape(-1, C)
        m64, v64 = z64.mean(0), z64.var(0, unbiased=False)
        close(outs[0][2], m64, rtol=1e-5, atol=1e-6, what='mean')
        close(outs[0][0], gamma.double().cpu() / torch.sqrt(v64 + 1e-3), rtol=1e-5, atol=1e-6, what='scale')


@pytest.mark.parametrize('M', [2048, 3000, 4095])
def test_dense_weight_gradient_deferred_two_stage_between_2048_and_4096_rows(ops, M):
    """ops.deferred_weight_grads() hands every Dense weight gradient over >= 2048 frames to the two-stage split kernels
    (ptts_dense_wgrad_bf16x6_partials + ptts_dense_wgrad_reduce_grouped, csrc/dense.hip); the threshold came down from 4096
    without a case below it.  Weights of a flat parameter buffer (the deferred path accumulates straight into the
    buffer's gradient views), two products that share one gradient buffer (first- and second-order sweeps of a layer do), a ragged
    row count: dW and db against the fp64 products, and the kernels named."""
    import torch.nn as nn
    from percivaltts_amd import layers
    g = gen(41 + M)
    K, N = 256, 256

    class Net(nn.Module):
        def __init__(self):
            super(Net, self).__init__()
            self.w = nn.Parameter(torch.randn(K, N, generator=g) / 16.0)
            self.b = nn.Parameter(torch.zeros(N))
    net = Net()
    flat = layers.FlatParams(net, torch.device('cuda'))
    x1 = torch.randn(M, K, generator=g, dtype=torch.float64)
    x2 = torch.randn(M, K, generator=g, dtype=torch.float64)
    dy1 = torch.randn(M, N, generator=g, dtype=torch.float64)
    dy2 = torch.randn(M, N, generator=g, dtype=torch.float64)
    ops.dense_split(True); ops.deterministic(False)
    flat.zero_grad()
    with ops._hip.KernelTimer() as kt, ops.deferred_weight_grads():          # (the queue is launched when the inner context exits)
        for x, dy in ((x1, dy1), (x2, dy2)):
            y = ops.dense(dev(x).view(1, M, K), net.w, net.b)
            y.backward(dev(dy).view(1, M, N))
    torch.cuda.synchronize()
    names = [n for n, _, _ in kt.durations_ms()]
    assert names.count('ptts_dense_wgrad_bf16x6_partials') == 2 and 'ptts_dense_wgrad_reduce_grouped' in names, names
    dW = x1.t() @ dy1 + x2.t() @ dy2
    db = dy1.sum(0) + dy2.sum(0)
    e = float((net.w.grad.double().cpu() - dW).norm() / dW.norm())
    assert e < 3e-6, e
    close(net.b.grad, db, rtol=2e-5, atol=2e-4, what='db')


@pytest.mark.parametrize('seg', [100, 0], ids=['segments', 'whole'])
@pytest.mark.parametrize('case', [(12, 400, 601, 256, 21), (16, 256, 70, 32, 5), (11, 400, 601, 256, 21), (9, 460, 128, 64, 9), (5, 1000, 64, 16, 3), (8, 512, 96, 48, 13)])
def test_conv1d_frequency_domain_forward(ops, case, seg):
    """ops._C1FFT (conv1d_fft(True)): the context Conv1D forward as DFT -> per-frequency products -> inverse DFT, each stage a batched
    bf16x6 split product (ptts_dense_bf16x6_batched), overlap-save over segments of S frames with windows of P = S + KW - 1 (no
    power-of-two transform) or over whole utterances.  Against the fp64 oracle at the
    tolerance of the time-domain kernels (fp32 arithmetic; the transforms' twiddles are exact to fp32 rounding), with bias, 'same'
    padding at both utterance borders, a batch that is no multiple of anything; the weight gradient by the correlation theorem from
    the same transforms (ops._C1FFT.wgrad) and the bias gradient, against the fp64 oracle as well."""
    B, T, Cin, N, KW = case
    g = gen(77 + B)
    x = torch.randn(B, T, Cin, generator=g, dtype=torch.float64)
    w = torch.randn(KW, Cin, N, generator=g, dtype=torch.float64) / math.sqrt(KW * Cin)
    b = torch.randn(N, generator=g, dtype=torch.float64)
    yr = O.conv1d_ntc(x, w, b)
    xd, wd, bd = dev(x), dev(w, True), dev(b, True)
    ops.conv1d_fft(True); ops.conv1d_split(True)
    seg0 = ops._C1FFT.seg_target
    ops._C1FFT.seg_target = seg                  # overlap-save over segments of about 100 frames (the default), or whole utterances
    try:
        assert ops._C1FFT.eligible(xd, wd)
        with ops._hip.KernelTimer() as kt:
            yd = ops.conv1d(xd, wd, bd)
        names = [n for n, _, _ in kt.durations_ms()]
        assert names.count('ptts_dense_bf16x6_batched') == 3 and 'ptts_conv1d_bf16x6' not in names, names
        scale = float(yr.abs().mean())
        e = float((yd.double().cpu() - yr).abs().max()) / scale
        assert e < 2e-5, 'frequency-domain conv1d: max error {:.3e} of mean |y|'.format(e)
        # the same input again (the critic after the generator): the transform of x is reused
        with ops._hip.KernelTimer() as kt2:
            yd2 = ops.conv1d(xd, wd, bd)
        names2 = [n for n, _, _ in kt2.durations_ms()]
        assert names2.count('ptts_dense_bf16x6_batched') == 2 and 'ptts_dft_mirror' not in names2, names2
        assert torch.equal(yd, yd2)
        dy = torch.randn(B, T, N, generator=g, dtype=torch.float64)
        with ops._hip.KernelTimer() as kt3:
            yd.backward(dev(dy))
        # the correlation theorem: dW from X^ and DY^ (an odd number of segments in the batch -- 2 B NS no multiple of 4 -- takes the
        # time-domain kernel)
        nseg = B * (T // ops._C1FFT.segment(T, KW))
        last = 'ptts_conv1d_freq_wgrad_inverse' if KW in ops._C1FFT.KWS else 'ptts_conv1d_freq_wgrad_combine'      # (KW = 13: the un-fused fallback)
        assert (last in [n for n, _, _ in kt3.durations_ms()]) == (nseg % 2 == 0)
        close(bd.grad, dy.sum((0, 1)), rtol=2e-5, atol=2e-4, what='db')
        wr = ref(w, True)
        O.conv1d_ntc(x, wr, b).backward(dy)
        e = float((wd.grad.double().cpu() - wr.grad).norm() / wr.grad.norm())
        assert e < 2e-5, e
    finally:
        ops._C1FFT.seg_target = seg0
        ops.conv1d_fft(None); ops.conv1d_split(None)


@pytest.mark.parametrize('case', [(3, 400, 601, 4, 100, -10, 120, 120), (3, 400, 256, 4, 100, 0, 120, 100), (2, 460, 70, 5, 92, -4, 100, 100),
                                  (5, 37, 13, 1, 37, -1, 40, 40), (2, 300, 260, 3, 100, -5, 110, 104), (2, 200, 512, 2, 100, 0, 100, 100)])
def test_frame_window_and_strided_planes_bit_for_bit(ops, case):
    """csrc/dense.hip, split3_dense_weight_strided_kernel (ptts_split3_frame_windows / ptts_split3_dense_weight_strided): the planes of
    the overlap-save windows of a frame sequence -- window z = (b, s) = rows x[b][row_off + s S + k], zero outside the utterance and
    for k >= kvalid -- BIT FOR BIT against oracle.np_split3_bf16 of the windows built on the host, in the fragment order of the
    batched products; and the strided form against the same matrices laid out at a regular stride."""
    B, T, C, NS, S, row_off, P, kvalid = case
    np = O.np
    lib = ops._hip.lib()
    g = gen(3 + C)
    x = torch.randn(B, T, C, generator=g, dtype=torch.float64).float()
    xd = x.cuda()
    Z = B * NS
    nb = lib.ptts_dense_planes_bytes(C, P)
    planes = torch.full((Z * nb,), 0xAB, dtype=torch.uint8, device='cuda')
    ops.call('ptts_split3_frame_windows', ops.ptr(xd), B, T, C, NS, S, row_off, P, kvalid, ops.ptr(planes), nb, ops.stream())
    NT, KS = -(-C // 256) * 16, -(-P // 32)
    got = planes.view(torch.bfloat16).float().view(Z, 3, NT, KS, 64, 8).cpu().numpy()
    win = np.zeros((Z, P, C), dtype=np.float32)
    xn = x.numpy()
    for b in range(B):
        for s in range(NS):
            for k in range(min(P, kvalid)):
                t = row_off + s * S + k
                if 0 <= t < T:
                    win[b * NS + s, k] = xn[b, t]

    def frag(mat):                                   # planes of mat [K][N] in fragment order [3][NT][KS][64][8]
        want = np.zeros((3, NT * 16, KS * 32), dtype=np.float32)
        for p, pl in enumerate(O.np_split3_bf16(mat)):
            want[p, :mat.shape[1], :mat.shape[0]] = pl.T
        return want.reshape(3, NT, 16, KS, 4, 8).transpose(0, 1, 3, 4, 2, 5).reshape(3, NT, KS, 64, 8)

    for z in range(Z):
        ref = frag(win[z])
        assert (got[z] == ref).all(), 'window {}: planes differ in {} entries'.format(z, int((got[z] != ref).sum()))
    # the strided form on the same matrices stored one after the other
    wd = torch.from_numpy(win).cuda()
    planes2 = torch.full((Z * nb,), 0xCD, dtype=torch.uint8, device='cuda')
    ops.call('ptts_split3_dense_weight_strided', ops.ptr(wd), P * C, ops.ptr(planes2), nb, Z, C, P, C, 0, ops.stream())
    torch.cuda.synchronize()
    assert torch.equal(planes, planes2)


def test_frequency_domain_building_blocks(ops):
    """The C-ABI pieces of the frequency-domain Conv1D on their own (ops._C1FFT composes them): ptts_dense_bf16x6_batched (products of
    one shape at regular strides, shared or per-member left operand, bias) against fp64 products; ptts_transpose_batched and
    ptts_dft_mirror exactly; ptts_conv1d_freq_kernel_planes against the fp64 twiddle product (the three planes summed);
    ptts_conv1d_freq_wgrad_inverse against the fp64 sum over the frequencies."""
    import ctypes
    np = O.np
    lib = ops._hip.lib()
    g = gen(91)
    # ---- batched products: C_z = A_z . B_z + bias, A shared (stride 0) and per member
    nbat, M, N, K = 5, 100, 70, 132
    A = torch.randn(nbat, M, K, generator=g, dtype=torch.float64).float().cuda()
    Bm = (torch.randn(nbat, K, N, generator=g, dtype=torch.float64) / K ** 0.5).float().cuda()
    bias = torch.randn(N, generator=g, dtype=torch.float64).float().cuda()
    npb = lib.ptts_dense_planes_bytes(N, K)
    planes = torch.empty(nbat * npb, dtype=torch.uint8, device='cuda')
    ops.call('ptts_split3_dense_weight_strided', ops.ptr(Bm), K * N, ops.ptr(planes), npb, nbat, N, K, N, 0, ops.stream())
    for shared in (False, True):
        C = torch.full((nbat, M, N), float('nan'), device='cuda')
        ops.call('ptts_dense_bf16x6_batched', ops.ptr(A), 0 if shared else M * K, ops.ptr(planes), npb, ops.ptr(bias), ops.ptr(C), M * N, nbat,
                 M, N, K, K, N, 3, ops.stream())
        for z in range(nbat):
            want = A[0 if shared else z].double().cpu() @ Bm[z].double().cpu() + bias.double().cpu()
            e = float((C[z].double().cpu() - want).abs().max() / want.abs().mean())
            assert e < 2e-5, ('batched product', shared, z, e)
    # ---- batched transpose, exactly
    src = torch.randn(7, 45, 70, generator=g).cuda()
    dst = torch.empty(7, 70, 45, device='cuda')
    ops.call('ptts_transpose_batched', ops.ptr(src), ops.ptr(dst), 7, 45, 70, ops.stream())
    assert torch.equal(dst, src.transpose(1, 2).contiguous())
    # ---- mirror: rows [Xr | .], [Xi | .] -> [Xr | -Xi], [Xi | Xr]
    NB, Z, Cin, Kh = 4, 6, 13, 16
    Ap = torch.zeros(NB, 2, Z, 2 * Kh, device='cuda')
    Ap[:, :, :, :Cin] = torch.randn(NB, 2, Z, Cin, generator=g).cuda()
    ref = Ap.clone()
    ref[:, 0, :, Kh:Kh + Cin] = -Ap[:, 1, :, :Cin]
    ref[:, 1, :, Kh:Kh + Cin] = Ap[:, 0, :, :Cin]
    ops.call('ptts_dft_mirror', ops.ptr(Ap), NB, Z, Cin, Kh, ops.stream())
    assert torch.equal(Ap, ref)
    # ---- the kernel's transform into planes: the planes summed = the twiddle product (fp32 rounding)
    KW, Cin, N, Kh, NB = 5, 13, 20, 16, 6
    w = (torch.randn(KW, Cin, N, generator=g, dtype=torch.float64) * 0.3).float().cuda()
    tw = torch.randn(2 * NB, KW, generator=g, dtype=torch.float64).float().cuda()
    npw = lib.ptts_dense_planes_bytes(N, 2 * Kh)
    pl = torch.empty(NB * npw, dtype=torch.uint8, device='cuda')
    ops.call('ptts_conv1d_freq_kernel_planes', ops.ptr(w), ops.ptr(tw), ops.ptr(pl), NB, KW, Cin, N, Kh, ops.stream())
    NT, KS = -(-N // 256) * 16, -(-(2 * Kh) // 32)
    got = pl.view(torch.bfloat16).float().view(NB, 3, NT, KS, 4, 16, 8).sum(1).cpu().numpy()     # [f][nt][ks][lg][li][e]
    what = np.einsum('rk,kcn->rcn', tw.double().cpu().numpy(), w.double().cpu().numpy()).reshape(NB, 2, Cin, N)
    for f in range(NB):
        for part in range(2):
            for c in (0, 5, Cin - 1):
                for n in (0, 7, N - 1):
                    k = part * Kh + c
                    v = got[f, n // 16, k // 32, (k % 32) // 8, n % 16, k % 8]
                    assert abs(v - what[f, part, c, n]) < 2e-6 * (1 + abs(what[f, part, c, n])), (f, part, c, n, v, what[f, part, c, n])
    assert float(np.abs(got[:, :, :, :, :, :][..., 0]).sum()) > 0
    # ---- the inverse transform of the weight gradient: dW[k][c][n] = sum_f t2[f][k] Gt[f][n][c] + t2[f][KW + k] Gt[f][n][Kh + c]
    KW, Cin, N, Kh, NB, TP = 5, 13, 20, 16, 9, 16
    Gt = torch.randn(NB, N, 2 * Kh, generator=g, dtype=torch.float64).float().cuda()
    t2 = torch.zeros(NB, TP, device='cuda')
    t2[:, :2 * KW] = torch.randn(NB, 2 * KW, generator=g).cuda()
    dW = torch.empty(KW, Cin, N, device='cuda')
    ws = torch.empty(lib.ptts_conv1d_freq_wgrad_inverse_workspace_bytes(KW, Cin, N), dtype=torch.uint8, device='cuda')
    ops.call('ptts_conv1d_freq_wgrad_inverse', ops.ptr(Gt), ops.ptr(t2), ops.ptr(dW), ops.ptr(ws), ws.numel(), NB, TP, KW, Cin, N, Kh, ops.stream())
    G64, t64 = Gt.double().cpu().numpy(), t2.double().cpu().numpy()
    want = np.einsum('fk,fnc->kcn', t64[:, :KW], G64[:, :, :Cin]) + np.einsum('fk,fnc->kcn', t64[:, KW:2 * KW], G64[:, :, Kh:Kh + Cin])
    assert float(np.abs(dW.double().cpu().numpy() - want).max()) < 1e-5 * (1 + float(np.abs(want).max()))


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(3, 100, 65), (1, 1, 65), (2, 37, 130), (300, 17, 3), (17, 270, 65), (64, 400, 65)])
@pytest.mark.parametrize('mode', ['none', 'lrelu', 'affine'])
def test_conv2d_forward_that_sums_its_outputs_for_the_batchnorm_behind_it(ops, case, mode):
    """ptts_conv2d_mfma_fwd_stats (the convolution of pCNN2D, reference networktts.py:122-126, in front of its BatchNormalization): the map
    is the plain launch's bit for bit, the per-workgroup rows add up to the per-channel sums of the map and of its squares, and
    ptts_bn_finalize_partials makes of them what ptts_bn_batch_stats makes of a pass over the map -- affine, batch moments, moving
    averages (fp64 oracle: mean / biased variance of the map)."""
    import ctypes
    B, T, F = case
    g = gen(91)
    x = torch.randn(B, T, F, 4, generator=g).cuda()
    w = (torch.randn(5, 5, 4, 4, generator=g) * 0.3).cuda()
    b = torch.randn(4, generator=g).cuda()
    sc = (torch.rand(4, generator=g) + 0.5).cuda() if mode == 'affine' else None
    sh = (torch.randn(4, generator=g) * 0.3).cuda() if mode == 'affine' else None
    in_mode = ops.IN_NONE if mode == 'none' else ops.IN_LRELU
    lib = ops._hip.lib()
    assert lib.ptts_conv2d_mfma_fwd_stats_supported(F, 1, in_mode) == 1
    y0 = ops._conv2d_fwd_raw(x, w, b, sc, sh, None, in_mode, 0.3, 1, ops.PAD_SAME)
    ops._BNStats.want, ops._BNStats.last = True, None
    try:
        y1 = ops._conv2d_fwd_raw(x, w, b, sc, sh, None, in_mode, 0.3, 1, ops.PAD_SAME)
    finally:
        ops._BNStats.want = False
    part, nrows, count = ops._BNStats.last
    ops._BNStats.last = None
    assert torch.equal(y0, y1) and count == B * T * F and 1 <= nrows <= 256
    sums = part[:nrows].sum(0).cpu()
    yd = y0.double().reshape(-1, 4).cpu()
    close(sums[:4], yd.sum(0), 1e-6, 1e-6 * float(yd.abs().sum(0).max()), 'channel sums')
    close(sums[4:], (yd * yd).sum(0), 1e-6, 0.0, 'channel sums of squares')
    # the finish against the one-pass statistics kernel and against the oracle's moments
    gamma = (torch.rand(4, generator=g) + 0.5).cuda(); beta = torch.randn(4, generator=g).cuda()
    outs = []
    for which in (0, 1):
        mm = torch.full((4,), 0.25, device='cuda'); mv = torch.full((4,), 2.0, device='cuda')
        scale, shift, mean, rstd = (torch.empty(4, device='cuda') for _ in range(4))
        rows = B * T * F
        if which == 0:
            ws = ops._workspace(lib.ptts_colstats_workspace_bytes(rows, 4), x.device)
            ops.call('ptts_bn_batch_stats', ops.ptr(y0), rows, 4, ops.ptr(gamma), ops.ptr(beta), ops.ptr(mm), ops.ptr(mv), 1e-3, 0.99, 1, 1,
                     ops.ptr(scale), ops.ptr(shift), ops.ptr(mean), ops.ptr(rstd), ops.ptr(ws), ws.numel(), ops.ptr(ops._stream_counter(x.device)), ops.stream())
        else:
            ops.call('ptts_bn_finalize_partials', ops.ptr(part), nrows, rows, 4, ops.ptr(gamma), ops.ptr(beta), ops.ptr(mm), ops.ptr(mv), 1e-3, 0.99, 1, 1,
                     ops.ptr(scale), ops.ptr(shift), ops.ptr(mean), ops.ptr(rstd), ops.stream())
        torch.cuda.synchronize()
        outs.append((scale, shift, mean, rstd, mm, mv))
    for nm, a, c in zip(('scale', 'shift', 'mean', 'rstd', 'moving mean', 'moving variance'), outs[1], outs[0]):
        close(a, c.cpu(), 2e-6, 2e-6, nm + ' (partial rows against the one-pass kernel)')
    mu = yd.mean(0); var = yd.var(0, unbiased=False)
    close(outs[1][2], mu, 1e-5, 1e-6, 'batch mean against the oracle')
    close(outs[1][3], 1.0 / torch.sqrt(var + 1e-3), 1e-5, 0.0, 'batch rstd against the oracle')


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(2048, 256, 256), (1500, 256, 260), (25600, 256, 256), (3001, 512, 64), (1100, 128, 2048)])
@pytest.mark.parametrize('mode', ['none', 'lrelu', 'affine'])
def test_dense_product_that_sums_its_columns_for_the_batchnorm_behind_it(ops, case, mode):
    """ptts_dense_bf16x6_stats (the Dense of pFC, reference networktts.py:59-63, in front of its BatchNormalization): the product is the plain
    launch's bit for bit, the per-row-tile rows add up to the column sums of the product and of its squares, and ptts_bn_finalize_partials
    gives the batch moments of the fp64 oracle."""
    import ctypes
    M, N, K = case
    g = gen(92)
    from percivaltts_amd import layers
    x = torch.randn(M, K, generator=g).cuda()
    w0 = torch.randn(K, N, generator=g) * 0.1

    class Holder(torch.nn.Module):               # (the split kernel keeps planes of weights that live in a flat parameter buffer)
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(w0.clone())
    h = Holder(); flat = layers.FlatParams(h, 'cuda')
    w = h.w
    b = torch.randn(N, generator=g).cuda()
    sc = (torch.rand(K, generator=g) + 0.5).cuda() if mode == 'affine' else None
    sh = (torch.randn(K, generator=g) * 0.3).cuda() if mode == 'affine' else None
    in_mode = ops.IN_NONE if mode == 'none' else ops.IN_LRELU
    y0 = torch.empty(M, N, device='cuda'); y1 = torch.empty(M, N, device='cuda')
    ops.gemm_raw(x, w, y0, M, N, K, bias=b, mode=in_mode, scale=sc, shift=sh, alpha=0.3)
    ops._BNStats.want, ops._BNStats.last = True, None
    try:
        ops.gemm_raw(x, w, y1, M, N, K, bias=b, mode=in_mode, scale=sc, shift=sh, alpha=0.3)
    finally:
        ops._BNStats.want = False
    assert ops._BNStats.last is not None, 'the split Dense kernel did not take this shape'
    part, nrows, count = ops._BNStats.last
    ops._BNStats.last = None
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and count == M and 1 <= nrows <= part.shape[0] and part.shape[1] == 2 * N
    yd = y0.double().cpu()
    sums = part[:nrows].sum(0).cpu()
    close(sums[:N], yd.sum(0), 1e-5, 1e-5 * float(yd.abs().sum(0).max()), 'column sums')
    close(sums[N:], (yd * yd).sum(0), 1e-5, 0.0, 'column sums of squares')
    gamma = (torch.rand(N, generator=g) + 0.5).cuda(); beta = torch.randn(N, generator=g).cuda()
    mm = torch.full((N,), 0.25, device='cuda'); mv = torch.full((N,), 2.0, device='cuda')
    scale, shift, mean, rstd = (torch.empty(N, device='cuda') for _ in range(4))
    ops.call('ptts_bn_finalize_partials', ops.ptr(part), nrows, M, N, ops.ptr(gamma), ops.ptr(beta), ops.ptr(mm), ops.ptr(mv), 1e-3, 0.99, 1, 0,
             ops.ptr(scale), ops.ptr(shift), ops.ptr(mean), ops.ptr(rstd), ops.stream())
    torch.cuda.synchronize()
    mu = yd.mean(0); var = yd.var(0, unbiased=False); rs = 1.0 / torch.sqrt(var + 1e-3)
    close(mean, mu, 1e-5, 1e-5, 'batch mean'); close(rstd, rs, 1e-5, 0.0, 'batch rstd')
    close(scale, gamma.double().cpu() * rs, 1e-5, 0.0, 'scale'); close(shift, beta.double().cpu() - mu * gamma.double().cpu() * rs, 1e-5, 1e-5, 'shift')
    close(mm, 0.25 * 0.99 + mu * 0.01, 1e-5, 1e-6, 'moving mean'); close(mv, 2.0 * 0.99 + var * 0.01, 1e-5, 1e-6, 'moving variance')


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(2048, 256, 256), (25600, 256, 256), (1500, 64, 512), (3001, 260, 128)])
def test_dense_backward_data_through_a_batchnorm_affine_in_its_store(ops, case):
    """ptts_dense_bf16x6_bwd_affine (the backward-data product of a Dense whose input was BatchNormalization + LeakyReLU, reference
    networktts.py:59-63) against the two-launch path it replaces (product, then ptts_affine_act_bwd) and against fp64: dz, and the
    gradients of the affine's scale and shift."""
    from percivaltts_amd import layers
    M, K, N = case                       # the layer: [M, K] . [K, N]; backward data: dy [M, N] -> dz [M, K]
    g = gen(93)
    z = torch.randn(M, K, generator=g).cuda()
    dy = torch.randn(M, N, generator=g).cuda()
    w0 = torch.randn(K, N, generator=g) * 0.1
    sc = (torch.rand(K, generator=g) + 0.5).cuda(); sh = (torch.randn(K, generator=g) * 0.3).cuda()

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(w0.clone())
    h = Holder(); flat = layers.FlatParams(h, 'cuda')
    w = h.w
    outs = []
    for on in (True, False):
        ops.conv_bn_stats(on)
        try:
            with ops._hip.KernelTimer() as kt:
                dz, dsc, dsh = ops._dense_bwd_data(dy, z, w, ops.IN_LRELU, sc, sh, 0.3, True)
            torch.cuda.synchronize()
        finally:
            ops.conv_bn_stats(None)
        outs.append((dz, dsc, dsh, [r[0] for r in kt.records]))
    assert 'ptts_dense_bf16x6_bwd_affine' in outs[0][3] and 'ptts_affine_act_bwd' not in outs[0][3]
    assert 'ptts_affine_act_bwd' in outs[1][3]
    da = dy.double().cpu() @ w.detach().double().cpu().t()
    zd = z.double().cpu()
    gd = da * torch.where(zd * sc.double().cpu() + sh.double().cpu() > 0, 1.0, 0.3)
    for (dz, dsc, dsh, _), what in zip(outs, ('in the store', 'two launches')):
        close(dz, gd * sc.double().cpu(), 2e-4, 2e-4 * float(gd.abs().mean()), 'dz ' + what)
        close(dsc, (gd * zd).sum(0), 1e-4, 1e-4 * float((gd * zd).abs().sum(0).mean()), 'dscale ' + what)
        close(dsh, gd.sum(0), 1e-4, 1e-4 * float(gd.abs().sum(0).mean()), 'dshift ' + what)


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(3, 100, 65), (2, 37, 130), (300, 17, 3), (17, 270, 65), (64, 400, 65)])
def test_conv2d_backward_through_a_batchnorm_affine_input_on_the_matrix_cores(ops, case):
    """ptts_conv2d_mfma_bwd_fused_affine (the backward of a generator Conv2D whose input was BatchNormalization + LeakyReLU, reference
    networktts.py:122-126) against the packed-FMA kernel it replaces (which the suite pins against the fp64 oracle) and against fp64 on a
    crop: dx w.r.t. the raw map, dW, and the gradients of the affine's scale and shift."""
    B, T, F = case
    g = gen(94)
    x = torch.randn(B, T, F, 4, generator=g).cuda()
    dy = torch.randn(B, T, F, 4, generator=g).cuda()
    w = (torch.randn(5, 5, 4, 4, generator=g) * 0.3).cuda()
    sc = (torch.rand(4, generator=g) + 0.5).cuda(); sh = (torch.randn(4, generator=g) * 0.3).cuda()
    res = []
    for on in (True, False):
        ops.conv_bn_stats(on)
        try:
            with ops._hip.KernelTimer() as kt:
                out = ops._conv2d_bwd_raw(dy, x, w, sc, sh, None, ops.IN_LRELU, 0.3, 1, ops.PAD_SAME, True, True, False, True)
            torch.cuda.synchronize()
        finally:
            ops.conv_bn_stats(None)
        res.append((out, [r[0] for r in kt.records]))
    (dx1, dw1, _, ds1, dh1), names1 = res[0]
    (dx0, dw0, _, ds0, dh0), names0 = res[1]
    assert 'ptts_conv2d_mfma_bwd_fused_affine' in names1 and 'ptts_conv2d_bwd' not in names1
    assert 'ptts_conv2d_bwd' in names0
    n = float(B * T * F)
    close(dx1, dx0.cpu(), 2e-4, 2e-4 * float(dx0.abs().mean()), 'dx')
    close(dw1, dw0.cpu(), 2e-4, 2e-5 * n ** 0.5, 'dW')
    close(ds1, ds0.cpu(), 2e-4, 2e-5 * n ** 0.5, 'dscale')
    close(dh1, dh0.cpu(), 2e-4, 2e-5 * n ** 0.5, 'dshift')
    # fp64 on the first utterance (autograd through the oracle's layer)
    xb = x[:1].double().cpu().requires_grad_(True)
    scd = sc.double().cpu().requires_grad_(True); shd = sh.double().cpu().requires_grad_(True)
    wd = w.double().cpu()
    y = O.conv2d_nhwc(O.lrelu(xb * scd + shd), wd, None)
    y.backward(dy[:1].double().cpu())
    close(dx1[:1], xb.grad, 2e-4, 2e-4 * float(xb.grad.abs().mean()), 'dx against fp64')
    if B == 1:
        close(ds1, scd.grad, 2e-4, 2e-5 * n ** 0.5, 'dscale against fp64')


@pytest.mark.gpu
def test_statistics_entry_points_refuse_what_they_cannot_do(ops):
    """Error behaviour of the round-4 entry points that leave per-workgroup sums: too few rows of room, an input transform the fused
    kernels have no form for, unaligned affine vectors -- a HipLibraryError with the reason, nothing launched, nothing written."""
    import ctypes
    from percivaltts_amd import layers
    lib = ops._hip.lib()
    g = gen(95)
    x = torch.randn(2, 40, 65, 4, generator=g).cuda()
    w = (torch.randn(5, 5, 4, 4, generator=g) * 0.3).cuda()
    tf = ops._C2M.table(w, False)
    y = torch.full_like(x, 7.0)
    part = torch.zeros(4, 8, dtype=torch.float64, device='cuda')
    n = ctypes.c_int(-1)
    with pytest.raises(ops._hip.HipLibraryError, match='room for'):
        ops.call('ptts_conv2d_mfma_fwd_stats', ops.ptr(x), ops.ptr(tf), None, None, None, ops.ptr(y), 2, 40, 65, 5, 2, ops.IN_LRELU, 0.3,
                 ops.ptr(part), 1, ctypes.byref(n), ops.stream())
    assert lib.ptts_conv2d_mfma_fwd_stats_supported(65, 2, ops.IN_LRELU) == 0 and lib.ptts_conv2d_mfma_fwd_stats_supported(65, 1, ops.IN_MASKMUL) == 0
    with pytest.raises(ops._hip.HipLibraryError, match='unsupported'):
        ops.call('ptts_conv2d_mfma_fwd_stats', ops.ptr(x), ops.ptr(tf), None, None, None, ops.ptr(y), 2, 40, 65, 5, 2, ops.IN_MASKMUL, 0.3,
                 ops.ptr(part), 256, ctypes.byref(n), ops.stream())
    torch.cuda.synchronize()
    assert n.value == -1 and float(y.min()) == 7.0 and float(part.abs().max()) == 0.0
    # the Dense forms
    M, K, N = 2048, 256, 256

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.randn(K, N, generator=g) * 0.1)
    h = Holder(); flat = layers.FlatParams(h, 'cuda')
    planes = ops._DenseSplit.get(h.w, K, N, N, 0)
    a = torch.randn(M, K, generator=g).cuda(); c = torch.empty(M, N, device='cuda')
    need = lib.ptts_dense_bf16x6_stats_rows(M, N)
    assert need >= M // 128 and lib.ptts_dense_bf16x6_stats_rows(0, N) == 0
    rows = torch.zeros(need, 2 * N, dtype=torch.float64, device='cuda')
    with pytest.raises(ops._hip.HipLibraryError, match='room for'):
        ops.call('ptts_dense_bf16x6_stats', ops.ptr(a), ops.ptr(planes), None, ops.ptr(c), M, N, K, K, N, ops.IN_NONE, None, None, 0.3,
                 ops.ptr(rows), need - 1, ctypes.byref(n), ops.stream())
    sc = torch.ones(N + 1, device='cuda')
    with pytest.raises(ops._hip.HipLibraryError, match='aligned'):
        ops.call('ptts_dense_bf16x6_bwd_affine', ops.ptr(a), ops.ptr(planes), ops.ptr(c), M, N, K, K, N, ops.ptr(c), ops.ptr(sc[1:]), ops.ptr(sc[1:]), 0.3,
                 ops.ptr(rows), need, ctypes.byref(n), ops.stream())
    with pytest.raises(ops._hip.HipLibraryError, match='bad args'):
        ops.call('ptts_partial_rows_sum', ops.ptr(rows), 0, 2 * N, ops.ptr(rows), ops.stream())
    with pytest.raises(ops._hip.HipLibraryError, match='aligned'):
        ops.call('ptts_conv2d_mfma_bwd_fused_affine', ops.ptr(x), ops.ptr(x), ops.ptr(ops._C2M.table(w, True)), ops.ptr(y), ops.ptr(rows), rows.numel() * 8,
                 ctypes.byref(n), ctypes.byref(n), 2, 40, 65, 5, 2, 0.3, ops.ptr(sc[1:]), ops.ptr(sc[1:]), ops.stream())
    torch.cuda.synchronize()
    assert float(rows.abs().max()) == 0.0

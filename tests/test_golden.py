"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the fp64 oracle).
CPU: the oracle still reproduces them (guards the checker against drift).
GPU: the HIP path, through the C ABI, matches them (rtol 5e-4, inside the north star's 1e-3)."""
import os

import numpy as np
import pytest
import torch

from oracle import percival_oracle as O

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    with np.load(os.path.join(HERE, name)) as z:
        return {k: z[k] for k in z.files}


def t64(a):
    return torch.tensor(np.asarray(a, dtype=np.float64))


def wlist(d, prefix):
    keys = sorted(k for k in d if k.startswith(prefix))
    return [d[k] for k in keys]


ARCH = O.Arch(425, 65, 17, hiddenwidth=2, ctx_nbcnnlayers=2, ctx_winlen=3, gen_nbcnnlayers=2, gen_nbfilters=2,
              gen_winlen=3, spec_freqlen=3)


# ------------------------------------------------------------------------------------------ CPU
def test_oracle_reproduces_conv2d_vectors():
    d = load('conv2d.npz')
    for (cin, cout) in ((1, 4), (4, 4), (4, 1)):
        for k in (3, 5):
            p = 'c{}{}k{}_'.format(cin, cout, k)
            x, w, b = t64(d[p + 'x']).requires_grad_(True), t64(d[p + 'w']).requires_grad_(True), t64(d[p + 'b']).requires_grad_(True)
            y = O.conv2d_nhwc(O.lrelu(x), w, b)
            y.backward(t64(d[p + 'dy']))
            np.testing.assert_allclose(y.detach().numpy(), d[p + 'y'], rtol=1e-10, atol=1e-12)
            np.testing.assert_allclose(x.grad.numpy(), d[p + 'dx'], rtol=1e-10, atol=1e-12)
            np.testing.assert_allclose(w.grad.numpy(), d[p + 'dw'], rtol=1e-10, atol=1e-12)
            # independent check of the stored forward with the numpy loop restatement
            x64 = d[p + 'x'].astype(np.float64)
            ref = O.np_conv2d_same(np.where(x64 > 0, x64, 0.3 * x64),
                                   d[p + 'w'].astype(np.float64), d[p + 'b'].astype(np.float64))
            np.testing.assert_allclose(d[p + 'y'], ref, rtol=1e-10, atol=1e-12)


def test_oracle_reproduces_wgan_vectors():
    d = load('wgan_testgeom.npz')
    gw, cw = [t64(w) for w in wlist(d, 'gw')], [t64(w) for w in wlist(d, 'cw')]
    X, Y, al = t64(d['X']), t64(d['Y']), t64(d['alpha'])
    np.testing.assert_allclose(O.generator_forward([w.clone() for w in gw], ARCH, X, False).numpy(), d['predict_infer'], rtol=1e-10, atol=1e-12)
    for w in cw: w.requires_grad_(True)
    total, parts = O.critic_step_loss(cw, gw, ARCH, X, Y, al)
    np.testing.assert_allclose([float(total.detach()), float(parts['valid'].detach()), float(parts['fake'].detach()), float(parts['gp'].detach())],
                               d['critic_loss'], rtol=1e-10)
    grads = torch.autograd.grad(total, cw)
    for g_, want in zip(grads, wlist(d, 'cgrad')):
        np.testing.assert_allclose(g_.numpy(), want, rtol=1e-9, atol=1e-12)


# ------------------------------------------------------------------------------------------ GPU
def dev(a):
    return torch.tensor(np.asarray(a, dtype=np.float32)).cuda().contiguous()


def close(got, want, rtol=5e-4, atol=1e-5, what='', kinks=False):
    """kinks: see tests/test_model_gpu.py::close (a LeakyReLU mask flipped by fp32 vs fp64 rounding of a ~0 pre-activation)."""
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, what
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    if kinks and np.linalg.norm(err) <= 2e-3 * np.linalg.norm(want) and err.max() <= 0.05 * np.abs(want).max():
        return
    assert (err <= tol).all(), '{}: {} of {} off, worst {:.3e} (max|want| {:.3e})'.format(what, int((err > tol).sum()), err.size, float(err.max()), float(np.abs(want).max()))


@pytest.mark.gpu
def test_hip_conv2d_matches_golden():
    from percivaltts_amd import ops
    d = load('conv2d.npz')
    for (cin, cout) in ((1, 4), (4, 4), (4, 1)):
        for k in (3, 5):
            p = 'c{}{}k{}_'.format(cin, cout, k)
            x, w, b = dev(d[p + 'x']).requires_grad_(True), dev(d[p + 'w']).requires_grad_(True), dev(d[p + 'b']).requires_grad_(True)
            y = ops.conv2d(ops.Lazy(x, lrelu=True), w, b)
            y.backward(dev(d[p + 'dy']))
            close(y, d[p + 'y'], what=p + 'y')
            close(x.grad, d[p + 'dx'], what=p + 'dx')
            close(w.grad, d[p + 'dw'], atol=1e-4, what=p + 'dw')
            close(b.grad, d[p + 'db'], atol=1e-4, what=p + 'db')


@pytest.mark.gpu
def test_hip_wgan_steps_match_golden():
    import percivaltts_amd
    from percivaltts_amd import vocoders, modeltts_common, networks_critic, optimizertts_wgan
    d = load('wgan_testgeom.npz')
    cfg = percivaltts_amd.configuration()
    cfg.arch_hiddenwidth = 2; cfg.arch_ctx_nbcnnlayers = 2; cfg.arch_ctx_winlen = 3
    cfg.arch_gen_nbcnnlayers = 2; cfg.arch_gen_nbfilters = 2; cfg.arch_gen_winlen = 3; cfg.arch_spec_freqlen = 3
    cfg.train_batch_size = 2
    cfg.train_wgan_critic_LSWGANtransidx = 30.0
    voc = vocoders.VocoderPML(16000, 0.005, 65, 17)
    mod = modeltts_common.DCNNF0SpecNoiseFeatures(425, voc, cfg)
    crit = networks_critic.Critic(voc, 425, cfg)
    mod.kerasmodel.set_weights(wlist(d, 'gw'))
    crit.model.set_weights(wlist(d, 'cw'))
    close(mod.predict(d['X']), d['predict_infer'], what='predict')
    opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype='WLSWGAN', critic=crit)
    opt.prepare()
    X, Y, al = dev(d['X']), dev(d['Y']), dev(d['alpha'])
    with torch.no_grad():
        close(crit.model(Y, X, training=False), d['critic_forward'], what='critic forward')
        close(mod.kerasmodel(X, training=True, memo={'freeze_bn_stats': True}), d['generator_train'], what='generator (batch stats)')
    opt.critic_opti.zero_grad()
    total, (lv, lf, gp) = opt.critic_loss(X, Y, al, training=True)
    close(torch.stack([total, lv, lf, gp]), d['critic_loss'], what='critic loss parts')
    total.backward()
    gmax = max(float(np.abs(g).max()) for g in wlist(d, 'cgrad'))
    for p, want in zip(opt.critic_opti.flat.params, wlist(d, 'cgrad')):
        close(p.grad, want, rtol=1e-3, atol=2e-5 * max(gmax, 1.0), what='critic grad {}'.format(want.shape), kinks=True)
    opt.gen_opti.zero_grad()
    for p in opt.critic_opti.flat.params: p.requires_grad_(False)
    lt, (lw, lls) = opt.generator_loss(X, Y, training=True)
    close(torch.stack([lt, lw, lls]), d['generator_loss'], what='generator loss parts')
    lt.backward()
    ggmax = max(float(np.abs(g).max()) for g in wlist(d, 'ggrad'))
    for p, want in zip(opt.gen_opti.flat.params, wlist(d, 'ggrad')):
        close(p.grad, want, rtol=2e-3, atol=5e-5 * max(ggmax, 1.0), what='generator grad {}'.format(want.shape), kinks=True)
    ws = mod.kerasmodel.weights()
    for k in sorted(k for k in d if k.startswith('gmov')):
        close(ws[int(k[4:])][1], d[k], what='moving statistic ' + k)


@pytest.mark.gpu
def test_hip_adam_and_lstm_match_golden():
    from percivaltts_amd import ops
    d = load('adam_lstm.npz')
    p = dev(d['adam_p0']); m = torch.zeros_like(p); v = torch.zeros_like(p)
    step = torch.zeros((), dtype=torch.int32, device='cuda')
    for t in (1, 2, 3):
        ops.adam_keras_step_(p, dev(d['adam_g%d' % t]), m, v, step, 1e-4, 0.5, 0.9, 1e-7)
        close(p, d['adam_p%d' % t], rtol=1e-5, atol=1e-6, what='adam step %d' % t)
    x, W, U, b = [dev(d['lstm_' + n]).requires_grad_(True) for n in ('x', 'W', 'U', 'b')]
    h = ops.lstm(x, W, U, b)
    close(h, d['lstm_h'], what='lstm h')
    h.backward(dev(d['lstm_dh']))
    for n, t in (('dx', x), ('dW', W), ('dU', U), ('db', b)):
        close(t.grad, d['lstm_' + n], atol=1e-4, what='lstm ' + n)

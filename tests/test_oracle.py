"""CPU tests: pin the torch-CPU oracle against plain-numpy loop restatements, the reference's only
known answer (2195 parameters) and finite differences."""
import math

import numpy as np
import torch

from oracle import percival_oracle as O

F64 = torch.float64   # explicit everywhere: a process-wide default dtype would leak into the GPU tests


def test_count_params_known_answers():
    # /root/reference/tests/test_smoke_tensorflowkeras.py:53
    assert O.count_params_generic(425, 4, 3, 65, 17) == 2195
    # test geometry and config 2
    tg = O.Arch(425, 65, 17, hiddenwidth=2, ctx_nbcnnlayers=2, ctx_winlen=3, gen_nbcnnlayers=2, gen_nbfilters=2,
                gen_winlen=3, spec_freqlen=3)
    assert O.count_params(O.generator_weight_shapes(tg)) == 3034
    assert O.count_params(O.critic_weight_shapes(tg)) == 2917
    c2 = O.Arch(601, 65, 20)
    assert O.count_params(O.generator_weight_shapes(c2)) == 4707471
    assert O.count_params(O.critic_weight_shapes(c2)) == 3629941


def test_conv2d_matches_numpy_loops():
    rng = np.random.RandomState(0)
    for (kt, kf, cin, cout, dil, causal) in [(5, 5, 1, 4, 1, False), (3, 3, 2, 2, 1, False), (5, 5, 4, 1, 1, False),
                                             (4, 2, 2, 3, 1, False), (3, 3, 4, 4, 2, True)]:
        x = rng.randn(2, 9, 7, cin)
        w = rng.randn(kt, kf, cin, cout)
        b = rng.randn(cout)
        ref = O.np_conv2d_same(x, w, b, dil_t=dil, causal=causal)
        got = O.conv2d_nhwc(torch.tensor(x), torch.tensor(w), torch.tensor(b), dil_t=dil, causal=causal).numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)


def test_conv1d_matches_numpy_loops():
    rng = np.random.RandomState(1)
    for kw in (3, 4, 21):
        x = rng.randn(2, 30, 5)
        w = rng.randn(kw, 5, 6)
        b = rng.randn(6)
        ref = O.np_conv1d_same(x, w, b)
        got = O.conv1d_ntc(torch.tensor(x), torch.tensor(w), torch.tensor(b)).numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)


def test_lstm_matches_numpy_loops():
    rng = np.random.RandomState(2)
    B, T, In, H = 3, 7, 5, 4
    x = rng.randn(B, T, In)
    W = rng.randn(In, 8 * H) * 0.3
    U = rng.randn(2, H, 4 * H) * 0.3
    b = rng.randn(8 * H) * 0.1
    ref = np.concatenate([O.np_lstm(x, W[:, :4 * H], U[0], b[:4 * H]),
                          O.np_lstm(x, W[:, 4 * H:], U[1], b[4 * H:], reverse=True)], axis=-1)
    got = O.blstm(torch.tensor(x), torch.tensor(W), torch.tensor(U), torch.tensor(b)).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)


def test_bn_and_adam_match_numpy():
    rng = np.random.RandomState(3)
    x = rng.randn(4, 6, 3)
    g, bt = rng.rand(3) + 0.5, rng.randn(3)
    ref, mean, var = O.np_bn_train(x, g, bt)
    mm, mv = torch.zeros(3, dtype=F64), torch.ones(3, dtype=F64)
    got = O.BN(torch.tensor(g), torch.tensor(bt), mm, mv)(torch.tensor(x), True, update=True).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(mm.numpy(), 0.01 * mean, rtol=1e-12)
    np.testing.assert_allclose(mv.numpy(), 0.99 + 0.01 * var, rtol=1e-12)
    p, gr = rng.randn(10), rng.randn(10)
    m, v = np.zeros(10), np.zeros(10)
    pt, mt, vt = torch.tensor(p.copy()), torch.zeros(10, dtype=F64), torch.zeros(10, dtype=F64)
    for t in (1, 2, 3):
        p, m, v = O.np_adam_keras(p, gr, m, v, t, 1e-3, 0.5, 0.9, 1e-7)
        O.adam_keras([pt], [torch.tensor(gr)], [mt], [vt], t, 1e-3, 0.5, 0.9, 1e-7)
    np.testing.assert_allclose(pt.numpy(), p, rtol=1e-12, atol=1e-15)


def _tiny():
    a = O.Arch(11, 9, 3, hiddenwidth=4, ctx_nbcnnlayers=2, ctx_winlen=3, gen_nbcnnlayers=2, gen_nbfilters=2,
               gen_winlen=3, spec_freqlen=3)
    cw = O.random_weights(O.critic_weight_shapes(a), seed=1)
    gw = O.random_weights(O.generator_weight_shapes(a), seed=2)
    g = torch.Generator().manual_seed(3)
    X = torch.rand(2, 8, 11, generator=g, dtype=F64) * 2 - 1
    Y = torch.randn(2, 8, a.outsize, generator=g, dtype=F64)
    al = torch.rand(2, generator=g, dtype=F64)
    return a, cw, gw, X, Y, al


def test_gradient_penalty_against_finite_differences():
    a, cw, gw, X, Y, al = _tiny()
    for w in cw:
        w.requires_grad_(True)
    total, parts = O.critic_step_loss(cw, gw, a, X, Y, al)
    grads = torch.autograd.grad(total, cw)
    # the penalty only sees the spec columns (networks_critic.py:58)
    g = parts['g'].detach()
    assert float(g[:, :, 0].abs().max()) == 0.0 and float(g[:, :, 1 + a.specsize:].abs().max()) == 0.0
    # central differences on a few entries of a few weights (loss is piecewise smooth: tiny step)
    eps = 1e-6
    rng = np.random.RandomState(0)
    for wi in (0, 2, len(cw) - 2, len(cw) - 8):
        w = cw[wi]
        for _ in range(3):
            idx = tuple(rng.randint(0, s) for s in w.shape)
            old = float(w[idx])
            def at(val):
                with torch.no_grad():
                    w[idx] = val
                return float(O.critic_step_loss(cw, gw, a, X, Y, al)[0])
            lp, lm = at(old + eps), at(old - eps)
            at(old)
            fd = (lp - lm) / (2 * eps)
            assert math.isclose(fd, float(grads[wi][idx]), rel_tol=2e-4, abs_tol=1e-7), (wi, idx, fd, float(grads[wi][idx]))


def test_generator_infer_vs_train_and_shapes():
    a, cw, gw, X, Y, al = _tiny()
    out_t = O.generator_forward(gw, a, X, training=True)
    out_i = O.generator_forward(gw, a, X, training=False)
    assert out_t.shape == (2, 8, a.outsize) and out_i.shape == out_t.shape
    assert not torch.allclose(out_t, out_i)
    nm = out_i[:, :, 1 + a.specsize:]
    assert float(nm.min()) > 0 and float(nm.max()) < 1          # sigmoid head (modeltts_common.py:121)


def test_wls_weights():
    w_ls, ww = O.wls_weights(65, 17, 0, 0.25, 32.0)
    assert w_ls.shape == (83,) and w_ls[0] == 1.0 and np.all(w_ls[66:] == 1.0)
    assert abs((1 - w_ls[1 + 32]) - 0.75 * 0.5) < 1e-12      # sigmoid centre
    assert 0 < ww < 0.75
    assert O.critic_runs(0) == 10 and O.critic_runs(25) == 5 and O.critic_runs(500) == 10


def test_bf16_three_way_split_restatement():
    """The oracle's restatement of the bf16x6 arithmetic of csrc/split.hip: the rounding agrees with torch's own
    fp32 -> bfloat16 conversion (round to nearest even), the three planes add up to the fp32 operand EXACTLY, and the six
    kept products reproduce an fp32 convolution to fp32 accuracy while three products do not."""
    g = np.random.RandomState(5)
    x = (g.randn(4096) * np.exp(6 * g.randn(4096))).astype(np.float32)
    x[:4] = [0.0, -0.0, 1.0, -3.0]
    # ties: a value exactly half way between two bf16 numbers must go to the even one
    x[4:6] = np.array([0x3F808000, 0x3F818000], dtype=np.uint32).view(np.float32)
    r = O.np_bf16_round(x)
    assert np.array_equal(r, torch.from_numpy(x).to(torch.bfloat16).float().numpy())
    x1, x2, x3 = O.np_split3_bf16(x)
    assert np.array_equal((x1.astype(np.float64) + x2 + x3).astype(np.float32), x)
    assert np.array_equal(x1.astype(np.float64) + x2.astype(np.float64) + x3.astype(np.float64), x.astype(np.float64))
    for p in (x1, x2, x3):                       # each plane is a bf16 number
        assert np.array_equal(O.np_bf16_round(p), p)
    # a small 'same' Conv1D: six products vs the exact product of the fp32 operands
    a = g.randn(2, 9, 7).astype(np.float32)
    w = (g.randn(5, 7, 3) * 0.3).astype(np.float32)
    b = g.randn(3).astype(np.float32)
    exact = O.np_conv1d_same(a.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    six = O.np_conv1d_same_bf16x6(a, w, b)
    scale = np.abs(exact).mean()
    assert np.abs(six - exact).max() / scale < 2e-7          # what is dropped: x2w3, x3w2, x3w3 ~ 2^-24 of a product
    a1 = O.np_bf16_round(a); w1 = O.np_bf16_round(w)
    plain = O.np_conv1d_same(a1.astype(np.float64), w1.astype(np.float64), b.astype(np.float64))
    assert np.abs(plain - exact).max() / scale > 1e-3        # a plain bf16 product is 4 orders of magnitude worse

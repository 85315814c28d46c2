"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing in percivaltts_amd/ may import this file.

A CPU restatement (PyTorch-CPU, float64 by default, float32 for the timed `cpu_baseline`) of the
reference's WGAN-GP training hot path, with the Keras-2.2 / TF-1.9 layer semantics written out
explicitly.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
the checker.

PARITY UNPINNED: the reference cannot be executed here (Python 2.7 + TensorFlow 1.9 + an empty
`external/pulsemodel` submodule) and its tests hold no numeric vectors for this
path.  The only reference-held known answer, `count_params() == 2195`
(/root/reference/tests/test_smoke_tensorflowkeras.py:53), is reproduced by `count_params_generic`.
Everything else is pinned (a) against plain-numpy loop restatements of the primitives in this file
(np_* functions, written directly from the Keras definitions) and (b) by finite differences, in
tests/test_oracle.py.

What each function follows (file:line under /root/reference/percivaltts/):
  lrelu / pFC / pCNN1D / pCNN2D ............ networktts.py:59-63,116-126
  lstm_keras / blstm ....................... networktts.py:72-96 (kl.LSTM, gates i,f,c,o)
  critic_forward ........................... networks_critic.py:44-96
  generator_forward ........................ modeltts_common.py:65-126 (DCNNF0SpecNoiseFeatures)
  generic_forward .......................... modeltts_common.py:36-58, networktts.py:136-225
  random_weighted_average .................. optimizertts_wgan.py:44-51
  gradient_penalty_loss .................... optimizertts_wgan.py:53-68
  wasserstein_loss ......................... optimizertts_wgan.py:70-71
  specweighted_lse_loss, wls_weights ....... optimizertts_wgan.py:73-79,186-213
  critic_step_loss / generator_step_loss ... optimizertts_wgan.py:115-154,157-213
  adam_keras ............................... keras.optimizers.Adam as configured at :145,172
"""
import math

import numpy as np
import torch
import torch.nn.functional as TF

BN_EPS = 1e-3        # keras BatchNormalization default epsilon
BN_MOMENTUM = 0.99   # keras default momentum
LRELU_ALPHA = 0.3    # keras.layers.LeakyReLU(alpha=0.3) everywhere in the reference


# ------------------------------------------------------------------------------------------------
# plain-numpy loop restatements of the primitives (slow, obviously-right; pin the torch versions)
# ------------------------------------------------------------------------------------------------
def _same_pads(k, dil=1):
    total = (k - 1) * dil
    lo = total // 2            # TF 'SAME': pad_before = total // 2
    return lo, total - lo


def np_conv2d_same(x, w, b=None, dil_t=1, causal=False):
    """x [B,T,F,Cin], w [KT,KF,Cin,Cout] (HWIO), cross-correlation, stride 1, zero 'same' padding."""
    B, T, F, Cin = x.shape
    KT, KF, _, Cout = w.shape
    pt, _ = _same_pads(KT, dil_t)
    if causal:
        pt = (KT - 1) * dil_t
    pf, _ = _same_pads(KF)
    y = np.zeros((B, T, F, Cout), dtype=np.float64)
    for kt in range(KT):
        for kf in range(KF):
            for t in range(T):
                tt = t + kt * dil_t - pt
                if tt < 0 or tt >= T:
                    continue
                for f in range(F):
                    ff = f + kf - pf
                    if ff < 0 or ff >= F:
                        continue
                    y[:, t, f, :] += x[:, tt, ff, :] @ w[kt, kf]
    if b is not None:
        y += b
    return y


def np_conv1d_same(x, w, b=None):
    """x [B,T,Cin], w [KW,Cin,Cout]."""
    B, T, Cin = x.shape
    KW, _, Cout = w.shape
    pl, _ = _same_pads(KW)
    y = np.zeros((B, T, Cout), dtype=np.float64)
    for k in range(KW):
        for t in range(T):
            tt = t + k - pl
            if 0 <= tt < T:
                y[:, t, :] += x[:, tt, :] @ w[k]
    if b is not None:
        y += b
    return y


def np_lstm(x, W, U, b, reverse=False):
    """Keras LSTM(activation=tanh, recurrent_activation=sigmoid, return_sequences): gates i,f,c,o.
    x [B,T,In], W [In,4H], U [H,4H], b [4H]."""
    B, T, _ = x.shape
    H = U.shape[0]
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    h = np.zeros((B, H)); c = np.zeros((B, H))
    out = np.zeros((B, T, H))
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        a = x[:, t] @ W + h @ U + b
        i, f, g, o = sig(a[:, :H]), sig(a[:, H:2 * H]), np.tanh(a[:, 2 * H:3 * H]), sig(a[:, 3 * H:])
        c = f * c + i * g
        h = o * np.tanh(c)
        out[:, t] = h      # Bidirectional re-reverses the backward layer's output: value at time t stays at t
    return out


def np_bn_train(x, gamma, beta):
    """BatchNormalization(axis=-1) in training mode: biased batch variance."""
    ax = tuple(range(x.ndim - 1))
    mean = x.mean(axis=ax)
    var = x.var(axis=ax)
    return gamma * (x - mean) / np.sqrt(var + BN_EPS) + beta, mean, var


def np_adam_keras(p, g, m, v, t, lr, b1, b2, eps):
    """Keras 2.2 Adam.get_updates: t counts from 1; epsilon OUTSIDE the sqrt, no bias-corrected m/v."""
    lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * g * g
    p = p - lr_t * m / (np.sqrt(v) + eps)
    return p, m, v


# ------------------------------------------------------------------------------------------------
# torch (CPU) primitives with Keras semantics; differentiable to any order
# ------------------------------------------------------------------------------------------------
def np_bf16_round(x):
    """fp32 -> bfloat16 (round to nearest even), returned as fp32 values with 8 significant bits.  The rounding the split
    pass of csrc/split.hip applies; restated on the bit pattern as IEEE defines it (no NaN/inf handling: finite inputs)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (r.astype(np.uint32) << 16).view(np.float32)


def np_split3_bf16(x):
    """The three-way bf16 split of an fp32 array: x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (subtractions in
    fp32, exact).  x1 + x2 + x3 == x exactly for finite fp32 x (3 x 8 significant bits cover fp32's 24)."""
    x = np.asarray(x, dtype=np.float32)
    x1 = np_bf16_round(x)
    r1 = (x - x1).astype(np.float32)
    x2 = np_bf16_round(r1)
    r2 = (r1 - x2).astype(np.float32)
    x3 = np_bf16_round(r2)
    return x1, x2, x3


def np_conv1d_same_bf16x6(x, w, b=None):
    """The context Conv1D forward as csrc/split.hip computes it: both fp32 operands split three ways, the six products
    x1w1, x1w2, x2w1, x1w3, x2w2, x3w1 summed (here in fp64: the kernel accumulates in fp32, which the tests allow for)."""
    xs = np_split3_bf16(x)
    ws = np_split3_bf16(w)
    out = 0.0
    for i, j in ((0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)):
        out = out + np_conv1d_same(xs[i].astype(np.float64), ws[j].astype(np.float64), None)
    if b is not None:
        out = out + np.asarray(b, dtype=np.float64)
    return out


def lrelu(x):
    return TF.leaky_relu(x, LRELU_ALPHA)


def conv2d_nhwc(x, w, b=None, dil_t=1, causal=False):
    KT, KF = w.shape[0], w.shape[1]
    pt_lo, pt_hi = _same_pads(KT, dil_t)
    if causal:
        pt_lo, pt_hi = (KT - 1) * dil_t, 0
    pf_lo, pf_hi = _same_pads(KF)
    xn = x.permute(0, 3, 1, 2)                                     # NCHW, H = time, W = freq
    xn = TF.pad(xn, (pf_lo, pf_hi, pt_lo, pt_hi))
    y = TF.conv2d(xn, w.permute(3, 2, 0, 1).contiguous(), b, dilation=(dil_t, 1))   # HWIO -> OIHW
    return y.permute(0, 2, 3, 1)


def conv1d_ntc(x, w, b=None):
    KW = w.shape[0]
    lo, hi = _same_pads(KW)
    xn = TF.pad(x.permute(0, 2, 1), (lo, hi))
    return TF.conv1d(xn, w.permute(2, 1, 0).contiguous(), b).permute(0, 2, 1)


def dense(x, w, b=None):
    y = x @ w
    return y if b is None else y + b


def pgcnn2d_product(x, wa, wb, dil_t=1, causal=False, ba=None, bb=None):
    """The gated product of pGCNN2D (networktts.py:128-134): Conv2D(x) * Conv2D(x, activation=sigmoid), before its
    BatchNormalization / LeakyReLU."""
    return conv2d_nhwc(x, wa, ba, dil_t=dil_t, causal=causal) * torch.sigmoid(conv2d_nhwc(x, wb, bb, dil_t=dil_t, causal=causal))


class BN(object):
    """gamma, beta, moving_mean, moving_var (the Keras weight order)."""
    def __init__(self, gamma, beta, mm, mv):
        self.gamma, self.beta, self.mm, self.mv = gamma, beta, mm, mv

    def __call__(self, x, training, update=False, unbiased_moving=False):
        ax = tuple(range(x.dim() - 1))
        if training:
            mean = x.mean(dim=ax)
            var = x.var(dim=ax, unbiased=False)
            if update:
                n = x.numel() // x.shape[-1]
                vm = var * n / (n - 1) if (unbiased_moving and n > 1) else var
                with torch.no_grad():
                    self.mm.mul_(BN_MOMENTUM).add_((1 - BN_MOMENTUM) * mean)
                    self.mv.mul_(BN_MOMENTUM).add_((1 - BN_MOMENTUM) * vm)
        else:
            mean, var = self.mm, self.mv
        return self.gamma * (x - mean) / torch.sqrt(var + BN_EPS) + self.beta


def lstm_keras(x, W, U, b, reverse=False):
    B, T, _ = x.shape
    H = U.shape[0]
    h = x.new_zeros((B, H)); c = x.new_zeros((B, H))
    xp = x @ W + b
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        a = xp[:, t] + h @ U
        i, f = torch.sigmoid(a[:, :H]), torch.sigmoid(a[:, H:2 * H])
        g, o = torch.tanh(a[:, 2 * H:3 * H]), torch.sigmoid(a[:, 3 * H:])
        c = f * c + i * g
        h = o * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs, dim=1)


def blstm(x, W, U, b):
    """Bidirectional(concat).  Combined layout used by the build: W [In, 2*4H] = [W_fwd | W_bwd], U [2,H,4H],
    b [2*4H]."""
    H = U.shape[1]
    G = 4 * H
    hf = lstm_keras(x, W[:, :G], U[0], b[:G], reverse=False)
    hb = lstm_keras(x, W[:, G:], U[1], b[G:], reverse=True)
    return torch.cat([hf, hb], dim=-1)


# ------------------------------------------------------------------------------------------------
# architecture description shared with the tests: weights are consumed IN CREATION ORDER of the
# reference's Keras layers; BatchNorm contributes gamma, beta, moving_mean, moving_var.
# ------------------------------------------------------------------------------------------------
class Arch(object):
    def __init__(self, ctxsize, specsize, noisesize, hiddenwidth=256, ctx_nbcnnlayers=1, ctx_winlen=21,
                 gen_nbcnnlayers=8, gen_nbfilters=4, gen_winlen=5, spec_freqlen=5, vuvsize=0,
                 gen_gated=False, gen_dilations=None, gen_causal=False):
        self.ctxsize, self.specsize, self.noisesize, self.vuvsize = ctxsize, specsize, noisesize, vuvsize
        self.H, self.nctx, self.kctx = hiddenwidth, ctx_nbcnnlayers, ctx_winlen
        self.L, self.C, self.kt, self.kf = gen_nbcnnlayers, gen_nbfilters, gen_winlen, spec_freqlen
        # generator spectral branch built from pGCNN2D (networktts.py:128-134, the commented alternative at
        # modeltts_common.py:99); time dilations / causal padding are build extensions (BASELINE configs[4]) that reduce
        # to the reference's layer at dilation 1, symmetric padding
        self.gated, self.dilations, self.causal = bool(gen_gated), gen_dilations, bool(gen_causal)

    @property
    def outsize(self):
        return 1 + self.specsize + self.noisesize + self.vuvsize


class _Take(object):
    def __init__(self, weights):
        self.w = list(weights)
        self.i = 0

    def __call__(self, n=1):
        out = self.w[self.i:self.i + n]
        assert len(out) == n, 'oracle: ran out of weights at {}'.format(self.i)
        self.i += n
        return out[0] if n == 1 else out

    def done(self):
        assert self.i == len(self.w), 'oracle: {} weights unused'.format(len(self.w) - self.i)


def critic_weight_shapes(a):
    """networks_critic.py:44-96 with bn=False (biases everywhere)."""
    s = []
    if a.L > 0:
        cin = 1
        for _ in range(a.L):
            s += [(a.kt, a.kf, cin, a.C), (a.C,)]
            cin = a.C
        specw = a.specsize * a.C
    else:
        cin = a.specsize
        for _ in range(3):
            s += [(cin, a.H), (a.H,)]
            cin = a.H
        specw = a.H
    cin = a.ctxsize
    for _ in range(a.nctx):
        s += [(a.kctx, cin, a.H), (a.H,)]
        cin = a.H
    for _ in range(2):
        s += [(cin, a.H), (a.H,)]
        cin = a.H
    cin = specw + a.H
    for _ in range(3):
        s += [(cin, a.H), (a.H,)]
        cin = a.H
    s += [(a.H, 1), (1,)]
    return s


def generator_weight_shapes(a):
    """modeltts_common.py:65-126: every pFC/pCNN1D/pCNN2D is bias-free + BatchNorm (4 vectors)."""
    s = []
    bn = lambda c: [(c,), (c,), (c,), (c,)]
    cin = a.ctxsize
    for _ in range(a.nctx):
        s += [(a.kctx, cin, a.H)] + bn(a.H)
        cin = a.H
    for _ in range(2):
        s += [(cin, a.H)] + bn(a.H)
        cin = a.H
    s += [(a.H, 8 * a.H), (2, a.H, 4 * a.H), (8 * a.H,)]          # BLSTM (combined layout)
    s += [(2 * a.H, 1), (1,)]                                      # f0 head
    s += [(a.H, a.specsize), (a.specsize,)]                        # spec projection
    cin = 1
    for _ in range(a.L):
        s += [(a.kt, a.kf, cin, a.C)] * (2 if getattr(a, 'gated', False) else 1) + bn(a.C)     # pGCNN2D: two kernels
        cin = a.C
    s += [(a.kt, a.kf, cin, 1), (1,)]                              # final Conv2D, bias, linear
    cin = a.H
    for _ in range(a.L // 2):                                      # Python-2 integer division at :119
        s += [(cin, a.H)] + bn(a.H)
        cin = a.H
    s += [(cin, a.noisesize), (a.noisesize,)]
    return s


def count_params(shapes):
    return int(sum(int(np.prod(s)) for s in shapes))


def count_params_generic(ctxsize, hidden, nlayers, specsize, nmsize):
    """Generic(layertypes=['FC']*n) + network_final for VocoderPML without MLPG
    (modeltts_common.py:36-58; networktts.py:59-63,192-199).  The reference asserts 2195 for
    (425, 4, 3, 65, 17) at tests/test_smoke_tensorflowkeras.py:53."""
    n, cin = 0, ctxsize
    for _ in range(nlayers):
        n += cin * hidden + 4 * hidden        # Dense(use_bias=False) + BN(gamma,beta,moving mean/var)
        cin = hidden
    n += cin * (1 + specsize) + (1 + specsize)   # lo_f0spec
    n += cin * nmsize + nmsize                   # lo_nm
    return n


def bf16_st(x):
    """Round to bfloat16 (through fp32, as the kernels hold fp32 values) with a straight-through gradient: the value the
    bf16-storage path keeps in HBM / feeds to the matrix cores, differentiable like the identity."""
    return x + (x.detach().to(torch.float32).to(torch.bfloat16).to(x.dtype) - x.detach())


def critic_forward(weights, a, features, ctx, bf16_stack=False):
    """D(features [B,T,out], ctx [B,T,ctxsize]) -> [B,T,1].
    bf16_stack (build extension, BASELINE configs[2]; the reference is fp32 throughout):
      True / 'layers'  the 4 -> 4 channel layers of the Conv2D stack multiply in bf16 -- activation rounded to bf16 after the
                       LeakyReLU, the kernel's bf16 copy, exact products, wide accumulation -- and the maps between them are
                       stored as bf16; the first layer, the map handed on to the dense layers, the biases and the master
                       weights stay fp32 (csrc/conv2d_mfma.hip with one plane);
      'chain'          the whole stack per launch (csrc/conv2d_chain.hip): the spectrum and EVERY kernel rounded to bf16, every
                       layer's output stored as bf16 AFTER the LeakyReLU (a_l = bf16(lrelu(z_l)), z_l never rounded), biases fp32."""
    take = _Take(weights)
    B, T = features.shape[0], features.shape[1]
    spec = features[:, :, 1:1 + a.specsize]                        # networks_critic.py:58
    if a.L > 0 and bf16_stack == 'chain':
        h = bf16_st(spec.reshape(B, T, a.specsize, 1))
        for li in range(a.L):
            w, b = take(2)
            h = bf16_st(lrelu(conv2d_nhwc(h, bf16_st(w), b)))
        h = h.reshape(B, T, a.specsize * a.C)
    elif a.L > 0:
        h = spec.reshape(B, T, a.specsize, 1)
        z = None
        for li in range(a.L):
            w, b = take(2)
            if bf16_stack and li >= 1:
                z = conv2d_nhwc(bf16_st(lrelu(z)), bf16_st(w), b)
                if li < a.L - 1:
                    z = bf16_st(z)                                 # stored as bf16
            else:
                z = conv2d_nhwc(h if li == 0 else lrelu(z), w, b)
        h = lrelu(z)
        h = h.reshape(B, T, a.specsize * a.C)                      # f-major, c-minor
    else:
        h = spec
        for _ in range(3):
            w, b = take(2)
            h = lrelu(dense(h, w, b))
    c = ctx
    for _ in range(a.nctx):
        w, b = take(2)
        c = lrelu(conv1d_ntc(c, w, b))
    for _ in range(2):
        w, b = take(2)
        c = lrelu(dense(c, w, b))
    p = torch.cat([h, c], dim=-1)
    for _ in range(3):
        w, b = take(2)
        p = lrelu(dense(p, w, b))
    w, b = take(2)
    out = dense(p, w, b)
    take.done()
    return out


def generator_forward(weights, a, ctx, training, update_moving=False):
    """G(ctx [B,T,ctxsize]) -> [B,T,1+spec+nm].  BN over all leading axes; 4-D BN moving variance uses the
    Bessel-corrected batch variance (TF fused batch norm), 3-D BN the biased one."""
    take = _Take(weights)
    B, T = ctx.shape[0], ctx.shape[1]

    def bnl(x, fused4d=False):
        g, bt, mm, mv = take(4)
        return lrelu(BN(g, bt, mm, mv)(x, training, update_moving, unbiased_moving=fused4d))

    h = ctx
    for _ in range(a.nctx):
        h = bnl(conv1d_ntc(h, take()))
    for _ in range(2):
        h = bnl(dense(h, take()))
    W, U, b = take(3)
    f0 = blstm(h, W, U, b)
    w, b = take(2)
    f0 = dense(f0, w, b)
    w, b = take(2)
    s = dense(h, w, b).reshape(B, T, a.specsize, 1)
    for li in range(a.L):
        if getattr(a, 'gated', False):
            dil = a.dilations[li % len(a.dilations)] if a.dilations else 1
            wa, wb = take(2)
            s = bnl(pgcnn2d_product(s, wa, wb, dil, a.causal), fused4d=True)
        else:
            s = bnl(conv2d_nhwc(s, take()), fused4d=True)
    w, b = take(2)
    s = conv2d_nhwc(s, w, b, causal=bool(getattr(a, 'gated', False) and a.causal)).reshape(B, T, a.specsize)
    n = h
    for _ in range(a.L // 2):
        n = bnl(dense(n, take()))
    w, b = take(2)
    n = torch.sigmoid(dense(n, w, b))
    take.done()
    return torch.cat([f0, s, n], dim=-1)


# ------------------------------------------------------------------------------------------------
# losses and steps
# ------------------------------------------------------------------------------------------------
def random_weighted_average(real, fake, alpha):
    """alpha [B] injected (the reference draws it from TF's RNG, optimizertts_wgan.py:50)."""
    al = alpha.reshape(-1, 1, 1)
    return al * real + (1 - al) * fake


def gradient_penalty_loss(v_hat, x_hat):
    g = torch.autograd.grad(v_hat.sum(), x_hat, create_graph=True)[0]
    n = torch.sqrt((g * g).sum(dim=tuple(range(1, g.dim()))))
    return ((1 - n) ** 2).mean(), g


def wasserstein_loss(target, pred):
    return (target * pred).mean()


def specweighted_lse_loss(y, yhat, w):
    return (((y - yhat) ** 2) * w).mean()


def nonlin_sigmoidparm(x, c=0.0, f=1.0):
    """backend_tensorflow.py:107-109"""
    return 1.0 / (1.0 + np.exp(-(x - c) * f))


def wls_weights(specsize, noisesize, vuvsize, LScoef, transidx, transcoef=1.0 / 8.0):
    """optimizertts_wgan.py:186-213 with train_wgan_critic_use_WGAN_incnoisefeature=False.
    Returns (w_ls [out], wgan_weight scalar).  `transidx` replaces sp.freq2fwspecidx(...) of the absent
    pulsemodel submodule."""
    els = [np.zeros(1)]
    if LScoef == 0.0:
        els.append(np.ones(specsize))
    else:
        els.append(nonlin_sigmoidparm(np.arange(specsize, dtype=np.float32), transidx, transcoef))
    els.append(np.zeros(noisesize))
    if vuvsize > 0:
        els.append(np.zeros(1))
    w = np.hstack(els) * (1.0 - LScoef)
    return 1.0 - w, float(np.mean(w))


def critic_step_loss(cw, gw, a, X, Y, alpha, gp_lambda=10.0, training=True, bf16_stack=False):
    """One critic loss (optimizertts_wgan.py:115-154): returns (total, parts dict).  training=False is the evaluation of
    update_validation_cost (:259-260, `critic_model.evaluate`, learning phase 0): G's BatchNorm uses its moving statistics;
    the gradient penalty is still part of the loss."""
    with torch.no_grad():
        fake = generator_forward(gw, a, X, training=training)      # frozen G; learning phase 1 -> batch statistics
    valid = critic_forward(cw, a, Y, X, bf16_stack)
    fake_v = critic_forward(cw, a, fake, X, bf16_stack)
    x_hat = random_weighted_average(Y, fake, alpha).detach().requires_grad_(True)
    v_hat = critic_forward(cw, a, x_hat, X, bf16_stack)
    gp, g = gradient_penalty_loss(v_hat, x_hat)
    l_valid = wasserstein_loss(-1.0, valid)
    l_fake = wasserstein_loss(+1.0, fake_v)
    total = l_valid + l_fake + gp_lambda * gp
    return total, {'valid': l_valid, 'fake': l_fake, 'gp': gp, 'g': g, 'fake_sample': fake}


def generator_step_loss(cw, gw, a, X, Y, errtype='WLSWGAN', w_ls=None, wgan_weight=1.0, update_moving=True, training=True):
    """Generator loss (optimizertts_wgan.py:157-213); training=False: `generator_model.evaluate` of :250-257 (BN inference)."""
    pred = generator_forward(gw, a, X, training=training, update_moving=update_moving and training)
    valid = critic_forward(cw, a, pred, X)
    l_w = wasserstein_loss(-1.0, valid)
    if errtype == 'WGAN':
        return l_w, {'wgan': l_w, 'pred': pred}
    l_ls = specweighted_lse_loss(Y, pred, w_ls)
    return wgan_weight * l_w + l_ls, {'wgan': l_w, 'ls': l_ls, 'pred': pred}


def adam_keras(params, grads, ms, vs, t, lr, b1, b2, eps=1e-7):
    """In-place Keras-2.2 Adam over lists of tensors; t is the 1-based step count."""
    lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    with torch.no_grad():
        for p, g, m, v in zip(params, grads, ms, vs):
            m.mul_(b1).add_((1 - b1) * g)
            v.mul_(b2).add_((1 - b2) * g * g)
            p.sub_(lr_t * m / (torch.sqrt(v) + eps))


def critic_runs(generator_updates):
    """optimizertts_wgan.py:225-228"""
    return 10 if (generator_updates < 25) or (generator_updates % 500 == 0) else 5


# ------------------------------------------------------------------------------------------------
# initialisers (Keras defaults), used to create identical weights for the build and the oracle
# ------------------------------------------------------------------------------------------------
def glorot_uniform(shape, gen):
    if len(shape) == 1:
        return torch.zeros(shape, dtype=torch.float64)
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * lim


def random_weights(shapes, seed=0, scale_bias=0.1, bn_noise=True):
    """Test weights: glorot kernels, NON-zero biases / BN parameters so that every term is exercised."""
    gen = torch.Generator().manual_seed(seed)
    out = []
    i = 0
    while i < len(shapes):
        s = shapes[i]
        if len(s) > 1:
            if len(s) == 3 and s[0] == 2:        # BLSTM recurrent [2,H,4H]
                out.append(torch.randn(s, generator=gen, dtype=torch.float64) / math.sqrt(s[1]))
            else:
                out.append(glorot_uniform(s, gen))
            i += 1
        else:
            # 1-D: bias, or the 4 BN vectors (recognised by four equal 1-D shapes in a row)
            if i + 3 < len(shapes) and all(shapes[i + k] == s for k in range(4)) and bn_noise:
                out.append(1.0 + 0.2 * torch.randn(s, generator=gen, dtype=torch.float64))     # gamma
                out.append(0.1 * torch.randn(s, generator=gen, dtype=torch.float64))           # beta
                out.append(0.1 * torch.randn(s, generator=gen, dtype=torch.float64))           # moving mean
                out.append(1.0 + 0.2 * torch.rand(s, generator=gen, dtype=torch.float64))      # moving var
                i += 4
            else:
                out.append(scale_bias * torch.randn(s, generator=gen, dtype=torch.float64))
                i += 1
    return out

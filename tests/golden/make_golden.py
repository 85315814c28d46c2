"""Generates the golden vectors of tests/golden/*.npz from the CPU oracle (fp64), seeds in the file.
Run from the repo root:  python tests/golden/make_golden.py
The reference itself cannot be executed offline, so these vectors pin the ORACLE (and the HIP
path against it); inputs and weights are stored as float32 values so that both sides start from identical numbers."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import percival_oracle as O   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_default_dtype(torch.float64)   # this script runs as its own process


def r32(t):
    """round to float32 values, keep float64 storage for the computation"""
    return t.to(torch.float32).to(torch.float64)


def conv2d_vectors():
    out = {}
    g = torch.Generator().manual_seed(1001)
    for (cin, cout) in ((1, 4), (4, 4), (4, 1)):
        for k in (3, 5):
            x = r32(torch.randn(2, 12, 9, cin, generator=g)).requires_grad_(True)
            w = r32(torch.randn(k, k, cin, cout, generator=g) * 0.3).requires_grad_(True)
            b = r32(torch.randn(cout, generator=g)).requires_grad_(True)
            dy = r32(torch.randn(2, 12, 9, cout, generator=g))
            y = O.conv2d_nhwc(O.lrelu(x), w, b)
            y.backward(dy)
            key = 'c{}{}k{}_'.format(cin, cout, k)
            for n, t in (('x', x), ('w', w), ('b', b), ('dy', dy), ('y', y), ('dx', x.grad), ('dw', w.grad), ('db', b.grad)):
                out[key + n] = t.detach().numpy().astype(np.float32 if n in ('x', 'w', 'b', 'dy') else np.float64)
    np.savez_compressed(os.path.join(HERE, 'conv2d.npz'), **out)


def wgan_vectors():
    """Critic and generator steps at the geometry of the reference's DCNN/WGAN smoke test
    (tests/test_smoke_tensorflowkeras.py:184-203): ctx 425, spec 65, nm 17, hidden 2, 2 ctx conv k3, 2 conv2d 3x3 x2."""
    a = O.Arch(425, 65, 17, hiddenwidth=2, ctx_nbcnnlayers=2, ctx_winlen=3, gen_nbcnnlayers=2, gen_nbfilters=2,
               gen_winlen=3, spec_freqlen=3)
    gw = [r32(w) for w in O.random_weights(O.generator_weight_shapes(a), seed=21)]
    cw = [r32(w) for w in O.random_weights(O.critic_weight_shapes(a), seed=22)]
    g = torch.Generator().manual_seed(1002)
    B, T = 2, 16
    X = r32(torch.rand(B, T, 425, generator=g) * 2 - 1)
    Y = r32(torch.randn(B, T, a.outsize, generator=g))
    al = r32(torch.rand(B, generator=g))
    out = {'X': X.numpy().astype(np.float32), 'Y': Y.numpy().astype(np.float32), 'alpha': al.numpy().astype(np.float32)}
    for i, w in enumerate(gw): out['gw%03d' % i] = w.numpy().astype(np.float32)
    for i, w in enumerate(cw): out['cw%03d' % i] = w.numpy().astype(np.float32)
    out['predict_infer'] = O.generator_forward([w.clone() for w in gw], a, X, training=False).numpy()
    out['generator_train'] = O.generator_forward([w.clone() for w in gw], a, X, training=True).numpy()
    out['critic_forward'] = O.critic_forward(cw, a, Y, X).numpy()
    cwg = [w.clone().requires_grad_(True) for w in cw]
    total, parts = O.critic_step_loss(cwg, [w.clone() for w in gw], a, X, Y, al, gp_lambda=10.0)
    grads = torch.autograd.grad(total, cwg)
    out['critic_loss'] = np.array([float(total), float(parts['valid']), float(parts['fake']), float(parts['gp'])])
    for i, gr in enumerate(grads): out['cgrad%03d' % i] = gr.numpy()
    # generator step (WLSWGAN, LScoef .25, transition index 30)
    w_ls, ww = O.wls_weights(65, 17, 0, 0.25, 30.0)
    shapes = O.generator_weight_shapes(a)
    train_idx, i = [], 0
    while i < len(shapes):
        if len(shapes[i]) == 1 and i + 3 < len(shapes) and all(shapes[i + k] == shapes[i] for k in range(4)):
            train_idx += [i, i + 1]; i += 4
        else:
            train_idx.append(i); i += 1
    gwg = [w.clone() for w in gw]
    for i in train_idx: gwg[i].requires_grad_(True)
    lt, lp = O.generator_step_loss(cw, gwg, a, X, Y, 'WLSWGAN', torch.tensor(w_ls), ww, update_moving=True)
    gg = torch.autograd.grad(lt, [gwg[i] for i in train_idx], allow_unused=True)
    out['generator_loss'] = np.array([float(lt), float(lp['wgan']), float(lp['ls'])])
    out['train_idx'] = np.array(train_idx)
    for k, (i, gr) in enumerate(zip(train_idx, gg)):
        out['ggrad%03d' % k] = (gr if gr is not None else torch.zeros_like(gwg[i])).numpy()
    for i, w in enumerate(gwg):      # BN moving statistics after the training forward
        if i not in train_idx: out['gmov%03d' % i] = w.detach().numpy()
    np.savez_compressed(os.path.join(HERE, 'wgan_testgeom.npz'), **out)


def adam_lstm_vectors():
    g = torch.Generator().manual_seed(1003)
    out = {}
    p = r32(torch.randn(257, generator=g)); m = torch.zeros(257); v = torch.zeros(257)
    out['adam_p0'] = p.numpy().astype(np.float32)
    for t in (1, 2, 3):
        gr = r32(torch.randn(257, generator=g))
        out['adam_g%d' % t] = gr.numpy().astype(np.float32)
        O.adam_keras([p], [gr], [m], [v], t, 1e-4, 0.5, 0.9, 1e-7)
        out['adam_p%d' % t] = p.numpy().copy()
    Bn, T, In, H = 3, 7, 5, 16
    x = r32(torch.randn(Bn, T, In, generator=g)).requires_grad_(True)
    W = r32(torch.randn(In, 8 * H, generator=g) * 0.4).requires_grad_(True)
    U = r32(torch.randn(2, H, 4 * H, generator=g) * 0.25).requires_grad_(True)
    b = r32(torch.randn(8 * H, generator=g) * 0.2).requires_grad_(True)
    dh = r32(torch.randn(Bn, T, 2 * H, generator=g))
    h = O.blstm(x, W, U, b)
    h.backward(dh)
    for n, t in (('x', x), ('W', W), ('U', U), ('b', b), ('dh', dh)):
        out['lstm_' + n] = t.detach().numpy().astype(np.float32)
    for n, t in (('h', h), ('dx', x.grad), ('dW', W.grad), ('dU', U.grad), ('db', b.grad)):
        out['lstm_' + n] = t.detach().numpy()
    np.savez_compressed(os.path.join(HERE, 'adam_lstm.npz'), **out)


if __name__ == '__main__':
    conv2d_vectors()
    wgan_vectors()
    adam_lstm_vectors()
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)), 'bytes')

"""A small Keras-functional-style graph layer over the HIP ops.

The reference builds its networks with the tf.keras functional API (`kl.Dense(...)(x)`,
`keras.Model(inputs, outputs)`; networktts.py:59-225, networks_critic.py:44-96,
modeltts_common.py:36-126).  To keep those builder functions source-compatible this module offers
the same idiom: calling a layer on a `Node` returns a `Node`; `Model(inputs, outputs)` is a
torch.nn.Module that evaluates the graph.  Values flowing along edges are tensors, `ops.Lazy`
(a pending BatchNorm-affine + LeakyReLU that the consumer's kernel applies on load) or `LazyConcat`.

Layouts and initialisers are the Keras ones (kernel [in,out], HWIO conv kernels, glorot_uniform,
orthogonal recurrent kernels, unit forget bias, BN gamma/beta/moving_mean/moving_variance), so
`count_params()` agrees with the reference's known answer (tests/test_smoke_tensorflowkeras.py:53).
"""
import math

import numpy as np
import os

import torch
import torch.nn as nn

from . import ops
from .ops import Lazy


# --------------------------------------------------------------------------------------------
# values
# --------------------------------------------------------------------------------------------
class LazyConcat(object):
    """Concatenation along the last axis that has not been materialised: a Dense consumer multiplies each
    part with its slice of the kernel instead (and can reuse the product of a part shared by several
    evaluations)."""
    def __init__(self, parts):
        self.parts = [ops.as_lazy(p) for p in parts]

    @property
    def shape(self):
        return tuple(self.parts[0].shape[:-1]) + (sum(p.shape[-1] for p in self.parts),)

    def tensor(self):
        return torch.cat([p.tensor() for p in self.parts], dim=-1)


def to_tensor(v):
    if isinstance(v, (Lazy, LazyConcat)):
        return v.tensor()
    return v


# --------------------------------------------------------------------------------------------
# initialisers (numpy global RNG: the reference seeds it with 123 at import, percivaltts.py:30-33)
# --------------------------------------------------------------------------------------------
def glorot_uniform(shape):
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return torch.from_numpy(np.random.uniform(-lim, lim, size=shape).astype(np.float32))


def orthogonal(shape):
    a = np.random.normal(0.0, 1.0, (shape[0], int(np.prod(shape[1:]))))
    u, _, v = np.linalg.svd(a, full_matrices=False)
    q = u if u.shape == a.shape else v
    return torch.from_numpy(q.reshape(shape).astype(np.float32))


# --------------------------------------------------------------------------------------------
# graph plumbing
# --------------------------------------------------------------------------------------------
class Node(object):
    """A symbolic tensor: `shape` is the tail after (batch, time), e.g. (D,) or (F, C)."""
    def __init__(self, layer, parents, shape, name=None):
        self.layer, self.parents, self.shape, self.name = layer, list(parents), tuple(shape), name


class Layer(nn.Module):
    """Base: `__call__` on Node(s) builds the graph, `compute` runs it."""
    def __init__(self, name=None):
        super(Layer, self).__init__()
        self.lname = name
        self.built = False

    def build(self, in_shapes):
        pass

    def out_shape(self, in_shapes):
        return in_shapes[0]

    def connect(self, inputs):
        single = isinstance(inputs, Node)
        nodes = [inputs] if single else list(inputs)
        shapes = [n.shape for n in nodes]
        if not self.built:
            self.build(shapes)
            self.built = True
        return Node(self, nodes, self.out_shape(shapes), self.lname)

    def __call__(self, *args, **kwargs):
        if len(args) == 1 and not kwargs and (isinstance(args[0], Node) or (
                isinstance(args[0], (list, tuple)) and len(args[0]) > 0 and isinstance(args[0][0], Node))):
            return self.connect(args[0])
        return super(Layer, self).__call__(*args, **kwargs)

    def compute(self, vals, training, memo):
        raise NotImplementedError

    def weights(self):
        """(name, tensor) in Keras weight order: trainable first as created, BN moving statistics last."""
        out = [(k, p) for k, p in self.named_parameters(recurse=False)]
        out += [(k, b) for k, b in self.named_buffers(recurse=False)]
        return out

    def config(self):
        return {}


class InputLayer(Layer):
    def __init__(self, shape, name=None):
        super(InputLayer, self).__init__(name)
        self.shape = tuple(shape)


def Input(shape, name=None):
    """keras.layers.Input(shape=(None, D)): the leading None is the time axis."""
    tail = tuple(s for s in shape[1:])
    lay = InputLayer(tail, name)
    return Node(lay, [], tail, name)


class Dense(Layer):
    """keras.layers.Dense(units, use_bias, activation) on the last axis."""
    def __init__(self, units, use_bias=True, activation=None, name=None):
        super(Dense, self).__init__(name)
        self.units, self.use_bias, self.activation = int(units), bool(use_bias), activation
        assert activation in (None, 'linear', 'sigmoid', 'tanh') or callable(activation) or isinstance(activation, tuple)

    def build(self, in_shapes):
        k = in_shapes[0][-1]
        self.kernel = nn.Parameter(glorot_uniform((k, self.units)))
        if self.use_bias:
            self.bias = nn.Parameter(torch.zeros(self.units))
        else:
            self.bias = None

    def out_shape(self, in_shapes):
        return tuple(in_shapes[0][:-1]) + (self.units,)

    def prime_part(self, i, widths, part, memo):
        """Product of concat part i with its rows of the kernel, stored in `memo` for every evaluation that shares it."""
        off = sum(widths[:i])
        part = ops.as_lazy(part)
        key = ('dense_part', id(self), i, id(part.z))
        if key not in memo:
            memo[key] = ops.dense(part, self.kernel[off:off + widths[i]], None)

    def compute(self, vals, training, memo):
        v = vals[0]
        if isinstance(v, LazyConcat) and memo is not None and len(v.parts) == 2 and \
                ('dense_part', id(self), 1, id(v.parts[1].z)) in memo and torch.is_tensor(v.parts[0].z) and v.parts[0].z.is_cuda:
            # two parts, the second one's product already made (shared by the stacked evaluations of forward_multi: the critic's
            # context branch): it joins the first part's product inside that product's store -- no add pass, no broadcast copy
            y = memo[('dense_part', id(self), 1, id(v.parts[1].z))]
            k0 = v.parts[0].shape[-1]
            lead = 1
            for d_ in v.parts[0].shape[:-1]: lead *= d_
            if lead % (y.numel() // y.shape[-1]) == 0:
                z = ops.dense(v.parts[0], self.kernel[:k0], self.bias, res=y)
                return _apply_activation(z, self.activation)
        if isinstance(v, LazyConcat):
            z, off = None, 0
            for i, part in enumerate(v.parts):
                k = part.shape[-1]
                wpart = self.kernel[off:off + k]
                off += k
                key = ('dense_part', id(self), i, id(part.z))
                bias = self.bias if i == 0 else None
                if memo is not None and i > 0 and key in memo:
                    y = memo[key]
                else:
                    y = ops.dense(part, wpart, bias)
                    if memo is not None and i > 0:
                        memo[key] = y
                if z is not None and y.shape[0] != z.shape[0]:
                    # a part shared by several stacked evaluations (forward_multi with a k*B variant): broadcast it
                    k = z.shape[0] // y.shape[0]
                    z = (z.view((k,) + tuple(y.shape)) + y).view(z.shape)
                else:
                    z = y if z is None else z + y
        elif getattr(self, 'bn_follows', False) and training and self.activation in (None, 'linear'):
            # the BatchNormalization behind this layer takes its statistics from sums the product's launch leaves (ops._BNStats)
            ops._BNStats.want, ops._BNStats.last = True, None
            try:
                z = ops.dense(v, self.kernel, self.bias)
            finally:
                ops._BNStats.want = False
            if ops._BNStats.last is not None:
                z._ptts_bn_partials, ops._BNStats.last = ops._BNStats.last, None
            return z
        else:
            z = ops.dense(v, self.kernel, self.bias)
        return _apply_activation(z, self.activation)


def _dense_compute_pair(self, vals0, vals1, training, memo):
    """Dense of two evaluations whose inputs lie back to back (Model.forward_multi_at's lockstep form): one forward launch (ops.dense_pair)."""
    v0, v1 = vals0[0], vals1[0]
    if isinstance(v0, LazyConcat) and isinstance(v1, LazyConcat) and memo is not None and len(v0.parts) == 2 and len(v1.parts) == 2 and \
            v0.parts[1].z is v1.parts[1].z and ('dense_part', id(self), 1, id(v0.parts[1].z)) in memo:
        y = memo[('dense_part', id(self), 1, id(v0.parts[1].z))]
        k0 = v0.parts[0].shape[-1]
        R = y.numel() // y.shape[-1]
        lead = lambda p: int(np.prod(p.shape[:-1]))
        if lead(v0.parts[0]) % R == 0 and lead(v1.parts[0]) % R == 0:
            z0, z1 = ops.dense_pair(v0.parts[0], v1.parts[0], self.kernel[:k0], self.bias, res=y)
            return _apply_activation(z0, self.activation), _apply_activation(z1, self.activation)
    if not isinstance(v0, LazyConcat) and not isinstance(v1, LazyConcat):
        z0, z1 = ops.dense_pair(v0, v1, self.kernel, self.bias)
        return _apply_activation(z0, self.activation), _apply_activation(z1, self.activation)
    return self.compute(vals0, training, memo), self.compute(vals1, training, memo)


Dense.compute_pair = _dense_compute_pair


def _apply_activation(z, activation):
    if activation in (None, 'linear'):
        return z
    if activation in ('sigmoid', 'tanh'):
        return ops.affine_act(z, None, None, activation)
    if isinstance(activation, tuple) and activation[0] == 'tanh_saturated':
        # nonlin_tanh_saturated (backend_tensorflow.py:78-81): coef * tanh(x)
        return ops.affine_act(z, None, None, 'tanh') * activation[1]
    return activation(z)


class BatchNormalization(Layer):
    """keras BatchNormalization(axis=-1): momentum .99, eps 1e-3.  Emits the per-channel affine as a Lazy."""
    def __init__(self, name=None):
        super(BatchNormalization, self).__init__(name)

    def build(self, in_shapes):
        C = in_shapes[0][-1]
        self.gamma = nn.Parameter(torch.ones(C))
        self.beta = nn.Parameter(torch.zeros(C))
        self.register_buffer('moving_mean', torch.zeros(C))
        self.register_buffer('moving_variance', torch.ones(C))
        # TF fuses BatchNorm for 4-D inputs; the fused kernel feeds the Bessel-corrected variance to the
        # moving average (3-D inputs take the unfused path with the biased variance).
        self.fused4d = len(in_shapes[0]) == 2

    def compute(self, vals, training, memo):
        z = to_tensor(vals[0])
        update = bool(training) and not (memo is not None and memo.get('freeze_bn_stats', False))
        z, scale, shift = ops.batchnorm_affine(z, self.gamma, self.beta, self.moving_mean, self.moving_variance,
                                               bool(training), update, self.fused4d)
        return Lazy(z, scale, shift, lrelu=False)


class LeakyReLU(Layer):
    def __init__(self, alpha=0.3, name=None):
        super(LeakyReLU, self).__init__(name)
        self.alpha = float(alpha)

    def compute(self, vals, training, memo):
        v = vals[0]
        if isinstance(v, Lazy) and not v.lrelu:
            return Lazy(v.z, v.scale, v.shift, lrelu=True, alpha=self.alpha)
        return Lazy(to_tensor(v), None, None, lrelu=True, alpha=self.alpha)

    def config(self):
        return {'alpha': self.alpha}


class Conv1D(Layer):
    """kl.Conv1D(filters, k, strides=1, padding='same', dilation_rate=1)."""
    def __init__(self, filters, kernel_size, use_bias=True, name=None):
        super(Conv1D, self).__init__(name)
        self.filters, self.kernel_size, self.use_bias = int(filters), int(kernel_size), bool(use_bias)

    def build(self, in_shapes):
        cin = in_shapes[0][-1]
        self.kernel = nn.Parameter(glorot_uniform((self.kernel_size, cin, self.filters)))
        self.bias = nn.Parameter(torch.zeros(self.filters)) if self.use_bias else None

    def out_shape(self, in_shapes):
        return (self.filters,)

    def compute(self, vals, training, memo):
        return ops.conv1d(to_tensor(vals[0]), self.kernel, self.bias)


class Conv2D(Layer):
    """kl.Conv2D(filters, [kt,kf], strides 1, padding 'same', channels_last); dilation/causality along time are
    build extensions (BASELINE config 5) that reduce to the reference at dil_t=1, causal=False."""
    def __init__(self, filters, kernel_size, use_bias=True, activation=None, dil_t=1, causal=False, name=None):
        super(Conv2D, self).__init__(name)
        self.filters, self.use_bias, self.activation = int(filters), bool(use_bias), activation
        self.kt, self.kf = int(kernel_size[0]), int(kernel_size[1])
        self.dil_t, self.causal = int(dil_t), bool(causal)
        self.bf16 = None          # 'out16' / 'out32': bf16 arithmetic with a bf16 / fp32 result map (ops.Conv2dFn); set by the network builder

    def build(self, in_shapes):
        cin = in_shapes[0][-1]
        self.kernel = nn.Parameter(glorot_uniform((self.kt, self.kf, cin, self.filters)))
        self.bias = nn.Parameter(torch.zeros(self.filters)) if self.use_bias else None

    def out_shape(self, in_shapes):
        return (in_shapes[0][0], self.filters)

    def compute(self, vals, training, memo):
        v = vals[0]
        if isinstance(v, LazyConcat):
            v = v.tensor()
        if getattr(self, 'bn_follows', False) and training and self.activation in (None, 'linear') and self.bf16 is None:
            # the BatchNormalization behind this layer takes its statistics from sums the convolution's launch leaves (ops._BNStats)
            ops._BNStats.want, ops._BNStats.last = True, None
            try:
                z = ops.conv2d(v, self.kernel, self.bias, self.dil_t, ops.PAD_CAUSAL if self.causal else ops.PAD_SAME, self.bf16)
            finally:
                ops._BNStats.want = False
            if ops._BNStats.last is not None:
                z._ptts_bn_partials, ops._BNStats.last = ops._BNStats.last, None
            return z
        z = ops.conv2d(v, self.kernel, self.bias, self.dil_t, ops.PAD_CAUSAL if self.causal else ops.PAD_SAME, self.bf16)
        return _apply_activation(z, self.activation)


def _conv2d_compute_pair(self, vals0, vals1, training, memo):
    v0, v1 = vals0[0], vals1[0]
    if isinstance(v0, LazyConcat) or isinstance(v1, LazyConcat) or self.activation not in (None, 'linear'):
        return self.compute(vals0, training, memo), self.compute(vals1, training, memo)
    return ops.conv2d_pair(v0, v1, self.kernel, self.bias, self.dil_t, ops.PAD_CAUSAL if self.causal else ops.PAD_SAME, self.bf16)


Conv2D.compute_pair = _conv2d_compute_pair


class Conv2DStack(Layer):
    """L x (kl.Conv2D(filters, [kt,kf], 'same', bias) -> kl.LeakyReLU(alpha)) of the critic (networks_critic.py:66-68) as ONE
    layer: the bf16-storage path of BASELINE configs[2] runs the whole stack per launch with the maps between the layers in
    the LDS (ops.conv2d_chain, csrc/conv2d_chain.hip).  Input [B,T,F] (the Reshape to one channel is implied), output
    [B,T,F,filters] with the last LeakyReLU applied.  The parameters are the L kernels and biases in Keras order
    (kernel_0, bias_0, kernel_1, ...), so weight lists, counts and checkpoints match the layer-by-layer graph."""
    def __init__(self, nlayers, filters, kernel_size, alpha=0.3, name=None):
        super(Conv2DStack, self).__init__(name)
        self.nlayers, self.filters, self.alpha = int(nlayers), int(filters), float(alpha)
        self.kt, self.kf = int(kernel_size[0]), int(kernel_size[1])

    def build(self, in_shapes):
        cin = 1
        for li in range(self.nlayers):
            setattr(self, 'kernel_{}'.format(li), nn.Parameter(glorot_uniform((self.kt, self.kf, cin, self.filters))))
            setattr(self, 'bias_{}'.format(li), nn.Parameter(torch.zeros(self.filters)))
            cin = self.filters

    def out_shape(self, in_shapes):
        return (in_shapes[0][0], self.filters)

    def kernels(self):
        return [getattr(self, 'kernel_{}'.format(li)) for li in range(self.nlayers)], [getattr(self, 'bias_{}'.format(li)) for li in range(self.nlayers)]

    def compute(self, vals, training, memo):
        x0 = to_tensor(vals[0])
        ws, bs = self.kernels()
        if x0.is_cuda and ops._C2C.supported(x0.shape[-1], ws):
            return ops.conv2d_chain(x0, ws, bs, self.alpha).float()
        # shapes without a chain kernel (F > 68): layer by layer on the one-plane matrix-core / stencil kernels
        v = x0.reshape(x0.shape[0], x0.shape[1], x0.shape[2], 1)
        for li, (w, b) in enumerate(zip(ws, bs)):
            z = ops.conv2d(v, w, b)
            v = Lazy(z, None, None, lrelu=True, alpha=self.alpha)
        return v

    def config(self):
        return {'nlayers': self.nlayers, 'filters': self.filters, 'kernel_size': [self.kt, self.kf], 'alpha': self.alpha}


class LSTM(Layer):
    """kl.LSTM(units, tanh, recurrent sigmoid, return_sequences=True), optionally kl.Bidirectional(concat).
    Weights are held combined: kernel [In, ndir*4H] = [fwd | bwd], recurrent_kernel [ndir,H,4H], bias [ndir*4H]."""
    def __init__(self, units, bidirectional=False, name=None):
        super(LSTM, self).__init__(name)
        self.units, self.ndir = int(units), 2 if bidirectional else 1

    def build(self, in_shapes):
        cin, H, nd = in_shapes[0][-1], self.units, self.ndir
        self.kernel = nn.Parameter(torch.cat([glorot_uniform((cin, 4 * H)) for _ in range(nd)], dim=1))
        self.recurrent_kernel = nn.Parameter(torch.stack([orthogonal((H, 4 * H)) for _ in range(nd)], dim=0))
        b = torch.zeros(nd, 4 * H)
        b[:, H:2 * H] = 1.0          # unit_forget_bias
        self.bias = nn.Parameter(b.reshape(-1))

    def out_shape(self, in_shapes):
        return (self.units * self.ndir,)

    def _input(self, vals):
        return to_tensor(vals[0] if not isinstance(vals[0], LazyConcat) else vals[0].tensor()).contiguous()

    def precompute(self, vals, training, memo):
        """The forward launches now, the autograd node later (`compute(..., pre=...)`): Model._run creates the node of a side-stream
        recurrence last, so that its backward chain -- the critical path of the generator step -- is the first thing the backward
        pass enqueues."""
        x = self._input(vals)
        with torch.no_grad():
            res = ops.lstm_launch(x.detach(), self.kernel.detach(), self.recurrent_kernel.detach(), self.bias.detach())
        # the input passes through an identity node created NOW, on this (side) stream: the late node's dx lands in that node's input
        # buffer -- a side-stream consumer -- and reaches the producer of x when the engine gets to this early, low-priority node.
        # Delivered directly, the engine would make the main stream wait for the whole backward chain at the moment the chain's node
        # finishes on the host, i.e. before the critic's backward is even enqueued.
        return {'x': ops.grad_gate(x) if x.requires_grad else x, 'res': res}

    def compute(self, vals, training, memo, pre=None):
        if pre is not None:
            return ops.lstm(pre['x'], self.kernel, self.recurrent_kernel, self.bias, pre=pre['res'])
        return ops.lstm(self._input(vals), self.kernel, self.recurrent_kernel, self.bias)


class GRU(Layer):
    """kl.GRU(reset_after=False) / Bidirectional: not on the WGAN hot path -> stock torch ops.
    Parameter layout follows Keras (kernel [In,3H], recurrent [H,3H], bias [3H]; gates z,r,h)."""
    def __init__(self, units, bidirectional=False, name=None):
        super(GRU, self).__init__(name)
        self.units, self.ndir = int(units), 2 if bidirectional else 1

    def build(self, in_shapes):
        cin, H, nd = in_shapes[0][-1], self.units, self.ndir
        self.kernel = nn.Parameter(torch.stack([glorot_uniform((cin, 3 * H)) for _ in range(nd)], dim=0))
        self.recurrent_kernel = nn.Parameter(torch.stack([orthogonal((H, 3 * H)) for _ in range(nd)], dim=0))
        self.bias = nn.Parameter(torch.zeros(nd, 3 * H))

    def out_shape(self, in_shapes):
        return (self.units * self.ndir,)

    def compute(self, vals, training, memo):
        x = to_tensor(vals[0])
        H = self.units
        outs = []
        for d in range(self.ndir):
            W, U, b = self.kernel[d], self.recurrent_kernel[d], self.bias[d]
            xp = x @ W + b
            h = x.new_zeros((x.shape[0], H))
            seq = [None] * x.shape[1]
            order = range(x.shape[1] - 1, -1, -1) if d == 1 else range(x.shape[1])
            for t in order:
                zr = torch.sigmoid(xp[:, t, :2 * H] + h @ U[:, :2 * H])
                z, r = zr[:, :H], zr[:, H:]
                hh = torch.tanh(xp[:, t, 2 * H:] + (r * h) @ U[:, 2 * H:])
                h = z * h + (1 - z) * hh
                seq[t] = h
            outs.append(torch.stack(seq, dim=1))
        return outs[0] if self.ndir == 1 else torch.cat(outs, dim=-1)


class Reshape(Layer):
    """kl.Reshape([-1] + tail): keeps (batch, time) and re-tiles the tail."""
    def __init__(self, tail, name=None):
        super(Reshape, self).__init__(name)
        self.tail = tuple(int(t) for t in tail)

    def out_shape(self, in_shapes):
        assert int(np.prod(in_shapes[0])) == int(np.prod(self.tail)), 'Reshape {} -> {}'.format(in_shapes[0], self.tail)
        return self.tail

    def compute(self, vals, training, memo):
        v = vals[0]
        if isinstance(v, LazyConcat):
            v = v.tensor()
        if isinstance(v, Lazy):
            z = v.z
            new = z.reshape(z.shape[0], z.shape[1], *self.tail)
            scale, shift = v.scale, v.shift
            if scale is not None and self.tail[-1] != z.shape[-1]:
                rep = self.tail[-1] // z.shape[-1]
                assert rep * z.shape[-1] == self.tail[-1], 'cannot carry a per-channel affine through this Reshape'
                scale, shift = scale.repeat(rep), shift.repeat(rep)
            return Lazy(new, scale, shift, v.lrelu, v.alpha)
        return v.reshape(v.shape[0], v.shape[1], *self.tail)


class SliceLast(Layer):
    """kl.Lambda(lambda x: x[:, :, a:b])"""
    def __init__(self, start, stop, name=None):
        super(SliceLast, self).__init__(name)
        self.start, self.stop = int(start), int(stop)

    def out_shape(self, in_shapes):
        return (self.stop - self.start,)

    def compute(self, vals, training, memo):
        return to_tensor(vals[0])[..., self.start:self.stop].contiguous()


class Concatenate(Layer):
    def out_shape(self, in_shapes):
        return tuple(in_shapes[0][:-1]) + (sum(s[-1] for s in in_shapes),)

    def compute(self, vals, training, memo):
        return LazyConcat(vals)


class Multiply(Layer):
    def compute(self, vals, training, memo):
        return to_tensor(vals[0]) * to_tensor(vals[1])


class GatedMultiply(Layer):
    """kl.Multiply()([conv_a(x), Conv2D(..., activation=sigmoid)(x)]) of the reference's gated convolution
    (networktts.py:128-134) with the sigmoid folded in: takes the two PRE-activations, one HIP pass (ptts_gated_mul_fwd)."""
    def compute(self, vals, training, memo):
        return ops.gated_mul(to_tensor(vals[0]), to_tensor(vals[1]))


class Activation(Layer):
    def __init__(self, activation, name=None):
        super(Activation, self).__init__(name)
        self.activation = activation

    def compute(self, vals, training, memo):
        return _apply_activation(to_tensor(vals[0]), self.activation)


class Dropout(Layer):
    """kl.Dropout(rate, noise_shape=(batch,1,None)) (networktts.py:65-70): one mask per sample and feature."""
    def __init__(self, rate, name=None):
        super(Dropout, self).__init__(name)
        self.rate = float(rate)

    def compute(self, vals, training, memo):
        x = to_tensor(vals[0])
        if not training or self.rate <= 0:
            return x
        keep = 1.0 - self.rate
        mask = (torch.rand(x.shape[0], 1, x.shape[2], device=x.device) < keep).to(x.dtype) / keep
        return x * mask


class GaussianNoiseInput(Layer):
    """networktts.py:36-56: concatenates `width` channels of N(0, stddev) noise."""
    def __init__(self, stddev=1.0, width=100, name=None):
        super(GaussianNoiseInput, self).__init__(name)
        self.stddev, self.width = float(stddev), int(width)

    def out_shape(self, in_shapes):
        return (in_shapes[0][-1] + self.width,)

    def compute(self, vals, training, memo):
        x = to_tensor(vals[0])
        noise = torch.randn(x.shape[0], x.shape[1], self.width, device=x.device) * self.stddev
        return torch.cat([x, noise], dim=-1)


# --------------------------------------------------------------------------------------------
# Model
# --------------------------------------------------------------------------------------------
_SIDE_STREAMS = {}

def side_streams(n, kind='side'):
    """The process-wide side streams (per device and kind), created once and shared by every model: each HIP stream
    created costs every later launch of the process (a second optimiser with streams of its own ran at half the speed of
    the first, tools/two_opt_probe.py), and a use never outlives the call that forked onto them."""
    key = (torch.cuda.current_device(), kind)
    ss = _SIDE_STREAMS.setdefault(key, [])
    while len(ss) < n:
        # high priority: a latency-bound chain of small kernels (the BLSTM branch) must not queue behind the wide ones
        ss.append(torch.cuda.Stream(priority=int(os.environ.get('PTTS_SIDE_PRIO', '-1'))))
    return ss[:n]


class Model(nn.Module):
    """keras.Model(inputs, outputs): evaluates the node graph; also the parameter container."""
    def __init__(self, inputs, outputs):
        super(Model, self).__init__()
        self.inputs = [inputs] if isinstance(inputs, Node) else list(inputs)
        self.single_output = isinstance(outputs, Node)
        self.outputs = [outputs] if self.single_output else list(outputs)
        order, seen = [], set()

        def visit(n):
            if id(n) in seen:
                return
            seen.add(id(n))
            for p in n.parents:
                visit(p)
            order.append(n)
        for o in self.outputs:
            visit(o)
        self.order = order
        # unique layers in graph order (an already-registered layer is shared, e.g. a frozen sub-network)
        lays, seen_l = [], set()
        for n in order:
            if id(n.layer) not in seen_l and not isinstance(n.layer, InputLayer):
                seen_l.add(id(n.layer))
                lays.append(n.layer)
        self.layers_list = nn.ModuleList(lays)
        # which inputs each node depends on
        self.deps = {}
        for n in order:
            if not n.parents:
                self.deps[id(n)] = {id(n)}
            else:
                s = set()
                for p in n.parents:
                    s |= self.deps[id(p)]
                self.deps[id(n)] = s

    def _run(self, feed, training, memo, values=None, only_dep=None, hold=None, cut_side=False):
        """Evaluate the nodes not yet in `values`.  `hold` (ids of nodes): these and everything that depends on them is
        left for a later call with the same `values` -- e.g. the generator's final concatenation, so that the critic can
        start on the spectral part while the side-stream branch still runs; the side-stream bookkeeping travels in
        values['__side__'].  Nodes tagged `stream = 1` (an independent branch such as the
        generator's latency-bound BLSTM) are enqueued on a side HIP stream when `self.parallel_branches` is set, so
        that they overlap with the rest of the graph; autograd replays the same streams in the backward pass (and,
        processing nodes in reverse creation order, enqueues the main-stream backward first).
        `cut_side`: a side branch gets its main-stream inputs as detached leaves (values['__cuts__'] = [(input tensor, leaf)]), so that
        the caller can run the branch's backward pass on its own, early, and hand the leaves' gradients to the rest of the graph later
        (optimizertts_wgan.generator_forward_early); values['__side_ev__'] marks the end of the branch's FORWARD on its stream -- the
        join then waits for that event, not for whatever else the caller has put on the stream since."""
        values = {} if values is None else values
        use_side = bool(getattr(self, 'parallel_branches', False)) and torch.cuda.is_available()
        side, cur, pending = None, None, False
        on_side = set()
        if '__side__' in values:
            on_side, pending, side, cur = values.pop('__side__')
        pre = values.pop('__pre__', {})       # node id -> launched-ahead results of a side-stream layer (Layer.precompute)
        late = bool(getattr(self, 'side_backward_first', False))
        held = set()
        order = self.order
        if use_side and os.environ.get('PTTS_SIDE_DEFER', '0') != '0':
            # (experiment, off: enqueue the main-stream nodes that do not need a side-stream result first.  Measured
            # 1.5 ms slower per generator step: the side branch's small kernels then queue behind the wide ones)
            side_nodes = set(id(n) for n in self.order if getattr(n, 'stream', 0))
            needs_side = set()
            for n in self.order:
                if id(n) in side_nodes or any(id(p) in needs_side or id(p) in side_nodes for p in n.parents):
                    needs_side.add(id(n))
            order = [n for n in self.order if id(n) not in needs_side] + [n for n in self.order if id(n) in needs_side]
        given = set(values.keys())                  # ready before this call (model inputs, shared values)
        start_ev = torch.cuda.current_stream().record_event() if use_side else None
        for n in order:
            if id(n) in values:
                continue
            if not n.parents:
                values[id(n)] = feed[id(n)]
                given.add(id(n))
                continue
            if hold and (id(n) in hold or any(id(p) in held for p in n.parents)):
                held.add(id(n))
                continue
            vals = [values[id(p)] for p in n.parents]
            if use_side and getattr(n, 'stream', 0):
                if side is None:
                    # NOT the first side stream: the branch may still be running when the caller evaluates ANOTHER model whose variants use
                    # it (the generator's forward hoisted in front of the critic step, optimizertts_wgan.device_step; the stacked critic
                    # pass needs one).  The second of the shared pool, not a stream of a kind of its own: one more HIP stream in the
                    # process (the fifth busy one) halved the speed of the step that runs the BLSTM in every critic step
                    # (bench.py's all_exact_work_reductions_off: 17.8 against 10.9 ms).
                    side = side_streams(2)[1]
                    cur = torch.cuda.current_stream()
                if id(n) in pre:
                    # launched in the earlier call; this one only creates the autograd node (no kernel reads the parents now)
                    with torch.cuda.stream(side):
                        values[id(n)] = n.layer.compute(vals, training, memo, pre=pre.pop(id(n)))
                    on_side.add(id(n))
                    pending = True
                    continue
                if not any(id(p) in on_side for p in n.parents):
                    # inputs come from the main stream: if they all existed before this call, wait only for the point
                    # where the call started, not for the main-stream work enqueued since
                    if all(id(p) in given for p in n.parents):
                        side.wait_event(start_ev)
                    else:
                        side.wait_stream(cur)
                if cut_side and torch.is_grad_enabled() and not any(id(p) in on_side for p in n.parents):
                    cvals = []
                    for p, v in zip(n.parents, vals):
                        t = v.tensor() if isinstance(v, LazyConcat) else to_tensor(v)
                        if torch.is_tensor(t) and t.requires_grad:
                            # the main graph goes on through an injection node (later consumers of this value take it from there):
                            # the gradient of the cut branch is added where the engine reaches that node, and only there does the
                            # main stream wait for the branch
                            holder = {}
                            t = ops.grad_inject(t, holder)
                            values[id(p)] = t
                            leaf = t.detach().requires_grad_(True)
                            values.setdefault('__cuts__', []).append((t, leaf, holder))
                            t = leaf
                        cvals.append(t)
                    vals = cvals
                if late and hold and not cut_side and torch.is_grad_enabled() and hasattr(n.layer, 'precompute'):
                    # a later call finishes the graph (`hold`): launch now, create the node then -- after everything the caller
                    # evaluates in between (the critic), i.e. with the highest priority of the backward pass
                    with torch.cuda.stream(side):
                        pre[id(n)] = n.layer.precompute(vals, training, memo)
                    held.add(id(n))
                    pending = True
                    continue
                with torch.cuda.stream(side):
                    values[id(n)] = n.layer.compute(vals, training, memo)
                on_side.add(id(n))
                pending = True
                continue
            if pending and any(id(p) in on_side for p in n.parents):
                ev = values.pop('__side_ev__', None)
                if ev is not None:
                    cur.wait_event(ev)
                else:
                    cur.wait_stream(side)
                pending = False
                for p in n.parents:
                    if id(p) in on_side:
                        t = values[id(p)]
                        t = t.z if isinstance(t, Lazy) else t
                        if torch.is_tensor(t):
                            t.record_stream(cur)
            values[id(n)] = n.layer.compute(vals, training, memo)
        if held:
            values['__side__'] = (on_side, pending, side, cur)      # the join happens in the call that finishes the graph
            values['__pre__'] = pre
            if cut_side and pending and side is not None:
                values['__side_ev__'] = side.record_event()
        elif pending:
            cur.wait_stream(side)
        return values

    def forward(self, *xs, **kw):
        training = kw.get('training', False)
        memo = kw.get('memo', None)
        if len(xs) == 1 and isinstance(xs[0], (list, tuple)):
            xs = tuple(xs[0])
        assert len(xs) == len(self.inputs), 'model expects {} inputs'.format(len(self.inputs))
        feed = {id(n): x for n, x in zip(self.inputs, xs)}
        values = self._run(feed, training, memo)
        outs = [to_tensor(values[id(o)]) for o in self.outputs]
        return outs[0] if self.single_output else outs

    def forward_multi(self, varying_index, variants, fixed, training=False, memo=None, parallel_streams=False):
        """Evaluate the model for several values of input `varying_index` while every node that does not
        depend on it (the critic's context branch) is computed once and shared."""
        memo = {} if memo is None else memo
        vin = self.inputs[varying_index]
        feed = {}
        fi = 0
        for i, n in enumerate(self.inputs):
            if i != varying_index:
                feed[id(n)] = fixed[fi]
                fi += 1
        shared = {}
        for n in self.order:
            if id(vin) in self.deps[id(n)]:
                continue
            if not n.parents:
                shared[id(n)] = feed[id(n)]
            else:
                shared[id(n)] = n.layer.compute([shared[id(p)] for p in n.parents], training, memo)
        # products of shared concat parts with their kernel rows (Dense over a LazyConcat) are computed once, up front
        for n in self.order:
            if isinstance(n.layer, Dense) and id(vin) in self.deps[id(n)] and isinstance(n.parents[0].layer, Concatenate):
                for i, pn in enumerate(n.parents[0].parents):
                    if i > 0 and id(pn) in shared:
                        n.layer.prime_part(i, [pp.shape[-1] for pp in n.parents[0].parents], shared[id(pn)], memo)
        results = []
        streams = self._variant_streams(len(variants)) if parallel_streams else None
        cur = torch.cuda.current_stream() if parallel_streams else None
        for vi, x in enumerate(variants):
            values = dict(shared)
            values[id(vin)] = x
            if streams is not None and vi > 0:
                st = streams[vi - 1]
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    values = self._run({}, training, memo, values)
                    outs = [to_tensor(values[id(o)]) for o in self.outputs]
            else:
                values = self._run({}, training, memo, values)
                outs = [to_tensor(values[id(o)]) for o in self.outputs]
            results.append(outs[0] if self.single_output else outs)
        if streams is not None:
            for vi in range(1, len(variants)):
                cur.wait_stream(streams[vi - 1])
                for o in (results[vi] if isinstance(results[vi], list) else [results[vi]]):
                    o.record_stream(cur)
        return results

    def forward_multi_at(self, node, variants, feed, training=False, memo=None, parallel_streams=False, shared_stream=False, pair=False):
        """forward_multi with the varying value at an INTERNAL node (`variants` are values of `node`, e.g. the critic's spectral slice:
        the stacked real / fake spectra and the interpolated sample, built by the optimiser without materialising the 86-column
        inputs they would be sliced from).  `feed` = {input node: tensor} for the inputs the rest of the graph needs; nodes that only
        `node` depends on (its own inputs) are not evaluated.  Everything that does not depend on `node` is computed once and shared."""
        memo = {} if memo is None else memo
        desc = set([id(node)])
        for n in self.order:
            if any(id(p) in desc for p in n.parents):
                desc.add(id(n))
        anc = set()
        def up(n):
            for p in n.parents:
                if id(p) not in anc:
                    anc.add(id(p)); up(p)
        up(node)
        feed = {id(k): v for k, v in feed.items()}
        shared = {}
        # shared_stream: the part of the graph that does not depend on `node` (the critic's context branch: Conv1D + two Dense layers and
        # its product with the first post-concat kernel) runs on a side stream BESIDE the variants' own part up to the first node that
        # needs it (the concatenation); autograd replays the fork in the backward pass.  One fork and one join per pass.
        side_sh = cur_sh = ev_sh = None
        join_nodes = None
        if shared_stream and torch.cuda.is_available() and not parallel_streams:
            join_nodes = set(id(n) for n in self.order if id(n) in desc and any(id(p) not in desc and id(p) not in anc for p in n.parents))
            if join_nodes:
                side_sh = side_streams(1, 'shared')[0]
                cur_sh = torch.cuda.current_stream()
                side_sh.wait_stream(cur_sh)
        import contextlib as _ctx
        with (torch.cuda.stream(side_sh) if side_sh is not None else _ctx.nullcontext()):
            for n in self.order:
                if id(n) in desc:
                    continue
                if not n.parents:
                    if id(n) in feed:
                        shared[id(n)] = feed[id(n)]
                    else:
                        assert id(n) in anc, 'forward_multi_at: input {} is needed but was not fed'.format(n.name)
                        shared[id(n)] = None
                elif id(n) in anc and any(shared.get(id(p)) is None for p in n.parents):
                    shared[id(n)] = None                  # feeds only `node`: its value is given
                else:
                    shared[id(n)] = n.layer.compute([shared[id(p)] for p in n.parents], training, memo)
            for n in self.order:
                if isinstance(n.layer, Dense) and id(n) in desc and isinstance(n.parents[0].layer, Concatenate):
                    for i, pn in enumerate(n.parents[0].parents):
                        if i > 0 and id(pn) in shared and shared[id(pn)] is not None:
                            n.layer.prime_part(i, [pp.shape[-1] for pp in n.parents[0].parents], shared[id(pn)], memo)
            if side_sh is not None:
                ev_sh = side_sh.record_event()
        if side_sh is not None:
            # the variants' own part first, all of them, then the join, then the rest
            results, partial = [], []
            for x in variants:
                values = dict(shared)
                values[id(node)] = x
                partial.append(self._run({}, training, memo, values, hold=join_nodes))
            cur_sh.wait_event(ev_sh)
            for v in shared.values():
                t = v.z if isinstance(v, Lazy) else v
                if torch.is_tensor(t):
                    t.record_stream(cur_sh)
            for k, t in memo.items():
                if isinstance(k, tuple) and k and k[0] == 'dense_part' and torch.is_tensor(t):
                    t.record_stream(cur_sh)
            for values in partial:
                values = self._run({}, training, memo, values)
                outs = [to_tensor(values[id(o)]) for o in self.outputs]
                results.append(outs[0] if self.single_output else outs)
            return results
        results = []
        if pair and len(variants) == 2 and not parallel_streams and ops._PairFlags.enabled and \
                not any(getattr(n, 'stream', 0) for n in self.order if id(n) in desc):
            # LOCKSTEP: the two evaluations walk the graph together, layer by layer; a layer that can (Conv2D, Dense: compute_pair) runs
            # both forwards as ONE launch when their inputs lie back to back in one buffer -- which its own two outputs then do for the
            # next layer.  The backward passes stay per evaluation (ops.Conv2dPairFn).
            vals = [dict(shared), dict(shared)]
            vals[0][id(node)], vals[1][id(node)] = variants[0], variants[1]
            for n in self.order:
                if id(n) not in desc or id(n) == id(node):
                    continue
                pv = [[vals[v][id(p)] for p in n.parents] for v in (0, 1)]
                if hasattr(n.layer, 'compute_pair'):
                    o0, o1 = n.layer.compute_pair(pv[0], pv[1], training, memo)
                else:
                    o0, o1 = n.layer.compute(pv[0], training, memo), n.layer.compute(pv[1], training, memo)
                vals[0][id(n)], vals[1][id(n)] = o0, o1
            for v in (0, 1):
                outs = [to_tensor(vals[v][id(o)]) for o in self.outputs]
                results.append(outs[0] if self.single_output else outs)
            return results
        streams = self._variant_streams(len(variants)) if parallel_streams else None
        cur = torch.cuda.current_stream() if parallel_streams else None
        for vi, x in enumerate(variants):
            values = dict(shared)
            values[id(node)] = x
            if streams is not None and vi > 0:
                st = streams[vi - 1]
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    values = self._run({}, training, memo, values)
                    outs = [to_tensor(values[id(o)]) for o in self.outputs]
            else:
                values = self._run({}, training, memo, values)
                outs = [to_tensor(values[id(o)]) for o in self.outputs]
            results.append(outs[0] if self.single_output else outs)
        if streams is not None:
            for vi in range(1, len(variants)):
                cur.wait_stream(streams[vi - 1])
                for o in (results[vi] if isinstance(results[vi], list) else [results[vi]]):
                    o.record_stream(cur)
        return results

    def _variant_streams(self, n):
        return side_streams(n - 1)

    # ---- Keras-like accessors ----------------------------------------------------------------
    def weights(self):
        out = []
        for li, lay in enumerate(self.layers_list):
            for k, t in lay.weights():
                out.append(('{}_{}/{}'.format(type(lay).__name__.lower(), li, k), t))
        return out

    def count_params(self):
        """keras Model.count_params(): trainable + non-trainable (BN moving statistics)."""
        return int(sum(t.numel() for _, t in self.weights()))

    def trainable_weights(self):
        return [p for p in self.parameters()]

    def summary(self, printfn=print):
        printfn('    {:<28}{:<18}{}'.format('layer', 'output tail', '#params'))
        seen = set()
        for n in self.order:
            if isinstance(n.layer, InputLayer) or id(n.layer) in seen:
                continue
            seen.add(id(n.layer))
            printfn('    {:<28}{:<18}{}'.format(type(n.layer).__name__, str(n.shape), sum(t.numel() for _, t in n.layer.weights())))
        printfn('    total params: {}'.format(self.count_params()))

    def to_json(self):
        import json
        desc = []
        for n in self.order:
            desc.append({'class': type(n.layer).__name__, 'shape': list(n.shape), 'name': n.name,
                         'parents': [self.order.index(p) for p in n.parents], 'config': n.layer.config()})
        return json.dumps({'format': 'percivaltts_amd.graph.v1', 'nodes': desc}, indent=1)

    def get_weights(self):
        return [t.detach().cpu().numpy().copy() for _, t in self.weights()]

    def set_weights(self, arrays):
        ws = self.weights()
        assert len(ws) == len(arrays), 'set_weights: {} arrays for {} weights'.format(len(arrays), len(ws))
        with torch.no_grad():
            for (k, t), a in zip(ws, arrays):
                a = torch.as_tensor(np.asarray(a), dtype=t.dtype)
                assert tuple(a.shape) == tuple(t.shape), 'set_weights: {} {} vs {}'.format(k, tuple(a.shape), tuple(t.shape))
                t.copy_(a.to(t.device))


class FlatParams(object):
    """All trainable weights of a model as views into ONE flat fp32 buffer, gradients likewise: one Adam launch and
    one all-reduce bucket per network."""
    def __init__(self, model, device):
        self.params = [p for p in model.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, dtype=torch.float32, device=device)
        self.grad = torch.zeros(n, dtype=torch.float32, device=device)
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat[off:off + k].copy_(p.detach().reshape(-1).to(device))
                p.data = self.flat[off:off + k].view(p.shape)
                p.grad = self.grad[off:off + k].view(p.shape)
                p._ptts_flat = self        # the buffer whose `epoch` counts in-place updates by raw kernels (Adam, clip)
                off += k
        self.numel = n
        self.epoch = 0

    def zero_grad(self):
        self.grad.zero_()

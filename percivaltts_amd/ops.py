"""torch.autograd.Functions over the HIP kernels of libpercival_hip.so.

Every numeric op of the WGAN-GP hot path goes through the C ABI (include/percival_hip.h); torch
supplies device memory, the autograd tape and streams only.  There is no CPU path.

The layers of the reference are Linear -> (BatchNorm) -> LeakyReLU (networktts.py:59-63,116-126,
networks_critic.py:66-68).  Here a layer's output is kept as its PRE-activation `z`; the pending
`a = lrelu(scale*z + shift)` travels as a `Lazy` and is applied by the next linear op while it
loads its input, so activations are never written to HBM.

Second-order support (the gradient penalty differentiates a gradient,
optimizertts_wgan.py:53-68): LeakyReLU is piecewise linear, so the backward of a
(lrelu -> linear) layer is linear in the incoming gradient with the same masks; Conv2dBwdData /
DenseBwdData are therefore Functions of their own whose backward is one masked forward sweep and
one weight-gradient sweep.
"""
import contextlib
import ctypes
import os

import torch

from . import _hip
from ._hip import call, ptr, stream, f32c

IN_NONE, IN_LRELU, IN_MASKMUL = 0, 1, 2
PAD_SAME, PAD_CAUSAL = 0, 1
ACT_NONE, ACT_LRELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3
ACT_CODES = {None: ACT_NONE, 'linear': ACT_NONE, 'lrelu': ACT_LRELU, 'sigmoid': ACT_SIGMOID, 'tanh': ACT_TANH}


class Lazy(object):
    """A pending activation: a = act(scale*z + shift), act in {none, lrelu(alpha)}; scale/shift per channel or None."""
    __slots__ = ('z', 'scale', 'shift', 'lrelu', 'alpha')

    def __init__(self, z, scale=None, shift=None, lrelu=False, alpha=0.3):
        self.z, self.scale, self.shift, self.lrelu, self.alpha = z, scale, shift, lrelu, alpha

    @property
    def shape(self):
        return self.z.shape

    def tensor(self):
        """Materialise (only where no consumer can fuse the transform)."""
        if self.scale is None and not self.lrelu:
            return self.z
        return affine_act(self.z, self.scale, self.shift, 'lrelu' if self.lrelu else None, self.alpha)


def as_lazy(v):
    return v if isinstance(v, Lazy) else Lazy(v)


def as_tensor(v):
    return v.tensor() if isinstance(v, Lazy) else v


def _fusable(lz):
    """(mode, scale, shift) a linear kernel can apply on load, or None if it must be materialised first."""
    if lz.lrelu:
        return IN_LRELU, lz.scale, lz.shift
    if lz.scale is None:
        return IN_NONE, None, None
    return None   # affine without LeakyReLU: not a fused mode


def _prep(v):
    lz = as_lazy(v)
    f = _fusable(lz)
    if f is None:
        return lz.tensor(), IN_NONE, None, None, lz.alpha
    return lz.z, f[0], f[1], f[2], lz.alpha


# ----------------------------------------------------------------------------------------------
# g-pass switch: while the gradient penalty takes d(sum v)/d(x_hat) with create_graph=True the
# parameter gradients of that pass are not wanted (they are not loss gradients); skipping them
# halves the work of the pass.
# ----------------------------------------------------------------------------------------------
class _Flags(object):
    skip_param_grads = False
    deterministic = os.environ.get('PTTS_DETERMINISTIC', '0') == '1'
    bf16_products = False


def bf16_products(on=None):
    """bf16 products (BASELINE configs[2]; build extension, the reference is fp32): the split GEMM kernels -- context Conv1D
    forward and weight gradient, Dense forward / backward-data / weight gradient, LSTM input projections -- form ONE product of
    the operands' bf16 roundings (plane 1 of the split they already hold) instead of six; fp32 accumulation, fp32 master
    weights, gradients and activations in HBM.  Products that do not take a split kernel (the 1-wide heads, small shapes,
    the recurrent LSTM products) stay fp32.  Returns the setting; `on=None` only reads it."""
    if on is not None:
        _Flags.bf16_products = bool(on)
    _hip.lib().ptts_set_bf16_products(1 if _Flags.bf16_products else 0)
    return _Flags.bf16_products


def deterministic(on=None):
    """Deterministic mode (PTTS_DETERMINISTIC=1): every reduction over workgroups runs in a fixed order -- no fp32 atomics.
    The stream-K GEMM tiles, the grouped / frame-major / bf16x6 weight-gradient kernels, the split context Conv1D and the
    grouped conv2d reduction (all of which combine partial sums with atomics) give way to their single-pass forms; two runs
    of the same step then agree bit for bit, at a price in speed (tests/test_fullsize_gpu.py, the resume-parity test).
    Returns the setting; `on=None` only reads it."""
    if on is not None:
        _Flags.deterministic = bool(on)
    _hip.lib().ptts_set_deterministic(1 if _Flags.deterministic else 0)
    return _Flags.deterministic


@contextlib.contextmanager
def input_grad_only():
    old = _Flags.skip_param_grads
    _Flags.skip_param_grads = True
    try:
        yield
    finally:
        _Flags.skip_param_grads = old


# ----------------------------------------------------------------------------------------------
# Deferred weight gradients: inside `deferred_weight_grads()` the Dense layers do not launch their
# dW = a^T.dy product (4 tiles of 128x128 over K = 25 600: launch, zero-fill and a 128-way split for
# 3.4 GFLOP) one by one; they queue it, and flush_weight_grads() runs every queued product of the
# backward pass in ONE grouped launch that accumulates straight into the parameters' .grad buffers
# (ptts_gemm_wgrad_grouped).  Only parameters whose .grad exists beforehand (FlatParams views) take
# part; everything else keeps the immediate path.
# ----------------------------------------------------------------------------------------------
class _Deferred(object):
    active = False
    items = []          # (x2, dy2, mask, scale, shift, gw, gb, M_out, N, K_rows, mode, alpha)
    conv_items = []     # (partials buffer, byte offset of the rows, nblocks, npart, nw, cout, gw, gb): conv2d backward passes awaiting their reduction
    streams = []


@contextlib.contextmanager
def deferred_weight_grads():
    old = _Deferred.active
    _Deferred.active = True
    try:
        yield
        flush_weight_grads()
    finally:
        _Deferred.active = old
        _Deferred.items = []
        _Deferred.conv_items = []
        _Deferred.streams = []


def deferred_detach():
    """Take the queued weight-gradient products and the list of streams to join out of the current deferred_weight_grads() context
    (they are handed to a later context with deferred_attach): the early backward pass of a side branch must not make the main stream
    wait for that branch when its context closes."""
    st = (_Deferred.items, _Deferred.conv_items, _Deferred.streams)
    _Deferred.items, _Deferred.conv_items, _Deferred.streams = [], [], []
    return st


def deferred_attach(st):
    if st is None:
        return
    items, conv_items, streams = st
    _Deferred.items = list(items) + _Deferred.items
    _Deferred.conv_items = list(conv_items) + _Deferred.conv_items
    for q in streams:
        if all(q.cuda_stream != r.cuda_stream for r in _Deferred.streams):
            _Deferred.streams.append(q)


def grad_target(p):
    """The persistent gradient buffer behind parameter `p` (a leaf with .grad allocated, or a contiguous row slice of
    one), as a tensor view with p's shape -- or None."""
    if p is None or not p.requires_grad:
        return None
    if p.is_leaf:
        g = p.grad
        return g if (g is not None and g.is_contiguous() and g.shape == p.shape) else None
    base = p._base
    if base is None or not base.is_leaf or base.grad is None or not p.is_contiguous() or not base.grad.is_contiguous():
        return None
    if base.stride() != base.grad.stride():
        return None
    off = p.storage_offset() - base.storage_offset()
    return base.grad.view(-1)[off:off + p.numel()].view(p.shape)


_WG_FLUSH_AT = int(__import__('os').environ.get('PTTS_WG_FLUSH', '3'))     # > 0: launch a stream's queue once it holds this many


def _defer_wgrad(x2, dy2, gw, gb, K_in, N, M_rows, mode, scale, shift, mask_src, alpha):
    cur = torch.cuda.current_stream()
    if all(cur.cuda_stream != s.cuda_stream for s in _Deferred.streams):
        _Deferred.streams.append(cur)
    _Deferred.items.append((cur.cuda_stream, x2, dy2, mask_src, scale, shift, gw, gb, K_in, N, M_rows, mode, alpha))
    if _WG_FLUSH_AT > 0:
        mine = [it for it in _Deferred.items if it[0] == cur.cuda_stream]
        if len(mine) >= _WG_FLUSH_AT:
            _Deferred.items = [it for it in _Deferred.items if it[0] != cur.cuda_stream]
            _launch_wgrads(mine)


def _wgrad_split_ok(x2, dy2, mask_src, scale, shift, K_in, N, M_rows):
    """The two-stage split weight-gradient kernels (csrc/dense.hip: partial tiles per workgroup, one grouped fixed-order reduce)
    take this product: every Dense / LSTM weight gradient over >= 2048 frames with N >= wgrad_min_n (16) outside deterministic
    mode -- 31 us + its share of the reduce at 256 x 256 x 25 600 against 60 us a product for the grouped fp32 launch, 188 us
    against 292 at N = 2048 (the LSTM projections)."""
    if not (_DenseSplit.enabled and not _Flags.deterministic and M_rows >= 2048 and N >= _DenseSplit.wgrad_min_n and K_in >= 16):
        return False
    if not _hip.lib().ptts_dense_wgrad_bf16x6_supported(K_in, N, M_rows, x2.stride(0), dy2.stride(0)):
        return False
    return all(t is None or t.data_ptr() % 16 == 0 for t in (x2, dy2, mask_src, scale, shift))


def _launch_wgrads(items):
    cur = torch.cuda.current_stream()
    rest, split_items = [], []
    for it in items:
        (_, x2, dy2, mask_src, scale, shift, gw, gb, K_in, N, M_rows, mode, alpha) = it
        if _wgrad_split_ok(x2, dy2, mask_src, scale, shift, K_in, N, M_rows):
            for t in (x2, dy2, mask_src):
                if t is not None:
                    t.record_stream(cur)
            # stage 1: the workgroups' partial tiles; stage 2 below, one launch for all products of this batch
            ws = torch.empty(_hip.lib().ptts_dense_wgrad_workspace_bytes(K_in, N, M_rows), dtype=torch.uint8, device=x2.device)
            nsplit = ctypes.c_int(0)
            call('ptts_dense_wgrad_bf16x6_partials', ptr(x2), ptr(dy2), ptr(mask_src), ptr(scale), ptr(shift), ptr(ws), ws.numel(),
                 ctypes.byref(nsplit), K_in, N, M_rows, x2.stride(0), dy2.stride(0), mode, alpha, stream(), tag=(K_in, N, M_rows))
            split_items.append((ws, nsplit.value, K_in, N, gw, gb))
        else:
            rest.append(it)
    if split_items:
        descs = (_hip.DenseWgradReduceDesc * len(split_items))()
        for d, (ws, nsplit, K_in, N, gw, gb) in zip(descs, split_items):
            d.partials, d.split, d.Kin, d.N, d.ldc = ws.data_ptr(), nsplit, K_in, N, gw.stride(0)
            d.C, d.colsum_b = gw.data_ptr(), (gb.data_ptr() if gb is not None else None)
        call('ptts_dense_wgrad_reduce_grouped', ctypes.cast(descs, ctypes.c_void_p), len(split_items), stream(), tag=(len(split_items),))
    items = rest
    if not items:
        return
    descs = (_hip.WGradDesc * len(items))()
    for d, (_, x2, dy2, mask_src, scale, shift, gw, gb, K_in, N, M_rows, mode, alpha) in zip(descs, items):
        for t in (x2, dy2, mask_src):
            if t is not None:
                t.record_stream(cur)
        d.A, d.B, d.C, d.colsum_b = x2.data_ptr(), dy2.data_ptr(), gw.data_ptr(), (gb.data_ptr() if gb is not None else None)
        d.in_scale = scale.data_ptr() if scale is not None else None
        d.in_shift = shift.data_ptr() if shift is not None else None
        d.mask_src = mask_src.data_ptr() if mask_src is not None else None
        d.M, d.N, d.K = K_in, N, M_rows
        d.lda, d.ldb, d.ldc = x2.stride(0), dy2.stride(0), gw.stride(0)
        d.in_mode, d.alpha = mode, alpha
    call('ptts_gemm_wgrad_grouped', ctypes.cast(descs, ctypes.c_void_p), len(items), stream(), tag=(len(items),))


def flush_weight_grads():
    items, conv_items = _Deferred.items, _Deferred.conv_items
    if not items and not conv_items:
        return
    cur = torch.cuda.current_stream()
    for s in _Deferred.streams:           # operands produced on the side streams of the backward pass
        if s.cuda_stream != cur.cuda_stream:
            cur.wait_stream(s)
    if items:
        _launch_wgrads(items)
    if conv_items:
        descs = (_hip.Conv2dReduceDesc * len(conv_items))()
        for d, (buf, off, nblocks, npart, nw, cout, gw, gb) in zip(descs, conv_items):
            buf.record_stream(cur)
            d.partials = buf.data_ptr() + off
            d.nblocks, d.npart, d.nw, d.cout = nblocks, npart, nw, cout
            d.dw = gw.data_ptr() if gw is not None else None
            d.dbias = gb.data_ptr() if gb is not None else None
        call('ptts_conv2d_reduce_grouped', ctypes.cast(descs, ctypes.c_void_p), len(conv_items), stream(), tag=(len(conv_items),))
    _Deferred.items = []
    _Deferred.conv_items = []
    _Deferred.streams = []


_ws_cache = {}
_counter_cache = {}


def _stream_counter(device):
    """A zeroed int32 per device and stream for kernels that elect their finishing workgroup (they leave it zero)."""
    key = (device.index, _hip.stream_id())
    c = _counter_cache.get(key)
    if c is None:
        c = _counter_cache[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return c


def _workspace(nbytes, device):
    """One growing scratch buffer per device and stream (kernels on one stream are ordered)."""
    key = (device.index, _hip.stream_id(), torch.cuda.is_current_stream_capturing())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# ----------------------------------------------------------------------------------------------
# raw kernel wrappers (no autograd)
# ----------------------------------------------------------------------------------------------
class _C2M(object):
    """The 4 -> 4 channel 5x5 Conv2D layers on the bf16 matrix cores (csrc/conv2d_mfma.hip): fp32 arithmetic by the
    three-way bf16 split of both operands (six products, fp32 accumulation), like the context Conv1D's split kernels.
    PTTS_CONV2D_MFMA=0 or conv2d_mfma(False) select the packed-FMA stencil of csrc/conv2d.hip.  The Toeplitz tables of a
    kernel (forward and transposed, 5.3 KB each) are rebuilt when the kernel changes (tensor version / flat-buffer epoch)."""
    default = os.environ.get('PTTS_CONV2D_MFMA', '1') == '1'
    enabled = default
    tables = {}         # (id(w), stream) -> (w, version, epoch, fwd table, bwd table)

    @staticmethod
    def eligible(x, w, dil_t):
        KT, KF, Cin, Cout = w.shape
        return x.is_cuda and Cin == 4 and Cout == 4 and KT == 5 and KF == 5 and dil_t in (1, 2, 4, 8)

    @classmethod
    def table(cls, w, transposed, planes=3):
        flat = getattr(w, '_ptts_flat', None)
        epoch = None if flat is None else flat.epoch
        sid = _hip.stream_id()
        key = (id(w), sid, planes)         # one copy per stream: the build is ordered with its consumers by the stream itself
        ent = cls.tables.get(key)
        if ent is None or ent[0] is not w or ent[1] != w._version or ent[2] != epoch or flat is None:
            nb = _hip.lib().ptts_conv2d_mfma_table_bytes(5)
            reuse = ent is not None and ent[0] is w
            tf = ent[3] if reuse else torch.empty(nb, dtype=torch.uint8, device=w.device)
            tb = ent[4] if reuse else torch.empty(nb, dtype=torch.uint8, device=w.device)
            if len(cls.tables) > 512:
                cls.tables = {}
            cls.tables[key] = ent = (w, w._version, epoch, tf, tb)
            # with it the other kernels of the same flat buffer whose tables on this stream are out of date (the 7 layers of a stack
            # after an update): one launch
            todo = [ent]
            if flat is not None:
                for k2, e2 in cls.tables.items():
                    if k2 is not key and k2[1] == sid and k2[2] == planes and getattr(e2[0], '_ptts_flat', None) is flat and \
                            (e2[1] != e2[0]._version or e2[2] != epoch):
                        cls.tables[k2] = e2 = (e2[0], e2[0]._version, epoch, e2[3], e2[4])
                        todo.append(e2)
            if len(todo) == 1:
                call('ptts_conv2d_mfma_tables', ptr(w), ptr(tf), ptr(tb), 5, 5, 4, 4, planes, stream(), tag=(5, 5, planes))
            else:
                n = len(todo)
                pw = (ctypes.c_void_p * n)(*[e[0].data_ptr() for e in todo])
                pf = (ctypes.c_void_p * n)(*[e[3].data_ptr() for e in todo])
                pb = (ctypes.c_void_p * n)(*[e[4].data_ptr() for e in todo])
                call('ptts_conv2d_mfma_tables_grouped', ctypes.cast(pw, ctypes.c_void_p), ctypes.cast(pf, ctypes.c_void_p),
                     ctypes.cast(pb, ctypes.c_void_p), n, planes, stream(), tag=(n, planes))
        return ent[4] if transposed else ent[3]

    @staticmethod
    def pad_t(dil_t, pad_mode):
        return 4 * dil_t if pad_mode == PAD_CAUSAL else 2 * dil_t

    @classmethod
    def clear(cls):
        """Mark every table out of date (keys and buffers stay: the next use rebuilds them, grouped)."""
        cls.tables = {k: (e[0], None, -1, e[3], e[4]) for k, e in cls.tables.items()}


def conv2d_mfma(on):
    """Switch the matrix-core Conv2D kernels on or off (None: the default)."""
    _C2M.enabled = _C2M.default if on is None else bool(on)
    _C2M.clear()


def conv2d_path_description():
    return ('4->4 5x5 layers: bf16x6 split on the bf16 matrix cores (fp32 accumulate); 1->4 / 4->1 layers: fp32 packed-FMA stencil'
            if _C2M.enabled else 'fp32 packed-FMA stencil')


def _is16(t):
    return t is not None and t.dtype == torch.bfloat16


class _BNStats(object):
    """A Conv2D / Dense layer that feeds a BatchNormalization layer asks its forward launch for the per-channel sums of what it stores
    (csrc/conv2d_mfma.hip, fwd_ws_kernel<.., STATS>; csrc/dense.hip, DenseArgs.stats): `want` is set around the layer's ops.conv2d /
    ops.dense call, `last` = (partial rows [n, 2 C] float64, number of rows written, values per channel) is what the launch left -- or
    None when the shape took another kernel.  The layer hangs it on its output (`_ptts_bn_partials`), BatchNormTrainFn finishes it with
    ptts_bn_finalize_partials instead of a statistics pass over the tensor (conv maps: ptts_bn_batch_stats, 16 us a layer and forward
    pass, 7 such layers in the generator; Dense outputs: ptts_colstats + ptts_bn_finalize, 22 us).  PTTS_CONV_BN_STATS=0: off."""
    enabled = os.environ.get('PTTS_CONV_BN_STATS', '1') == '1'
    want = False
    last = None


def conv_bn_stats(on):
    _BNStats.enabled = (os.environ.get('PTTS_CONV_BN_STATS', '1') == '1') if on is None else bool(on)


def _conv2d_mfma_fwd(x, w, table, b, scale, shift, mask_src, out_mask, mode, alpha, dil_t, pad_t, planes=3, out_bf16=False):
    B, T, F, _ = x.shape
    assert planes in (1, 3) and (planes == 1 or not (_is16(x) or out_bf16)), 'bf16 tensors need the one-plane (bf16 arithmetic) kernels'
    assert mask_src is None or mask_src.dtype == x.dtype, 'conv2d: the mask source has the input\'s storage type'
    y = torch.empty((B, T, F, 4), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    assert out_mask is None or out_mask.dtype == y.dtype, 'conv2d: the output mask has the output\'s storage type'
    if _BNStats.want and _BNStats.enabled and planes == 3 and mask_src is None and out_mask is None and \
            _hip.lib().ptts_conv2d_mfma_fwd_stats_supported(F, dil_t, mode):
        part = torch.empty((256, 8), dtype=torch.float64, device=x.device)
        nrows = ctypes.c_int(0)
        call('ptts_conv2d_mfma_fwd_stats', ptr(x), ptr(table), ptr(b), ptr(scale), ptr(shift), ptr(y), B, T, F, 5, pad_t, mode, alpha,
             ptr(part), 256, ctypes.byref(nrows), stream(), tag=(B, T, F, 4, 4, mode, 'stats'))
        _BNStats.last = (part, nrows.value, B * T * F)
        return y
    call('ptts_conv2d_mfma_fwd', ptr(x), ptr(table), ptr(b), ptr(scale), ptr(shift), ptr(mask_src), ptr(out_mask), ptr(y),
         B, T, F, 5, dil_t, pad_t, mode, alpha, planes, int(_is16(x)), int(out_bf16), stream(),
         tag=(B, T, F, 4, 4, mode, int(out_mask is not None), planes))
    return y


def _conv2d_mfma_wgrad(dy, x, mask_src, mode, alpha, dil_t, pad_t, planes=3):
    """Partial sums of dW / dbias: (buffer, nblocks, npart); the rows start 4096 bytes into the buffer."""
    B, T, F, _ = x.shape
    assert planes == 1 or not (_is16(x) or _is16(dy))
    assert mask_src is None or mask_src.dtype == x.dtype
    nws = _hip.lib().ptts_conv2d_mfma_wgrad_workspace_bytes(B, T)
    buf = torch.empty(int(nws), dtype=torch.uint8, device=x.device)
    nblocks, npart = ctypes.c_int(0), ctypes.c_int(0)
    call('ptts_conv2d_mfma_wgrad_partials', ptr(dy), ptr(x), ptr(mask_src), ptr(buf), buf.numel(), ctypes.byref(nblocks),
         ctypes.byref(npart), B, T, F, 5, dil_t, pad_t, mode, alpha, planes, int(_is16(x)), int(_is16(dy)), stream(),
         tag=(B, T, F, 4, 4, mode, planes))
    return buf, nblocks.value, npart.value


class _C2MFused(object):
    """The fused backward launches of csrc/conv2d_mfma.hip (c2m::bwd_ws_kernel; dilation 1, fp32 maps): dx + dW of a layer, and the
    second-order sweep's masked forward + dW, as ONE launch each -- one staging of the shared operand.  PTTS_C2M_FUSED=0 /
    conv2d_fused(False) select the separate launches (A/B, tests)."""
    default = os.environ.get('PTTS_C2M_FUSED', '1') == '1'
    enabled = default

    @staticmethod
    def ok(x, dil_t, pad_mode, planes):
        return _C2MFused.enabled and planes == 3 and dil_t == 1 and pad_mode == PAD_SAME and not _is16(x)


def conv2d_fused(on):
    _C2MFused.enabled = _C2MFused.default if on is None else bool(on)


def _conv2d_mfma_bwd_fused(kind, p, q, mask_src, w, alpha):
    """kind 1: (dy, x) -> dx, partial rows of dW / dbias; kind 2: (u, dy, mask_src = x) -> cot_dy, partial rows of dW.
    Returns (y, buffer, nblocks, npart); the rows start 4096 bytes into the buffer."""
    B, T, F, _ = p.shape
    f32c(p, 'conv2d_bwd_fused.p'); f32c(q, 'conv2d_bwd_fused.q'); f32c(mask_src)
    assert q.shape == p.shape and (mask_src is None or mask_src.shape == p.shape)
    y = torch.empty_like(p)
    nws = _hip.lib().ptts_conv2d_mfma_bwd_fused_workspace_bytes(B, T)
    buf = torch.empty(int(nws), dtype=torch.uint8, device=p.device)
    nblocks, npart = ctypes.c_int(0), ctypes.c_int(0)
    call('ptts_conv2d_mfma_bwd_fused', ptr(p), ptr(q), ptr(mask_src), ptr(_C2M.table(w, kind == 1)), ptr(y), ptr(buf), buf.numel(),
         ctypes.byref(nblocks), ctypes.byref(npart), B, T, F, 5, 2, kind, alpha, stream(), tag=(B, T, F, kind))
    return y, buf, nblocks.value, npart.value


def _conv2d_mfma_bwd_fused_affine(dy, x, w, alpha, scale, shift):
    """Kind 1 for a layer input lrelu(scale x + shift): (dy, x) -> dx (w.r.t. the raw x), partial rows of dW / dbias / dscale / dshift."""
    B, T, F, _ = dy.shape
    f32c(dy, 'conv2d_bwd_fused.dy'); f32c(x, 'conv2d_bwd_fused.x'); f32c(scale); f32c(shift)
    assert x.shape == dy.shape
    y = torch.empty_like(dy)
    nws = _hip.lib().ptts_conv2d_mfma_bwd_fused_workspace_bytes(B, T)
    buf = torch.empty(int(nws), dtype=torch.uint8, device=dy.device)
    nblocks, npart = ctypes.c_int(0), ctypes.c_int(0)
    call('ptts_conv2d_mfma_bwd_fused_affine', ptr(dy), ptr(x), ptr(_C2M.table(w, True)), ptr(y), ptr(buf), buf.numel(),
         ctypes.byref(nblocks), ctypes.byref(npart), B, T, F, 5, 2, alpha, ptr(scale), ptr(shift), stream(), tag=(B, T, F, 'affine'))
    return y, buf, nblocks.value, npart.value


def _st(t, name='tensor'):
    """Validate a conv2d map of the bf16-storage path: contiguous device tensor, fp32 or bf16."""
    if t is None:
        return None
    if not (t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and t.is_contiguous()):
        raise _hip.HipLibraryError('{}: expected a contiguous float32 / bfloat16 device tensor, got {} {} contiguous={}'.format(
            name, t.device, t.dtype, t.is_contiguous()))
    return t


def _conv2d_fwd_raw(x, w, b, scale, shift, mask_src, mode, alpha, dil_t, pad_mode, planes=3, out_bf16=False):
    if planes == 1:
        # bf16 arithmetic / storage (BASELINE configs[2]): matrix-core kernels only
        _st(x, 'conv2d.x'); _st(mask_src, 'conv2d.mask_src'); f32c(w, 'conv2d.w'); f32c(b, 'conv2d.b'); f32c(scale); f32c(shift)
        if not (_C2M.eligible(x, w, dil_t) and dil_t == 1 and 0.0 <= alpha <= 1.0):
            raise _hip.HipLibraryError('conv2d: bf16 storage is built for the 4 -> 4 channel 5x5 layers at dilation 1')
        return _conv2d_mfma_fwd(x, w, _C2M.table(w, False, 1), b, scale, shift, mask_src, None, mode, alpha, dil_t,
                                _C2M.pad_t(dil_t, pad_mode), 1, out_bf16)
    f32c(x, 'conv2d.x'); f32c(w, 'conv2d.w'); f32c(b, 'conv2d.b'); f32c(scale); f32c(shift); f32c(mask_src)
    B, T, F, Cin = x.shape
    KT, KF, Ci2, Cout = w.shape
    assert Ci2 == Cin, 'conv2d: kernel Cin {} != input {}'.format(Ci2, Cin)
    assert b is None or b.numel() == Cout
    assert scale is None or (scale.numel() == Cin and shift.numel() == Cin)
    assert mask_src is None or mask_src.shape == x.shape
    if _C2M.enabled and _C2M.eligible(x, w, dil_t) and 0.0 <= alpha <= 1.0:
        return _conv2d_mfma_fwd(x, w, _C2M.table(w, False), b, scale, shift, mask_src, None, mode, alpha, dil_t,
                                _C2M.pad_t(dil_t, pad_mode))
    y = torch.empty((B, T, F, Cout), dtype=torch.float32, device=x.device)
    call('ptts_conv2d_fwd', ptr(x), ptr(w), ptr(b), ptr(scale), ptr(shift), ptr(mask_src), ptr(y),
         B, T, F, Cin, Cout, KT, KF, dil_t, pad_mode, mode, alpha, stream(), tag=(B, T, F, Cin, Cout, mode))
    return y


def _conv2d_bwd_raw(dy, x, w, scale, shift, mask_src, mode, alpha, dil_t, pad_mode,
                    want_dx, want_dw, want_db, want_affine, planes=3):
    if planes == 1:
        _st(dy, 'conv2d_bwd.dy'); _st(x, 'conv2d_bwd.x'); _st(mask_src); f32c(w)
        assert scale is None and not want_affine, 'conv2d: no BatchNorm-fused input on the bf16-storage path'
    else:
        f32c(dy, 'conv2d_bwd.dy'); f32c(x, 'conv2d_bwd.x'); f32c(w); f32c(scale); f32c(shift); f32c(mask_src)
    B, T, F, Cin = x.shape
    KT, KF, _, Cout = w.shape
    assert dy.shape == (B, T, F, Cout), 'conv2d_bwd: dy shape {} vs {}'.format(tuple(dy.shape), (B, T, F, Cout))
    assert mask_src is None or mask_src.shape == x.shape
    dev = x.device
    if (planes == 1 or _C2M.enabled) and _C2M.eligible(x, w, dil_t) and scale is None and not want_affine and 0.0 <= alpha <= 1.0:
        # matrix-core kernels: backward data = forward through the transposed table with the layer input's LeakyReLU mask
        # in the store; weight gradient = per-workgroup partial sums + the grouped reduction.  The gradient of a map is
        # stored like the map (bf16 storage: dx has x's type)
        pad_t = _C2M.pad_t(dil_t, pad_mode)
        dx = dw = db = None
        fused = want_dx and (want_dw or want_db) and mode == IN_LRELU and _C2MFused.ok(x, dil_t, pad_mode, planes)
        if fused:
            dx, buf, nblocks, npart = _conv2d_mfma_bwd_fused(1, dy, x, None, w, alpha)
        elif want_dx:
            assert mode != IN_MASKMUL, 'conv2d_bwd: dx is not defined for MASKMUL (weight-only sweep)'
            dx = _conv2d_mfma_fwd(dy, w, _C2M.table(w, True, planes), None, None, None, None, x if mode == IN_LRELU else None,
                                  IN_NONE, alpha, dil_t, 4 * dil_t - pad_t, planes, _is16(x))
        if want_dw or want_db:
            if not fused:
                buf, nblocks, npart = _conv2d_mfma_wgrad(dy, x, mask_src, mode, alpha, dil_t, pad_t, planes)
            dw = torch.zeros_like(w) if want_dw else None
            db = torch.zeros(Cout, dtype=torch.float32, device=dev) if want_db else None
            desc = (_hip.Conv2dReduceDesc * 1)()
            desc[0].partials = buf.data_ptr() + 4096
            desc[0].nblocks, desc[0].npart, desc[0].nw, desc[0].cout = nblocks, npart, KT * KF * Cin * Cout, Cout
            desc[0].dw = dw.data_ptr() if dw is not None else None
            desc[0].dbias = db.data_ptr() if db is not None else None
            call('ptts_conv2d_reduce_grouped', ctypes.cast(desc, ctypes.c_void_p), 1, stream(), tag=(1,))
        return dx, dw, db, None, None
    if planes == 3 and scale is not None and mode == IN_LRELU and want_dx and mask_src is None and _C2M.enabled and _BNStats.enabled and \
            _C2M.eligible(x, w, dil_t) and 0.0 <= alpha <= 1.0 and _C2MFused.ok(x, dil_t, pad_mode, planes) and \
            scale.data_ptr() % 16 == 0 and shift.data_ptr() % 16 == 0:
        # a BatchNormalization in front of the layer (the generator's stack): the fused matrix-core launch with that input's affine in its
        # Q staging, its mask and scale in the store of dx, and the affine's two gradient sums behind dW / dbias in the partial rows
        dx, buf, nblocks, npart = _conv2d_mfma_bwd_fused_affine(dy, x, w, alpha, scale, shift)
        nw = KT * KF * Cin * Cout
        dw = torch.zeros_like(w) if want_dw else None
        db = torch.zeros(Cout, dtype=torch.float32, device=dev) if want_db else None
        daff = torch.zeros(2 * Cin, dtype=torch.float32, device=dev) if want_affine else None
        nd = int(want_dw or want_db) + int(want_affine)
        if nd:
            desc = (_hip.Conv2dReduceDesc * nd)()
            i = 0
            if want_dw or want_db:
                desc[i].partials = buf.data_ptr() + 4096
                desc[i].nblocks, desc[i].npart, desc[i].nw, desc[i].cout = nblocks, npart, nw, Cout
                desc[i].dw = dw.data_ptr() if dw is not None else None
                desc[i].dbias = db.data_ptr() if db is not None else None
                i += 1
            if want_affine:
                desc[i].partials = buf.data_ptr() + 4096 + 4 * (nw + Cout)        # (the eight sums behind dW and dbias in every row)
                desc[i].nblocks, desc[i].npart, desc[i].nw, desc[i].cout = nblocks, npart, Cin, Cin
                desc[i].dw = daff.data_ptr()
                desc[i].dbias = daff.data_ptr() + 4 * Cin
            call('ptts_conv2d_reduce_grouped', ctypes.cast(desc, ctypes.c_void_p), nd, stream(), tag=(nd,))
        return dx, dw, db, (daff[:Cin] if want_affine else None), (daff[Cin:] if want_affine else None)
    dx = torch.empty_like(x) if want_dx else None
    dw = torch.empty_like(w) if (want_dw or want_db) else None
    db = torch.empty(Cout, dtype=torch.float32, device=dev) if want_db else None
    dscale = torch.empty(Cin, dtype=torch.float32, device=dev) if want_affine else None
    dshift = torch.empty(Cin, dtype=torch.float32, device=dev) if want_affine else None
    nws = _hip.lib().ptts_conv2d_bwd_workspace_bytes(B, T, F, Cin, Cout, KT, KF, dil_t)
    ws = _workspace(nws, dev)
    call('ptts_conv2d_bwd', ptr(dy), ptr(x), ptr(w), ptr(scale), ptr(shift), ptr(mask_src),
         ptr(dx), ptr(dw), ptr(db), ptr(dscale), ptr(dshift), ptr(ws), ws.numel(),
         B, T, F, Cin, Cout, KT, KF, dil_t, pad_mode, mode, alpha, stream(),
         tag=(B, T, F, Cin, Cout, mode, int(want_dx), int(want_dw or want_db), int(want_affine)))
    return dx, (dw if want_dw else None), db, dscale, dshift


def _conv2d_bwd_deferred(dy, x, w, mask_src, mode, alpha, dil_t, pad_mode, want_dx, gw, gb, planes=3):
    """conv2d backward whose dw / dbias stay as per-workgroup partial sums in a buffer of their own; the reduction into
    the gradient buffers gw / gb is queued for the grouped launch at flush time.  Returns dx (or None), or False when
    this shape has no tiled kernel."""
    B, T, F, Cin = x.shape
    KT, KF, _, Cout = w.shape
    if (planes == 1 or _C2M.enabled) and _C2M.eligible(x, w, dil_t) and 0.0 <= alpha <= 1.0:
        pad_t = _C2M.pad_t(dil_t, pad_mode)
        dx = None
        if want_dx and mode == IN_LRELU and _C2MFused.ok(x, dil_t, pad_mode, planes):
            dx, buf, nblocks, npart = _conv2d_mfma_bwd_fused(1, dy, x, None, w, alpha)      # dx + dW + dbias: one launch
        else:
            if want_dx:
                dx = _conv2d_mfma_fwd(dy, w, _C2M.table(w, True, planes), None, None, None, None, x if mode == IN_LRELU else None,
                                      IN_NONE, alpha, dil_t, 4 * dil_t - pad_t, planes, _is16(x))
            buf, nblocks, npart = _conv2d_mfma_wgrad(dy, x, mask_src, mode, alpha, dil_t, pad_t, planes)
        cur = torch.cuda.current_stream()
        if all(cur.cuda_stream != st.cuda_stream for st in _Deferred.streams):
            _Deferred.streams.append(cur)
        _Deferred.conv_items.append((buf, 4096, nblocks, npart, KT * KF * Cin * Cout, Cout, gw, gb))
        return dx
    nws = _hip.lib().ptts_conv2d_bwd_workspace_bytes(B, T, F, Cin, Cout, KT, KF, dil_t)
    if nws <= 16:
        return False
    dev = x.device
    buf = torch.empty(int(nws), dtype=torch.uint8, device=dev)
    dx = torch.empty_like(x) if want_dx else None
    nblocks = ctypes.c_int(0)
    call('ptts_conv2d_bwd_partials', ptr(dy), ptr(x), ptr(w), ptr(mask_src), ptr(dx), ptr(buf), buf.numel(),
         ctypes.byref(nblocks), B, T, F, Cin, Cout, KT, KF, dil_t, pad_mode, mode, alpha, stream(),
         tag=(B, T, F, Cin, Cout, mode, int(want_dx), 1, 0))
    cur = torch.cuda.current_stream()
    if all(cur.cuda_stream != s.cuda_stream for s in _Deferred.streams):
        _Deferred.streams.append(cur)
    nw = KT * KF * Cin * Cout
    _Deferred.conv_items.append((buf, 4096, nblocks.value, nw + Cout + 2 * Cin, nw, Cout, gw, gb))
    return dx


class _DenseSplit(object):
    """The Dense products with M >> N (forward, backward-data, the masked forward of the second-order sweep, the LSTM
    input projections) as bf16x6 split products (csrc/dense.hip): fp32 arithmetic on the bf16 matrix cores, like the context
    Conv1D and the Conv2D stacks.  The weight operand's planes are built once per update and kept per (weight view, stream).
    PTTS_DENSE_SPLIT=0 or dense_split(False) select the fp32-MFMA kernels of csrc/gemm.hip."""
    default = os.environ.get('PTTS_DENSE_SPLIT', '1') == '1'
    enabled = default
    planes = {}      # (id(owner), data_ptr, K, N, ldb, transB, stream) -> (owner, version, epoch, planes)
    wgrad_min_n = int(os.environ.get('PTTS_DENSE_WGRAD_MIN_N', '16'))      # narrowest weight gradient the split kernel takes

    @classmethod
    def get(cls, Bm, K, N, ldb, transB):
        """Planes of B[K][N] (transB: stored [N][K]) if Bm is (a view of) a weight of a flat parameter buffer, else None."""
        owner = Bm if hasattr(Bm, '_ptts_flat') else getattr(Bm, '_base', None)
        flat = getattr(owner, '_ptts_flat', None)
        if flat is None:
            return None
        sid = _hip.stream_id()
        key = (id(owner), Bm.data_ptr(), K, N, ldb, transB, sid)
        ent = cls.planes.get(key)
        if ent is None or ent[0] is not owner or ent[1] != owner._version or ent[2] != flat.epoch:
            reuse = ent is not None and ent[0] is owner
            buf = ent[3] if reuse else torch.empty(_hip.lib().ptts_dense_planes_bytes(N, K), dtype=torch.uint8, device=Bm.device)
            if len(cls.planes) > 512:
                cls.planes = {}
            cls.planes[key] = ent = (owner, owner._version, flat.epoch, buf, Bm)
            # ... and with it every other weight of the same flat buffer whose planes on this stream are out of date: after an update
            # all of a network's Dense kernels need theirs again, one grouped launch instead of one launch per kernel and direction
            todo = [(key, ent)]
            for k2, e2 in cls.planes.items():
                if k2 is not key and k2[6] == sid and len(e2) == 5 and getattr(e2[0], '_ptts_flat', None) is flat and \
                        (e2[1] != e2[0]._version or e2[2] != flat.epoch) and e2[4].data_ptr() == k2[1]:
                    todo.append((k2, e2))
            if len(todo) == 1:
                call('ptts_split3_dense_weight', ptr(Bm), ldb, K, N, transB, ptr(buf), stream(), tag=(K, N, transB))
            else:
                descs = (_hip.DenseSplitDesc * len(todo))()
                for d, (k2, e2) in zip(descs, todo):
                    d.w, d.planes, d.ldw, d.K, d.N, d.transposed = k2[1], e2[3].data_ptr(), k2[4], k2[2], k2[3], k2[5]
                    cls.planes[k2] = (e2[0], e2[0]._version, flat.epoch, e2[3], e2[4])
                call('ptts_split3_dense_weight_grouped', ctypes.cast(descs, ctypes.c_void_p), len(todo), stream(), tag=(len(todo),))
        return ent[3]

    @classmethod
    def clear(cls):
        """Mark every plane set out of date (keys and buffers stay: the next use rebuilds them, grouped)."""
        cls.planes = {k: (e[0], None, None, e[3], e[4]) for k, e in cls.planes.items() if len(e) == 5}

    @classmethod
    def eligible(cls, A, C, M, N, K, lda, ldc, in_side):
        """in_side: the tensors read beside A in 16-byte pieces (mask_src, scale, shift); C, bias and out_mask may be
        unaligned or N / ldc no multiple of 4 (the 65-bin spectral head): the kernel then stores element-wise."""
        if not (cls.enabled and M >= 1024 and N >= 16 and K >= 16 and K % 4 == 0 and lda % 4 == 0):
            return False
        if not _hip.lib().ptts_dense_bf16x6_supported(M, N, K, lda, ldc):       # (K > 65536, too many column blocks: the fp32 kernel)
            return False
        return all(t is None or t.data_ptr() % 16 == 0 for t in (A,) + tuple(in_side))


def dense_split(on):
    """Dense products as bf16x6 split products (True, the default) or on the fp32 matrix pipe (False); None restores the default."""
    _DenseSplit.enabled = _DenseSplit.default if on is None else bool(on)


def _ptr_off(t, off):
    """A tensor whose data pointer lies `off` elements behind t's (no copy, whatever t's strides are)."""
    return None if t is None else t.as_strided((1,), (1,), t.storage_offset() + off)


def gemm_raw(A, Bm, C, M, N, K, transA=0, lda=None, rows_per_seg=None, seg_stride=0, transB=0, ldb=None, ldc=None,
             bias=None, mode=IN_NONE, scale=None, shift=None, mask_src=None, alpha=0.3, accumulate=0, out_mask=None,
             colsum_b=None, res=None):
    """C[M,N] (+)= opA(A).opB(B) (+bias).  Pointers may be views with offsets; dims are the caller's contract.
    res [R, N] contiguous, M a multiple of R: C[m] += res[m % R] (the split Dense kernel adds it in its store; elsewhere an add pass)."""
    if res is not None:
        R = res.numel() // N
        assert not accumulate and out_mask is None and M % R == 0 and res.is_contiguous()
        N1 = N if (N <= 256 or N % 256 == 0 or N % 256 > 32) else N - N % 256
        if transA == 0 and seg_stride == 0 and (rows_per_seg is None or rows_per_seg == M) and colsum_b is None and N1 == N and \
                _DenseSplit.eligible(A, C, M, N, K, K if lda is None else lda, N if ldc is None else ldc, (scale, shift, mask_src)):
            planes = _DenseSplit.get(Bm, K, N, (N if transB == 0 else K) if ldb is None else ldb, transB)
            if planes is not None:
                call('ptts_dense_bf16x6_res', ptr(A), ptr(planes), ptr(bias), ptr(C), M, N, K, K if lda is None else lda, N if ldc is None else ldc,
                     mode, ptr(scale), ptr(shift), ptr(mask_src), alpha, ptr(res), R, N, None, stream(), tag=(M, N, K, transB, 'res'))
                return C
        gemm_raw(A, Bm, C, M, N, K, transA, lda, rows_per_seg, seg_stride, transB, ldb, ldc, bias, mode, scale, shift, mask_src, alpha)
        Cv = C.view(M // R, R, N) if (ldc is None or ldc == N) else None
        assert Cv is not None, 'gemm_raw: a residual needs a dense C'
        Cv.add_(res.view(1, R, N))
        return C
    for t in (A, Bm, C, bias, scale, shift, mask_src):
        if t is not None:
            assert t.is_cuda and t.dtype == torch.float32
    if lda is None:
        lda = K if transA == 0 else M
    if rows_per_seg is None:
        rows_per_seg = M if transA == 0 else K
    if ldb is None:
        ldb = N if transB == 0 else K
    if ldc is None:
        ldc = N
    if transA == 1 and transB == 0 and seg_stride == 0 and rows_per_seg == K and out_mask is None and bias is None \
            and _wgrad_split_ok(A.as_strided((1, 1), (lda, 1)), Bm.as_strided((1, 1), (ldb, 1)), mask_src, scale, shift, M, N, K):
        # a weight gradient C[M,N] = T(A)[K,M]^T . B[K,N] outside the deferred queue (the LSTM kernels): the split kernel adds
        # into C, so C (and the bias-gradient row) start from zero
        if not accumulate:
            C.zero_()
        if colsum_b is not None:
            colsum_b.zero_()
        ws = _workspace(_hip.lib().ptts_dense_wgrad_workspace_bytes(M, N, K), A.device)
        call('ptts_dense_wgrad_bf16x6', ptr(A), ptr(Bm), ptr(mask_src), ptr(scale), ptr(shift), ptr(C), ptr(colsum_b),
             ptr(ws), ws.numel(), M, N, K, lda, ldb, ldc, mode, alpha, stream(), tag=(M, N, K))
        return C
    if transA == 0 and seg_stride == 0 and rows_per_seg == M and colsum_b is None:
        # M >> N products against a weight: the bf16x6 split kernel; a few columns beyond a multiple of 256 (the 260-wide
        # spectral part) go to the thin fp32 kernel
        N1 = N if (N <= 256 or N % 256 == 0 or N % 256 > 32) else N - N % 256
        if _DenseSplit.eligible(A, C, M, N1, K, lda, ldc, (scale, shift, mask_src)):
            planes = _DenseSplit.get(Bm, K, N1, ldb, transB)
            if planes is not None and _BNStats.want and _BNStats.enabled and N1 == N and N % 4 == 0 and ldc % 4 == 0 and mask_src is None and \
                    out_mask is None and not accumulate and mode in (IN_NONE, IN_LRELU):
                # a Dense layer in front of a BatchNormalization layer: the launch leaves the column sums of what it stores (_BNStats)
                cap = _hip.lib().ptts_dense_bf16x6_stats_rows(M, N)
                part = torch.empty((cap, 2 * N), dtype=torch.float64, device=A.device)
                nrows = ctypes.c_int(0)
                call('ptts_dense_bf16x6_stats', ptr(A), ptr(planes), ptr(bias), ptr(C), M, N, K, lda, ldc, mode, ptr(scale), ptr(shift),
                     alpha, ptr(part), cap, ctypes.byref(nrows), stream(), tag=(M, N, K, transB, 'stats'))
                _BNStats.last = (part, nrows.value, M)
                return C
            if planes is not None:
                call('ptts_dense_bf16x6', ptr(A), ptr(planes), ptr(bias), ptr(C), M, N1, K, lda, ldc, mode, ptr(scale), ptr(shift),
                     ptr(mask_src), alpha, accumulate, ptr(out_mask), stream(), tag=(M, N1, K, transB))
                if N1 < N:
                    gemm_raw(A, _ptr_off(Bm, N1 * (ldb if transB else 1)), _ptr_off(C, N1), M, N - N1, K, lda=lda,
                             transB=transB, ldb=ldb, ldc=ldc, bias=_ptr_off(bias, N1), mode=mode, scale=scale,
                             shift=shift, mask_src=mask_src, alpha=alpha, accumulate=accumulate, out_mask=_ptr_off(out_mask, N1))
                return C
    call('ptts_gemm', ptr(A), ptr(Bm), ptr(bias), ptr(C), M, N, K, transA, lda, rows_per_seg, seg_stride,
         transB, ldb, ldc, mode, ptr(scale), ptr(shift), ptr(mask_src), alpha, accumulate, ptr(out_mask), ptr(colsum_b), stream(),
         tag=(M, N, K, transA, transB, int(seg_stride != 0)))
    return C


def colsums(x2d, mode=IN_NONE, scale=None, shift=None, mask_src=None, alpha=0.3):
    """fp64 [2C]: column sums and sums of squares of transform(x)."""
    f32c(x2d, 'colsums.x')
    rows, C = x2d.shape
    sums = torch.empty(2 * C, dtype=torch.float64, device=x2d.device)
    nws = _hip.lib().ptts_colstats_workspace_bytes(rows, C)
    ws = _workspace(nws, x2d.device)
    call('ptts_colstats', ptr(x2d), rows, C, mode, ptr(scale), ptr(shift), ptr(mask_src), alpha, ptr(sums),
         ptr(ws), ws.numel(), stream(), tag=(rows, C, mode))
    return sums


def _colsum_f32(x2d):
    return colsums(x2d)[:x2d.shape[1]].to(torch.float32)


def _affine_act_bwd_raw(dy, x, y, scale, shift, act, alpha, want_dx=True):
    f32c(dy, 'act_bwd.dy'); f32c(x, 'act_bwd.x')
    C = x.shape[-1]
    rows = x.numel() // C
    dx = torch.empty_like(x) if want_dx else None
    dsums = torch.empty(2 * C, dtype=torch.float64, device=x.device)
    nws = _hip.lib().ptts_colstats_workspace_bytes(rows, C)
    ws = _workspace(nws, x.device)
    call('ptts_affine_act_bwd', ptr(dy), ptr(x), ptr(y), ptr(scale), ptr(shift), ptr(dx), ptr(dsums), ptr(ws),
         ws.numel(), rows, C, act, alpha, stream())
    d32 = dsums.to(torch.float32)            # (one cast for both sums)
    return dx, d32[:C], d32[C:]


# ----------------------------------------------------------------------------------------------
# Conv2D  (kl.Conv2D: networks_critic.py:67; networktts.py:123; modeltts_common.py:100)
# ----------------------------------------------------------------------------------------------
class Conv2dFn(torch.autograd.Function):
    """bf16 = None: fp32 arithmetic and storage.  bf16 = 'out16' / 'out32' (BASELINE configs[2]): bf16 arithmetic on the
    matrix cores -- the input map may be stored as bf16 or fp32, the result is stored as bf16 / fp32; the gradient of a
    map is stored like the map; weight gradients and master weights stay fp32."""
    @staticmethod
    def forward(ctx, x, w, b, scale, shift, mode, alpha, dil_t, pad_mode, bf16=None):
        ctx.save_for_backward(x, w, scale, shift)
        ctx.has_b = b is not None
        ctx.cfg = (mode, alpha, dil_t, pad_mode)
        ctx.planes = 1 if bf16 else 3
        # persistent gradient buffers of kernel / bias (deferred, grouped reduction of the backward's partial sums)
        ctx.gw = grad_target(w) if _Deferred.active else None
        ctx.gb = grad_target(b) if (_Deferred.active and b is not None) else None
        ctx.set_materialize_grads(False)
        return _conv2d_fwd_raw(x, w, b, scale, shift, None, mode, alpha, dil_t, pad_mode, ctx.planes, bf16 == 'out16')

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            # no gradient arrives: the node is reachable only as the MASK operand of a backward-data Function of the gradient
            # penalty (x_hat's evaluation, whose own output is not part of the loss).  Materialised zeros would run this
            # layer's whole first-order backward -- dx and dW launches that add zeros -- once per critic step.
            return (None,) * 10
        x, w, scale, shift = ctx.saved_tensors
        mode, alpha, dil_t, pad_mode = ctx.cfg
        planes = ctx.planes
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_b and ctx.needs_input_grad[2]
        need_aff = scale is not None and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        dy = dy.contiguous()
        if _Flags.skip_param_grads:
            need_w = need_b = need_aff = False
        if torch.is_grad_enabled() and need_x:
            # differentiable backward (gradient penalty): dx is itself a Function of (dy, w)
            if scale is not None:
                raise RuntimeError('second-order gradients through a BatchNorm-fused conv2d are not supported')
            dx = Conv2dBwdDataFn.apply(dy, x, w, mode, alpha, dil_t, pad_mode, ctx.gw, planes)
            dw = db = None
            if need_w or need_b:
                with torch.no_grad():
                    if _conv2d_can_defer(ctx, need_w, need_b) and \
                            _conv2d_bwd_deferred(dy, x, w, None, mode, alpha, dil_t, pad_mode, False,
                                                 ctx.gw if need_w else None, ctx.gb if need_b else None, planes) is not False:
                        return dx, None, None, None, None, None, None, None, None, None
                    _, dw, db, _, _ = _conv2d_bwd_raw(dy, x, w, None, None, None, mode, alpha, dil_t, pad_mode,
                                                      False, need_w, need_b, False, planes)
            return dx, dw, db, None, None, None, None, None, None, None
        if (need_w or need_b) and not need_aff and scale is None and _conv2d_can_defer(ctx, need_w, need_b):
            dxd = _conv2d_bwd_deferred(dy, x, w, None, mode, alpha, dil_t, pad_mode, need_x,
                                       ctx.gw if need_w else None, ctx.gb if need_b else None, planes)
            if dxd is not False:
                return dxd, None, None, None, None, None, None, None, None, None
        dx, dw, db, dscale, dshift = _conv2d_bwd_raw(dy, x, w, scale, shift, None, mode, alpha, dil_t, pad_mode,
                                                     need_x, need_w, need_b, need_aff, planes)
        return dx, dw, db, dscale, dshift, None, None, None, None, None


def _conv2d_can_defer(ctx, need_w, need_b):
    return _Deferred.active and not _Flags.deterministic and (not need_w or ctx.gw is not None) and (not need_b or ctx.gb is not None)


class Conv2dBwdDataFn(torch.autograd.Function):
    """dx = d(a)/d(x) * conv^T(dy, w).  Linear in dy and in w; its backward is the second-order sweep."""
    @staticmethod
    def forward(ctx, dy, x, w, mode, alpha, dil_t, pad_mode, gw=None, planes=3):
        ctx.save_for_backward(dy, x, w)
        ctx.cfg = (mode, alpha, dil_t, pad_mode)
        ctx.gw = gw
        ctx.planes = planes
        ctx.set_materialize_grads(False)
        dx, _, _, _, _ = _conv2d_bwd_raw(dy, x, w, None, None, None, mode, alpha, dil_t, pad_mode,
                                         True, False, False, False, planes)
        return dx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, u):
        if u is None:
            return (None,) * 9
        dy, x, w = ctx.saved_tensors
        mode, alpha, dil_t, pad_mode = ctx.cfg
        planes = ctx.planes
        u = u.contiguous()
        m2, msk = (IN_MASKMUL, x) if mode == IN_LRELU else (IN_NONE, None)
        cot_dy = cot_w = None
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[2] and m2 == IN_MASKMUL and _C2M.enabled and _C2M.eligible(u, w, dil_t) and \
                0.0 <= alpha <= 1.0 and _C2MFused.ok(u, dil_t, pad_mode, planes):
            # the masked forward and the weight gradient of the sweep read the same staged tile u . lrelu'(x): one launch
            defer = _Deferred.active and not _Flags.deterministic and ctx.gw is not None
            cot_dy, buf, nblocks, npart = _conv2d_mfma_bwd_fused(2, u, dy, msk, w, alpha)
            if defer:
                cur = torch.cuda.current_stream()
                if all(cur.cuda_stream != st.cuda_stream for st in _Deferred.streams):
                    _Deferred.streams.append(cur)
                _Deferred.conv_items.append((buf, 4096, nblocks, npart, w.numel(), w.shape[3], ctx.gw, None))
                return cot_dy, None, None, None, None, None, None, None, None
            cot_w = torch.zeros_like(w)
            desc = (_hip.Conv2dReduceDesc * 1)()
            desc[0].partials = buf.data_ptr() + 4096
            desc[0].nblocks, desc[0].npart, desc[0].nw, desc[0].cout = nblocks, npart, w.numel(), w.shape[3]
            desc[0].dw = cot_w.data_ptr()
            desc[0].dbias = None
            call('ptts_conv2d_reduce_grouped', ctypes.cast(desc, ctypes.c_void_p), 1, stream(), tag=(1,))
            return cot_dy, None, cot_w, None, None, None, None, None, None
        if ctx.needs_input_grad[0]:
            cot_dy = _conv2d_fwd_raw(u, w, None, None, None, msk, m2, alpha, dil_t, pad_mode, planes, _is16(dy))
        if ctx.needs_input_grad[2]:
            if _Deferred.active and not _Flags.deterministic and ctx.gw is not None and \
                    _conv2d_bwd_deferred(dy, u, w, msk, m2, alpha, dil_t, pad_mode, False, ctx.gw, None, planes) is not False:
                return cot_dy, None, None, None, None, None, None, None, None
            _, cot_w, _, _, _ = _conv2d_bwd_raw(dy, u, w, None, None, msk, m2, alpha, dil_t, pad_mode,
                                                False, True, False, False, planes)
        return cot_dy, None, cot_w, None, None, None, None, None, None


def conv2d(v, w, b=None, dil_t=1, pad_mode=PAD_SAME, bf16=None):
    """z_out = conv2d(act(v), w) + b on [B,T,F,Cin]; `v` is a tensor or a Lazy.  bf16: see Conv2dFn."""
    z, mode, scale, shift, alpha = _prep(v)
    return Conv2dFn.apply(z, w, b, scale, shift, mode, alpha, dil_t, pad_mode, bf16)


# ----------------------------------------------------------------------------------------------
# Two evaluations of one layer as ONE forward launch (round 4): the critic step evaluates its critic on the stacked real / fake batch (2B)
# and on the interpolated sample (B) with the same weights.  Their forward passes are the same kernels on independent rows, so when the two
# inputs lie back to back in one buffer the forward is ONE launch over 3B rows, and its outputs -- two views of one buffer -- lie back to
# back for the next layer.  The backward passes differ (first-order for the stacked batch, the gradient penalty's passes for x^): each
# evaluation's backward is the single-evaluation Function's, run on that evaluation's tensors alone.
# ----------------------------------------------------------------------------------------------
class _PairFlags(object):
    enabled = os.environ.get('PTTS_PAIR_FORWARD', '1') == '1'


def pair_forward(on):
    _PairFlags.enabled = (os.environ.get('PTTS_PAIR_FORWARD', '1') == '1') if on is None else bool(on)


def _adjacent(x0, x1):
    return (torch.is_tensor(x0) and torch.is_tensor(x1) and x0.is_cuda and x0.dtype == torch.float32 and x1.dtype == torch.float32 and
            x0.is_contiguous() and x1.is_contiguous() and x0.shape[1:] == x1.shape[1:] and
            x1.data_ptr() == x0.data_ptr() + x0.numel() * 4 and
            x0.untyped_storage().data_ptr() == x1.untyped_storage().data_ptr())


def _cat_view(x0, x1):
    """The rows of x0 followed by the rows of x1 as one tensor (no copy: the two lie back to back in one storage)."""
    return torch.as_strided(x0, (x0.shape[0] + x1.shape[0],) + tuple(x0.shape[1:]), x0.stride())


class _FakeCtx(object):
    """What a single-evaluation Function's backward reads from its ctx (the pair Functions run it on one evaluation's tensors)."""
    def __init__(self, saved, needs, **attrs):
        self.saved_tensors, self.needs_input_grad = saved, needs
        for k, v in attrs.items():
            setattr(self, k, v)


def _sum_opt(a, b):
    return b if a is None else (a if b is None else a + b)


class Conv2dPairFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, x1, w, b, mode, alpha, dil_t, pad_mode):
        y = _conv2d_fwd_raw(_cat_view(x0, x1), w, b, None, None, None, mode, alpha, dil_t, pad_mode, 3, False)
        ctx.save_for_backward(x0, x1, w)
        ctx.has_b = b is not None
        ctx.cfg = (mode, alpha, dil_t, pad_mode)
        ctx.gw = grad_target(w) if _Deferred.active else None
        ctx.gb = grad_target(b) if (_Deferred.active and b is not None) else None
        ctx.set_materialize_grads(False)
        return y[:x0.shape[0]], y[x0.shape[0]:]

    @staticmethod
    def backward(ctx, dy0, dy1):
        x0, x1, w = ctx.saved_tensors
        nig = ctx.needs_input_grad
        dxs, dw, db = [None, None], None, None
        for i, (dy, x) in enumerate(((dy0, x0), (dy1, x1))):
            if dy is None:
                continue
            c = _FakeCtx((x, w, None, None), (nig[i], nig[2], nig[3], False, False), has_b=ctx.has_b, cfg=ctx.cfg, planes=3, gw=ctx.gw, gb=ctx.gb)
            g = Conv2dFn.backward(c, dy)
            dxs[i], dw, db = g[0], _sum_opt(dw, g[1]), _sum_opt(db, g[2])
        return dxs[0], dxs[1], dw, db, None, None, None, None


def conv2d_pair(v0, v1, w, b=None, dil_t=1, pad_mode=PAD_SAME, bf16=None):
    """conv2d of two evaluations; ONE forward launch when their inputs lie back to back, else two calls."""
    z0, mode0, sc0, sh0, al0 = _prep(v0)
    z1, mode1, sc1, sh1, al1 = _prep(v1)
    if _PairFlags.enabled and bf16 is None and sc0 is None and sc1 is None and mode0 == mode1 and al0 == al1 and _adjacent(z0, z1):
        return Conv2dPairFn.apply(z0, z1, w, b, mode0, al0, dil_t, pad_mode)
    return conv2d(v0, w, b, dil_t, pad_mode, bf16), conv2d(v1, w, b, dil_t, pad_mode, bf16)


# ----------------------------------------------------------------------------------------------
# The critic's whole Conv2D stack per launch (csrc/conv2d_chain.hip; networks_critic.py:64-70): bf16 storage path of
# BASELINE configs[2].  L x (Conv2D 5x5, 4 filters, bias, LeakyReLU) with the maps between the layers held in the LDS;
# the stored maps are POST-activation (they serve as operands and as LeakyReLU masks: slope > 0 keeps the sign).
# ----------------------------------------------------------------------------------------------
class _C2C(object):
    tables = {}         # (ids of the kernels, stream) -> (kernels, biases, versions, epoch, table buffer)

    @staticmethod
    def supported(F, ws):
        if not ws:
            return False
        KT, KF, cin0, Cc = ws[0].shape
        if any(tuple(w.shape) != (KT, KF, 4, Cc) for w in ws[1:]):
            return False
        return bool(_hip.lib().ptts_conv2d_chain_supported(int(F), len(ws), int(cin0), int(Cc), int(KT), int(KF)))

    @classmethod
    def table(cls, ws, bs):
        flat = getattr(ws[0], '_ptts_flat', None)
        epoch = None if flat is None else flat.epoch
        sid = _hip.stream_id()
        key = (tuple(id(w) for w in ws), sid)
        vers = tuple(w._version for w in ws) + tuple(-1 if b is None else b._version for b in bs)
        ent = cls.tables.get(key)
        if ent is None or any(a is not b for a, b in zip(ent[0], ws)) or ent[2] != vers or ent[3] != epoch or flat is None:
            tab = ent[4] if ent is not None else torch.empty(_hip.lib().ptts_conv2d_chain_tables_bytes(), dtype=torch.uint8, device=ws[0].device)
            for w in ws:
                f32c(w, 'conv2d_chain.w')
            wp = (ctypes.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
            bp = (ctypes.c_void_p * len(ws))(*[None if b is None else f32c(b, 'conv2d_chain.b').data_ptr() for b in bs])
            call('ptts_conv2d_chain_tables', wp, bp, ptr(tab), len(ws), int(ws[0].shape[2]), stream(), tag=(len(ws),))
            ent = (tuple(ws), tuple(bs), vers, epoch, tab)
            if len(cls.tables) > 64:
                cls.tables = {}
            cls.tables[key] = ent
        return ent[4]

    @classmethod
    def clear(cls):
        cls.tables = {}


def _chain_x0(x0):
    """[B,T,F] fp32 with unit stride along F and whole rows between frames (a column slice of [B,T,D] qualifies)."""
    if not (x0.is_cuda and x0.dtype == torch.float32 and x0.dim() == 3 and x0.stride(2) == 1 and x0.stride(0) == x0.shape[1] * x0.stride(1)):
        x0 = x0.contiguous()
        if not (x0.is_cuda and x0.dtype == torch.float32):
            raise _hip.HipLibraryError('conv2d_chain: expected a float32 device tensor [B,T,F], got {} {}'.format(x0.device, x0.dtype))
    return x0


def _chain_reduce(parts, nblocks, npart, ws, gws, gbs, want_b):
    """Partial rows [L][nblocks][npart] -> the gradient buffers: queued for the grouped launch inside deferred_weight_grads()
    (gws / gbs: the parameters' .grad views), else reduced now into fresh tensors, which are returned."""
    L = len(ws)
    deferred = gws is not None and all(g is not None for g in gws) and (not want_b or all(g is not None for g in gbs))
    if deferred:
        cur = torch.cuda.current_stream()
        if all(cur.cuda_stream != st.cuda_stream for st in _Deferred.streams):
            _Deferred.streams.append(cur)
        for l, w in enumerate(ws):
            _Deferred.conv_items.append((parts, l * nblocks * npart * 4, nblocks, npart, w.numel(), w.shape[3], gws[l], gbs[l] if want_b else None))
        return None, None
    dws = [torch.zeros_like(w) for w in ws]
    dbs = [torch.zeros(w.shape[3], dtype=torch.float32, device=w.device) for w in ws] if want_b else [None] * L
    descs = (_hip.Conv2dReduceDesc * L)()
    for l, (d, w) in enumerate(zip(descs, ws)):
        d.partials = parts.data_ptr() + l * nblocks * npart * 4
        d.nblocks, d.npart, d.nw, d.cout = nblocks, npart, w.numel(), w.shape[3]
        d.dw = dws[l].data_ptr()
        d.dbias = dbs[l].data_ptr() if want_b else None
    call('ptts_conv2d_reduce_grouped', ctypes.cast(descs, ctypes.c_void_p), L, stream(), tag=(L,))
    return dws, dbs


def _chain_defer_targets(ws, bs):
    if not (_Deferred.active and not _Flags.deterministic):
        return None, None
    return [grad_target(w) for w in ws], [None if b is None else grad_target(b) for b in bs]


class Conv2dChainFn(torch.autograd.Function):
    """a_L = stack(x0): x0 [B,T,F] fp32 -> [B,T,F,4] bf16 (post-activation).  Arguments after alpha: w_1, b_1, ..., w_L, b_L."""
    @staticmethod
    def forward(ctx, x0, alpha, *wb):
        ws, bs = list(wb[0::2]), list(wb[1::2])
        x0 = _chain_x0(x0)
        B, T, F = x0.shape
        L = len(ws)
        tab = _C2C.table(ws, bs)
        FP = (F + 1) & ~1
        maps = torch.empty((max(L - 1, 1), B, T, FP, 4), dtype=torch.bfloat16, device=x0.device)
        a_last = torch.empty((B, T, F, 4), dtype=torch.bfloat16, device=x0.device)
        call('ptts_conv2d_chain_fwd', ptr(x0), x0.stride(1), ptr(tab), ptr(maps), ptr(a_last), B, T, F, L, alpha, stream(), tag=(B, T, F, L))
        ctx.save_for_backward(x0, maps, a_last, tab, *ws)
        ctx.alpha, ctx.L, ctx.has_b = alpha, L, [b is not None for b in bs]
        ctx.gws, ctx.gbs = _chain_defer_targets(ws, bs)
        ctx.set_materialize_grads(False)
        return a_last

    @staticmethod
    def backward(ctx, d_last):
        if d_last is None:           # nothing flows back (e.g. the node is reachable only through a mask operand)
            return (None,) * (2 + 2 * ctx.L)
        x0, maps, a_last, tab = ctx.saved_tensors[:4]
        a_last = a_last.detach()     # an operand of the backward kernels, not a path of the graph: piecewise-constant masks
        ws = list(ctx.saved_tensors[4:])
        L, alpha = ctx.L, ctx.alpha
        B, T, F = x0.shape
        need_x = ctx.needs_input_grad[0]
        need_w = any(ctx.needs_input_grad[2 + 2 * l] or (ctx.has_b[l] and ctx.needs_input_grad[3 + 2 * l]) for l in range(L))
        if _Flags.skip_param_grads:
            need_w = False
        d_last = _st(d_last.contiguous(), 'conv2d_chain.d_last')
        g0 = None
        grads = [None] * (2 * L)
        if need_x:
            if torch.is_grad_enabled():
                # differentiable backward (gradient penalty): its own backward is the second-order sweep
                g0 = Conv2dChainBwdDataFn.apply(d_last, maps, a_last, tab, alpha, (ctx.gws, int(ws[0].shape[2])), *ws)
            else:
                g0 = torch.empty((B, T, F), dtype=torch.float32, device=x0.device)
                call('ptts_conv2d_chain_bwd_data', ptr(d_last), int(_is16(d_last)), ptr(maps), ptr(a_last), ptr(tab), None, ptr(g0),
                     B, T, F, L, alpha, stream(), tag=(B, T, F, L, 0))
        if need_w:
            with torch.no_grad():
                parts = torch.empty(_hip.lib().ptts_conv2d_chain_partials_bytes(L), dtype=torch.uint8, device=x0.device)
                nblocks, npart = ctypes.c_int(0), ctypes.c_int(0)
                call('ptts_conv2d_chain_bwd', ptr(d_last), int(_is16(d_last)), ptr(x0), x0.stride(1), ptr(maps), ptr(a_last), ptr(tab),
                     ptr(parts), parts.numel(), ctypes.byref(nblocks), ctypes.byref(npart), B, T, F, L, int(ws[0].shape[2]), alpha, stream(),
                     tag=(B, T, F, L))
                want_b = any(ctx.has_b)
                # the targets were noted at FORWARD time; queue only if a deferred_weight_grads() context is open NOW (a backward run
                # after the forward's context closed would append to a list nobody flushes, and the gradients would be lost)
                live = _Deferred.active and not _Flags.deterministic
                dws, dbs = _chain_reduce(parts, nblocks.value, npart.value, ws, ctx.gws if live else None, ctx.gbs if live else None, want_b)
                if dws is not None:
                    for l in range(L):
                        grads[2 * l] = dws[l]
                        grads[2 * l + 1] = dbs[l] if ctx.has_b[l] else None
        return (g0, None) + tuple(grads)


class Conv2dChainBwdDataFn(torch.autograd.Function):
    """g0 = d(sum d_last . a_L)/d(x0) through the stack; linear in d_last and in every kernel.  Its backward is the second-order
    sweep of the gradient penalty (optimizertts_wgan.py:53-68)."""
    @staticmethod
    def forward(ctx, d_last, maps, a_last, tab, alpha, extra, *ws):
        gws, cin0 = extra
        B, T, F, _ = a_last.shape
        L = len(ws)
        FP = (F + 1) & ~1
        gmaps = torch.empty((L, B, T, FP, 4), dtype=torch.bfloat16, device=a_last.device)
        g0 = torch.empty((B, T, F), dtype=torch.float32, device=a_last.device)
        call('ptts_conv2d_chain_bwd_data', ptr(d_last), int(_is16(d_last)), ptr(maps), ptr(a_last), ptr(tab), ptr(gmaps), ptr(g0),
             B, T, F, L, alpha, stream(), tag=(B, T, F, L, 1))
        ctx.save_for_backward(gmaps, maps, a_last, tab, *ws)
        ctx.cfg = (alpha, L, cin0, d_last.dtype)
        ctx.gws = gws
        ctx.set_materialize_grads(False)
        return g0

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, u0):
        if u0 is None:
            return (None,) * (6 + len(ctx.saved_tensors) - 4)
        gmaps, maps, a_last, tab = ctx.saved_tensors[:4]
        ws = list(ctx.saved_tensors[4:])
        alpha, L, cin0, d_dtype = ctx.cfg
        B, T, F, _ = a_last.shape
        u0 = f32c(u0.contiguous(), 'conv2d_chain.u0')
        out = torch.empty((B, T, F, 4), dtype=d_dtype, device=u0.device)
        parts = torch.empty(_hip.lib().ptts_conv2d_chain_partials_bytes(L), dtype=torch.uint8, device=u0.device)
        nblocks, npart = ctypes.c_int(0), ctypes.c_int(0)
        call('ptts_conv2d_chain_second', ptr(u0), ptr(gmaps), ptr(maps), ptr(a_last), ptr(tab), ptr(out), int(d_dtype == torch.bfloat16),
             ptr(parts), parts.numel(), ctypes.byref(nblocks), ctypes.byref(npart), B, T, F, L, cin0, alpha, stream(), tag=(B, T, F, L))
        need_w = any(ctx.needs_input_grad[6 + l] for l in range(L))
        grads = [None] * L
        if need_w:
            gws = ctx.gws if (_Deferred.active and not _Flags.deterministic) else None
            dws, _ = _chain_reduce(parts, nblocks.value, npart.value, ws, gws, [None] * L, False)
            if dws is not None:
                grads = dws
        return (out if ctx.needs_input_grad[0] else None, None, None, None, None, None) + tuple(grads)


def conv2d_chain(x0, ws, bs, alpha=0.3):
    """The whole stack: x0 [B,T,F] fp32 -> a_L [B,T,F,4] bf16 (LeakyReLU applied)."""
    wb = []
    for w, b in zip(ws, bs):
        wb += [w, b]
    return Conv2dChainFn.apply(x0, float(alpha), *wb)


# ----------------------------------------------------------------------------------------------
# Dense  (keras Dense: networktts.py:60 and the heads)
# ----------------------------------------------------------------------------------------------
def _dense_bwd_data(dy2, x2, w, mode, scale, shift, alpha, want_affine):
    """da = dy.W^T then dx = da * lrelu'(p) * scale (+ dscale/dshift sums)."""
    M, N = dy2.shape
    K = w.shape[0]
    da = torch.empty((M, K), dtype=torch.float32, device=dy2.device)
    if mode == IN_LRELU and scale is None:
        # no BatchNorm in front: the LeakyReLU mask of the layer input goes into the GEMM epilogue
        gemm_raw(dy2, w, da, M, K, N, transB=1, ldb=N, alpha=alpha, out_mask=x2)
        return da, None, None
    if mode == IN_LRELU and _BNStats.enabled and K % 4 == 0 and _DenseSplit.eligible(dy2, da, M, K, N, N, K, (x2, scale, shift)):
        # a BatchNormalization in front: the mask of lrelu(scale z + shift), the factor scale and the two column sums that are the
        # affine's gradients go into the product's store (ptts_dense_bf16x6_bwd_affine) -- no pass over da and z behind it
        planes = _DenseSplit.get(w, N, K, N, 1)
        if planes is not None:
            cap = _hip.lib().ptts_dense_bf16x6_stats_rows(M, K)
            part = torch.empty((cap, 2 * K), dtype=torch.float64, device=dy2.device)
            nrows = ctypes.c_int(0)
            call('ptts_dense_bf16x6_bwd_affine', ptr(dy2), ptr(planes), ptr(da), M, K, N, N, K, ptr(x2), ptr(scale), ptr(shift), alpha,
                 ptr(part), cap, ctypes.byref(nrows), stream(), tag=(M, K, N, 1, 'bwd_affine'))
            if not want_affine:
                return da, None, None
            dsums = torch.empty(2 * K, dtype=torch.float64, device=dy2.device)
            call('ptts_partial_rows_sum', ptr(part), nrows.value, 2 * K, ptr(dsums), stream())
            d32 = dsums.to(torch.float32)
            return da, d32[:K], d32[K:]
    gemm_raw(dy2, w, da, M, K, N, transB=1, ldb=N)
    if mode == IN_NONE:
        return da, None, None
    dx, dscale, dshift = _affine_act_bwd_raw(da, x2, None, scale, shift, ACT_LRELU, alpha)
    if not want_affine:
        dscale = dshift = None
    return dx, dscale, dshift


class DenseFn(torch.autograd.Function):
    """res (optional, [R..., N] with x's leading size a multiple of R): added to every R-row block of the product in the kernel's store
    -- the product of a concatenation part that k stacked evaluations share (layers.Dense over a LazyConcat)."""
    @staticmethod
    def forward(ctx, x, w, b, scale, shift, mode, alpha, res=None):
        f32c(x, 'dense.x'); f32c(w, 'dense.w')
        K, N = w.shape
        assert x.shape[-1] == K, 'dense: input width {} != kernel rows {}'.format(x.shape[-1], K)
        M = x.numel() // K
        y = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
        gemm_raw(x, w, y, M, N, K, bias=b, mode=mode, scale=scale, shift=shift, alpha=alpha, res=None if res is None else f32c(res, 'dense.res'))
        ctx.res_shape = None if res is None else tuple(res.shape)
        ctx.save_for_backward(x, w, scale, shift)
        ctx.has_b = b is not None
        ctx.cfg = (mode, alpha)
        # persistent gradient buffers of the kernel / bias (for the deferred, grouped weight-gradient launch)
        ctx.gw = grad_target(w) if _Deferred.active else None
        ctx.gb = grad_target(b) if (_Deferred.active and b is not None) else None
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:               # see Conv2dFn.backward
            return (None,) * 8
        x, w, scale, shift = ctx.saved_tensors
        mode, alpha = ctx.cfg
        K, N = w.shape
        M = x.numel() // K
        dres = None
        if ctx.res_shape is not None and ctx.needs_input_grad[7]:
            # gradient of the shared part: the sum of dy over the k row blocks it was added to (differentiable: plain torch ops)
            R = 1
            for d_ in ctx.res_shape: R *= d_
            k = dy.numel() // R
            dres = dy.reshape(ctx.res_shape) if k == 1 else dy.reshape((k,) + ctx.res_shape).sum(0)
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_b and ctx.needs_input_grad[2]
        need_aff = scale is not None and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        if _Flags.skip_param_grads:
            need_w = need_b = need_aff = False
        dy = dy.contiguous()
        dy2, x2 = dy.view(M, N), x.view(M, K)
        second_order = torch.is_grad_enabled() and need_x
        dx = dw = db = dscale = dshift = None
        if second_order:
            if scale is not None:
                raise RuntimeError('second-order gradients through a BatchNorm-fused dense layer are not supported')
            dx = DenseBwdDataFn.apply(dy, x, w, mode, alpha, ctx.gw)
        with torch.no_grad():
            if need_x and not second_order:
                dx2, dscale, dshift = _dense_bwd_data(dy2, x2, w, mode, scale, shift, alpha, need_aff)
                dx = dx2.view(x.shape)
            elif need_aff:
                _, dscale, dshift = _dense_bwd_data(dy2, x2, w, mode, scale, shift, alpha, True)
            if need_w and _Deferred.active and not _Flags.deterministic and ctx.gw is not None and N > 4 and (not need_b or ctx.gb is not None) \
                    and M >= 2048 and (scale is None or mode == IN_LRELU):
                # queued for the grouped launch, which adds into the .grad buffers itself (no dw / db for autograd)
                _defer_wgrad(x2, dy2, ctx.gw, ctx.gb if need_b else None, K, N, M, mode, scale, shift, None, alpha)
                need_w = need_b = False
            if need_w:
                dw = torch.empty_like(w)
                # the bias gradient (column sums of dy) rides on the weight-gradient product, whose B operand is dy
                fuse_b = need_b and N > 4
                if fuse_b:
                    db = torch.empty(N, dtype=torch.float32, device=dy.device)
                gemm_raw(x2, dy2, dw, K, N, M, transA=1, lda=K, rows_per_seg=M,
                         mode=mode, scale=scale, shift=shift, alpha=alpha, colsum_b=db if fuse_b else None)
            if need_b and db is None:
                db = _colsum_f32(dy2)
        return dx, dw, db, dscale, dshift, None, None, dres


class DenseBwdDataFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, x, w, mode, alpha, gw=None):
        K, N = w.shape
        M = x.numel() // K
        ctx.save_for_backward(dy, x, w)
        ctx.cfg = (mode, alpha)
        ctx.gw = gw                      # the kernel's persistent gradient buffer (deferred grouped launch), or None
        ctx.set_materialize_grads(False)
        dx2, _, _ = _dense_bwd_data(dy.view(M, N), x.view(M, K), w, mode, None, None, alpha, False)
        return dx2.view(x.shape)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, u):
        if u is None:
            return (None,) * 6
        dy, x, w = ctx.saved_tensors
        mode, alpha = ctx.cfg
        K, N = w.shape
        M = x.numel() // K
        u2 = u.contiguous().view(M, K)
        m2, msk = (IN_MASKMUL, x.view(M, K)) if mode == IN_LRELU else (IN_NONE, None)
        cot_dy = cot_w = None
        if ctx.needs_input_grad[0]:
            cot_dy = torch.empty((M, N), dtype=torch.float32, device=u.device)
            gemm_raw(u2, w, cot_dy, M, N, K, mode=m2, mask_src=msk, alpha=alpha)
            cot_dy = cot_dy.view(dy.shape)
        if ctx.needs_input_grad[2]:
            if _Deferred.active and not _Flags.deterministic and ctx.gw is not None and N > 4 and M >= 2048:
                _defer_wgrad(u2, dy.view(M, N), ctx.gw, None, K, N, M, m2, None, None, msk, alpha)
            else:
                cot_w = torch.empty_like(w)
                gemm_raw(u2, dy.view(M, N), cot_w, K, N, M, transA=1, lda=K, rows_per_seg=M,
                         mode=m2, mask_src=msk, alpha=alpha)
        return cot_dy, None, cot_w, None, None, None


class DensePairFn(torch.autograd.Function):
    """See Conv2dPairFn.  res (optional): the shared part's product, added to every R-row block of the 3B rows."""
    @staticmethod
    def forward(ctx, x0, x1, w, b, mode, alpha, res):
        K, N = w.shape
        xc = _cat_view(x0, x1)
        M = xc.numel() // K
        y = torch.empty(xc.shape[:-1] + (N,), dtype=torch.float32, device=xc.device)
        gemm_raw(xc, w, y, M, N, K, bias=b, mode=mode, alpha=alpha, res=res)
        ctx.save_for_backward(x0, x1, w)
        ctx.has_b = b is not None
        ctx.cfg = (mode, alpha)
        ctx.res_shape = None if res is None else tuple(res.shape)
        ctx.gw = grad_target(w) if _Deferred.active else None
        ctx.gb = grad_target(b) if (_Deferred.active and b is not None) else None
        ctx.set_materialize_grads(False)
        return y[:x0.shape[0]], y[x0.shape[0]:]

    @staticmethod
    def backward(ctx, dy0, dy1):
        x0, x1, w = ctx.saved_tensors
        nig = ctx.needs_input_grad
        dxs, dw, db, dres = [None, None], None, None, None
        for i, (dy, x) in enumerate(((dy0, x0), (dy1, x1))):
            if dy is None:
                continue
            c = _FakeCtx((x, w, None, None), (nig[i], nig[2], nig[3], False, False, False, False, nig[6]), has_b=ctx.has_b, cfg=ctx.cfg,
                         res_shape=ctx.res_shape, gw=ctx.gw, gb=ctx.gb)
            g = DenseFn.backward(c, dy)
            dxs[i], dw, db, dres = g[0], _sum_opt(dw, g[1]), _sum_opt(db, g[2]), _sum_opt(dres, g[7])
        return dxs[0], dxs[1], dw, db, None, None, dres


def dense_pair(v0, v1, w, b=None, res=None):
    """dense of two evaluations; ONE forward launch when their inputs lie back to back, else two calls."""
    z0, mode0, sc0, sh0, al0 = _prep(v0)
    z1, mode1, sc1, sh1, al1 = _prep(v1)
    if _PairFlags.enabled and sc0 is None and sc1 is None and mode0 == mode1 and al0 == al1 and _adjacent(z0, z1):
        return DensePairFn.apply(z0, z1, w, b, mode0, al0, None if res is None else res.contiguous())
    return dense(v0, w, b, res), dense(v1, w, b, res)


def dense(v, w, b=None, res=None):
    """z_out = act(v).W + b over the last axis (+ res, broadcast over row blocks: see DenseFn)."""
    z, mode, scale, shift, alpha = _prep(v)
    if res is None:
        return DenseFn.apply(z, w, b, scale, shift, mode, alpha)
    return DenseFn.apply(z, w, b, scale, shift, mode, alpha, res.contiguous())


# ----------------------------------------------------------------------------------------------
# Conv1D over time as an implicit GEMM  (kl.Conv1D: networktts.py:117)
# ----------------------------------------------------------------------------------------------
def _pad_time(a, lo, hi):
    B, T, C = a.shape
    ap = torch.zeros((B, T + lo + hi, C), dtype=torch.float32, device=a.device)
    ap[:, lo:lo + T].copy_(a)
    return ap


class _C1Cache(object):
    """The last context-Conv1D product computed WITHOUT autograd inside one device step (the generator's context conv
    of the critic step's fake sample): the generator step that follows on the same batch multiplies the same input by the
    same, not yet updated, kernel -- it takes the stored product (and padded input) instead of 165 GFLOP again."""
    enabled = False
    capture = False
    key = None
    ap = None
    y = None


def conv1d_cache(on):
    _C1Cache.enabled = bool(on)
    _C1Cache.key = _C1Cache.ap = _C1Cache.y = None


class _C1Split(object):
    """Context-Conv1D forward as an fp32 product on the bf16 matrix cores (csrc/split.hip: three-way bf16 split of both
    operands, six bf16 MFMA products, fp32 accumulation -- fp32-level accuracy, see tools/bf16x6_accuracy.py and
    tests/test_ops_gpu.py).  ON by default (round-1 verdict: admissible as fp32 arithmetic -- planes checked bit for bit
    against oracle.np_split3_bf16, every kernel tested against the fp64 oracle at its fp32 sibling's tolerance);
    PTTS_CONV1D_SPLIT=0, conv1d_split(False) or cfg.train_wgan_split_bf16 = False select the fp32-MFMA kernels.
    The planes of the frames are kept for the tensor they were made from (generator and critic convolve the same
    context input, in the critic step and again in the generator step); the planes of a kernel until it is updated."""
    default = os.environ.get('PTTS_CONV1D_SPLIT', '1') == '1'
    enabled = default
    x_src = None        # the tensor (kept alive: its address cannot be reused) ...
    x_key = None        # ... its version / shape / padding
    x_planes = None
    w_planes = {}       # id(w) -> (w, version, flat epoch, planes)

    @staticmethod
    def eligible(a, w):
        B, T, Cin = a.shape
        KW, _, N = w.shape
        return a.is_cuda and N % 128 == 0 and (T >= 128 or B == 1) and 2 * (KW - 1) + 127 < 176

    @classmethod
    def frames(cls, a, pl, pr):
        B, T, Cin = a.shape
        Cp = (Cin + 31) // 32 * 32
        key = (a._version, tuple(a.shape), pl, pr, _hip.stream_id())
        if cls.x_src is a and cls.x_key == key:
            return cls.x_planes, Cp
        planes = torch.empty((3, Cp // 32, B, T + pl + pr, 32), dtype=torch.bfloat16, device=a.device)
        call('ptts_split3_frames', ptr(a), ptr(planes[0]), ptr(planes[1]), ptr(planes[2]), B, T, Cin, pl, pr, Cp, stream(),
             tag=(B, T, Cin))
        cls.x_src, cls.x_key, cls.x_planes = a, key, planes
        return planes, Cp

    @classmethod
    def kernel(cls, w):
        KW, Cin, N = w.shape
        Cp = (Cin + 31) // 32 * 32
        flat = getattr(w, '_ptts_flat', None)
        epoch = None if flat is None else flat.epoch
        ent = cls.w_planes.get(id(w))
        sid = _hip.stream_id()
        if ent is not None and ent[0] is w and ent[1] == w._version and ent[2] == epoch and ent[4] == sid and flat is not None:
            return ent[3]
        planes = ent[3] if ent is not None and ent[0] is w and ent[4] == sid else torch.empty((3, Cp // 32, N, KW, 32), dtype=torch.bfloat16, device=w.device)
        call('ptts_split3_weight_t', ptr(w), ptr(planes[0]), ptr(planes[1]), ptr(planes[2]), KW, Cin, N, Cp, stream(),
             tag=(KW, Cin, N))
        cls.w_planes[id(w)] = (w, w._version, epoch, planes, sid)
        return planes

    WGRAD_KW = (3, 5, 21)        # instantiations of wgrad_bf16x6_kernel<KW>
    xt_src = None
    xt_key = None
    xt_planes = None

    @staticmethod
    def eligible_wgrad(KW, N):
        return KW in _C1Split.WGRAD_KW and N % 32 == 0

    @staticmethod
    def plane_len(B, T, KW):
        qsteps = (B * (T + KW - 1) + 31) // 32
        return (qsteps * 32 + 64 + 63) // 64 * 64

    @staticmethod
    def transposed(x, B, T, C, pl, Tp, Pp):
        """Three frame-major bf16 planes [3][Crows][Pp] of x [B][T][C]: element (c, b Tp + pl + t), zero elsewhere."""
        Crows = (C + 63) // 64 * 64
        planes = torch.empty((3, Crows, Pp), dtype=torch.bfloat16, device=x.device)
        call('ptts_split3_frames_t', ptr(x), ptr(planes[0]), ptr(planes[1]), ptr(planes[2]), B, T, C, pl, Tp, Crows, Pp, stream(),
             tag=(B, T, C))
        return planes, Crows

    @classmethod
    def frames_t(cls, src, saved, padded, B, T, C, KW):
        """Frame-major planes of the zero-padded frames of the layer input (`saved`: the padded buffer [B][T+KW-1][C], or the
        input itself, padded on the fly by the split pass), kept for `src` (the tensor the layer was called with): the
        weight gradients of generator and critic read the same context input."""
        Tp = T + KW - 1
        Pp = cls.plane_len(B, T, KW)
        key = None if src is None else (src._version, tuple(src.shape), KW, _hip.stream_id())
        if src is not None and cls.xt_src is src and cls.xt_key == key:
            return cls.xt_planes
        if padded:
            planes, Crows = cls.transposed(saved, B, Tp, C, 0, Tp, Pp)
        else:
            planes, Crows = cls.transposed(saved, B, T, C, (KW - 1) // 2, Tp, Pp)
        out = (planes, Crows, Pp)
        if src is not None:
            cls.xt_src, cls.xt_key, cls.xt_planes = src, key, out
        return out

    @classmethod
    def clear(cls):
        cls.x_src = cls.x_key = cls.x_planes = None
        cls.xt_src = cls.xt_key = cls.xt_planes = None
        cls.w_planes = {}


class _C1WgradT(object):
    """Context-Conv1D weight gradient in exact fp32 over frame-major operands (csrc/conv1d_wgrad.hip): both operands are
    transposed once so that the reduction index is contiguous, and the kernel forms every tap's operand in registers.
    PTTS_WGRAD_T=0 switches back to the stream-K product of gemm.hip.  The transposed frames are kept for the tensor the
    layer was called with (generator and critic convolve the same context input)."""
    enabled = os.environ.get('PTTS_WGRAD_T', '1') == '1'
    src = None
    key = None
    planes = None

    @classmethod
    def frames_t(cls, src, ap, KW):
        B, Tp, C = ap.shape
        Pp = _C1Split.plane_len(B, Tp - (KW - 1), KW)
        Crows = (C + 63) // 64 * 64
        key = None if src is None else (src._version, tuple(src.shape), KW, _hip.stream_id())
        if src is not None and cls.src is src and cls.key == key:
            return cls.planes
        xt = torch.empty((Crows, Pp), dtype=torch.float32, device=ap.device)
        call('ptts_transpose_frames', ptr(ap), ptr(xt), B, Tp, C, 0, Tp, Crows, Pp, stream(), tag=(B, Tp, C))
        out = (xt, Crows, Pp)
        if src is not None:
            cls.src, cls.key, cls.planes = src, key, out
        return out

    @classmethod
    def clear(cls):
        cls.src = cls.key = cls.planes = None


def clear_caches():
    """Forget every operand derived from a weight or an input (bf16 planes, transposed frames, Toeplitz tables, the reused
    context-Conv1D product).  A hipGraph capture must start from empty caches: an entry made before the capture would be
    taken instead of being recomputed INSIDE the graph, and every replay would then read the stale copy."""
    _C1Split.clear()
    _C1FFT.clear()
    _C1WgradT.clear()
    _C2M.clear()
    _C2C.clear()
    _DenseSplit.clear()
    _C1Cache.key = _C1Cache.ap = _C1Cache.y = None


class _C1FFT(object):
    """Context Conv1D in the frequency domain (round 3), overlap-save over segments of S output frames:
        y[b, sS + t] = sum_k x[b, sS + t + k - pl] w[k]   is, per segment, a circular correlation of length P >= S + KW - 1 of the
        window xseg = x[b, sS - pl ... sS - pl + P) (zero outside the utterance) with the kernel, i.e.  C^_f = X^_f H_f,
        H_f = sum_k w[k] e^{+2 pi i f k / P},  valid for t < S.
    Every stage is a batched bf16x6 split product of csrc/dense.hip (fp32 arithmetic: six bf16 MFMA products, fp32 accumulation) --
    the transforms are matrix products with the twiddle matrices, so there is no FFT kernel and P need not be a power of two:
        1  X^[(f,part), z, c]   = D [2 NB x P] . xseg_z [P x Cin]        z = (b, s); once per input: generator and critic convolve the same
        2  Y^_f [(part,z), n]   = [Xr | -Xi ; Xi | Xr]_f [2 B NS x 2 Kh] . [Hr ; Hi]_f [2 Kh x N]      for the NB = P/2 + 1 frequencies
        3  y_z [S x N]          = E [S x 2 NB] . Y^_z [2 NB x N] + bias
    and the weight gradient by the correlation theorem, sum_t xseg[t + k] dyseg[t] = (1/P) sum_f X^_f conj(DY^_f) e^{2 pi i f k / P}:
        4  DY^ = D . dyseg_z (the S gradient frames of a segment, zero-padded to P);  G_f^T = DY^_f^T [N x 2 B NS] . [..]_f [2 B NS x 2 Kh]
           = (Gr | -Gi)_f^T;  dW[k] = sum_f T2c[k,f] Gr_f + T2s[k,f] (-(-Gi_f))  (fused inverse transform, ptts_conv1d_freq_wgrad_inverse).
    KW = 21 taps over Cin = 601 channels cost 2 T KW Cin N flop per sample in the time domain; here the taps are gone from the big
    product.  Segments (S = 100 at T = 400: P = 120, 61 frequencies instead of 211 for the whole utterance at once) keep the
    transformed kernel small -- its planes are what the per-frequency product streams: 114 MB instead of 394 -- at 20 % more flops.
    Reference: kl.Conv1D of networktts.py:116-120 (pCNN1D), 'same' padding, cross-correlation as TF computes it.
    On by default for fp32 arithmetic (not in the one-product bf16 mode, whose time-domain kernel is as fast); PTTS_CONV1D_FFT=0 /
    conv1d_fft(False) select the time-domain kernels; PTTS_CONV1D_FFT_SEG=0 transforms whole utterances (one segment)."""
    default = os.environ.get('PTTS_CONV1D_FFT', '1') == '1'
    enabled = default
    KWS = (3, 5, 7, 9, 11, 21)      # instantiations of conv1d_wdft_planes_kernel<KW> / conv1d_freq_wgrad_inverse_kernel<KW>
    wgrad_enabled = os.environ.get('PTTS_CONV1D_FFT_WGRAD', '1') == '1'
    # BASELINE configs[2] (ops.bf16_products): the frequency-domain path with its BIG products -- the per-frequency product of the forward
    # and the per-frequency correlation of the weight gradient, 80 % of its flops -- as ONE bf16 product of the operands' roundings
    # (X^ and H^ / DY^ rounded once, fp32 accumulation), the small transforms (DFT, inverse DFT) still six products: the same single
    # operand rounding per product as the time-domain one-product kernels it replaces, a sixth of the matrix work.  PTTS_CONV1D_FFT_BF16=0
    # keeps the time-domain kernels in that mode.
    bf16_one_product = os.environ.get('PTTS_CONV1D_FFT_BF16', '1') == '1'
    seg_target = int(os.environ.get('PTTS_CONV1D_FFT_SEG', '100'))      # preferred segment length (0: one segment per utterance)
    consts = {}         # (T, KW, device, stream) -> dict of the geometry and the twiddle operands
    x_src = None; x_key = None; x_hat = None
    xw_src = None; xw_key = None; xw_planes = None
    w_hat = {}          # (id(w), stream) -> (w, version, epoch, None, planes, geometry)
    bufs = {}           # persistent buffers by (name, key, stream)

    @staticmethod
    def eligible(a, w):
        B, T, Cin = a.shape
        KW, _, N = w.shape
        if not (a.is_cuda and B * T >= 4096 and KW >= 3 and KW % 2 == 1 and T % 4 == 0 and N % 4 == 0 and Cin >= 64):
            return False
        # launch limits of the stages: ptts_split3_frame_windows and ptts_dense_bf16x6_batched put the B * NS segments on
        # gridDim.y / z (<= 65535); larger batches take the time-domain kernels instead of failing in the middle of the forward
        return B * (T // _C1FFT.segment(T, KW)) <= 65535

    @classmethod
    def segment(cls, T, KW):
        """S: the divisor of T closest to the target (a multiple of 4, at least 2 KW), or T itself."""
        tgt = cls.seg_target
        if tgt <= 0:
            return T
        best = T
        for S in range(4, T + 1, 4):
            if T % S == 0 and S >= 2 * KW and abs(S - tgt) < abs(best - tgt):
                best = S
        return best

    @classmethod
    def _buf(cls, name, key, nfloats, dev):
        sid = _hip.stream_id()
        k = (name, key, sid)
        t = cls.bufs.get(k)
        if t is None:
            # one buffer per (name, stream): an entry of another shape (hundreds of MB) is dropped when the shape changes
            for k2 in [k2 for k2 in cls.bufs if k2[0] == name and k2[2] == sid]:
                del cls.bufs[k2]
            t = cls.bufs[k] = torch.zeros(nfloats, dtype=torch.float32, device=dev)
        return t

    @classmethod
    def _scratch(cls, name, nbytes, dev):
        """A persistent byte buffer per (name, stream), grown on demand: the transforms' temporaries are hundreds of megabytes a
        call and would otherwise churn the caching allocator.  Valid until the next call that asks for the same name."""
        k = ('scratch', name, _hip.stream_id())
        t = cls.bufs.get(k)
        if t is None or t.numel() < nbytes:
            t = cls.bufs[k] = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        return t

    @classmethod
    def const(cls, T, KW, Cin, dev):
        key = (T, KW, Cin, dev.index, _hip.stream_id(), cls.seg_target)
        c = cls.consts.get(key)
        if c is not None:
            return c
        import math
        S = cls.segment(T, KW)
        NS = T // S
        P = (S + KW - 1 + 3) // 4 * 4            # the window: a reduction length of the first product (a multiple of 4), even
        NB = P // 2 + 1
        pl = (KW - 1) // 2
        f = torch.arange(NB, dtype=torch.float64).view(NB, 1)
        i = torch.arange(P, dtype=torch.float64).view(1, P)
        th = 2.0 * math.pi * torch.remainder(f * i, P) / P                                       # [NB, P]
        D = torch.stack([torch.cos(th), -torch.sin(th)], dim=1).reshape(2 * NB, P)               # rows (f, part): Xr, Xi
        cf = torch.full((NB,), 2.0, dtype=torch.float64); cf[0] = 1.0; cf[P // 2] = 1.0
        R = 2 * NB
        Rp = (R + 3) // 4 * 4
        E = torch.zeros(S, Rp, dtype=torch.float64)                                              # y[t] = sum_r E[t, r] Y^[r],  t < S
        E[:, 0:R:2] = (torch.cos(th[:, :S]) * cf.view(NB, 1) / P).t()
        E[:, 1:R:2] = (-torch.sin(th[:, :S]) * cf.view(NB, 1) / P).t()
        k = torch.arange(KW, dtype=torch.float64).view(1, KW)
        tk = 2.0 * math.pi * torch.remainder(f * k, P) / P                                       # [NB, KW]
        Tw = torch.stack([torch.cos(tk), torch.sin(tk)], dim=1).reshape(R, KW)                   # rows (f, part): Hr, Hi  (H_f = sum_k w[k] e^{+i tk})
        NBp = (NB + 3) // 4 * 4
        T2 = torch.zeros(2 * KW, NBp, dtype=torch.float64)                                      # dW[k] = sum_f T2[k, f] Gr_f + T2[KW + k, f] (-Gi)_f ... see wgrad
        T2[:KW, :NB] = (torch.cos(tk) * cf.view(NB, 1) / P).t()
        T2[KW:, :NB] = (torch.sin(tk) * cf.view(NB, 1) / P).t()
        TP = (2 * KW + 15) // 16 * 16
        T2f = torch.zeros(NB, TP, dtype=torch.float64)                                          # the same, frequency-major (the fused inverse kernel)
        T2f[:, :2 * KW] = T2[:, :NB].t()
        to = lambda m: m.to(torch.float32).to(dev).contiguous()
        c = {'S': S, 'NS': NS, 'P': P, 'NB': NB, 'R': R, 'Rp': Rp, 'NBp': NBp, 'TP': TP, 'pl': pl, 'Kh': (Cin + 7) // 8 * 8,
             'D': to(D), 'E': to(E), 'Tw': to(Tw), 'T2': to(T2), 'T2f': to(T2f)}
        cls.consts[key] = c
        return c

    @classmethod
    def x_transform(cls, a, KW):
        """[Xr | -Xi ; Xi | Xr] per frequency: Ap [NB][2][B NS][2 Kh] fp32, kept for the tensor it was made from."""
        B, T, Cin = a.shape
        key = (a._version, tuple(a.shape), KW, _hip.stream_id())
        if cls.x_src is a and cls.x_key == key:
            return cls.x_hat
        c = cls.const(T, KW, Cin, a.device)
        NB, Kh, NS, S, P = c['NB'], c['Kh'], c['NS'], c['S'], c['P']
        Z = B * NS
        Ap = cls._buf('Ap', (B, T, Cin, KW, S), NB * 2 * Z * 2 * Kh, a.device)                     # pad columns stay zero
        lib = _hip.lib()
        npb = lib.ptts_dense_planes_bytes(Cin, P)
        xpl = cls._scratch('xpl', Z * npb, a.device)
        call('ptts_split3_frame_windows', ptr(a), B, T, Cin, NS, S, -c['pl'], P, P, ptr(xpl), npb, stream(), tag=('x', Z))
        call('ptts_dense_bf16x6_batched', ptr(c['D']), 0, ptr(xpl), npb, None, ptr(Ap), 2 * Kh, Z, c['R'], Cin, P, P, Z * 2 * Kh, 3, stream(),
             tag=('dft', Z, c['R'], Cin, P))
        call('ptts_dft_mirror', ptr(Ap), NB, Z, Cin, Kh, stream())
        cls.x_src, cls.x_key, cls.x_hat = a, key, Ap
        return Ap

    @classmethod
    def kernel(cls, w, T):
        """Planes of [Hr ; Hi]_f [2 Kh x N] for all frequencies, rebuilt when the kernel changes."""
        KW, Cin, N = w.shape
        flat = getattr(w, '_ptts_flat', None)
        epoch = None if flat is None else flat.epoch
        sid = _hip.stream_id()
        c = cls.const(T, KW, Cin, w.device)
        NB, Kh = c['NB'], c['Kh']
        geo = (c['S'], NB, Kh)
        ent = cls.w_hat.get((id(w), sid))
        if ent is not None and ent[0] is w and ent[1] == w._version and ent[2] == epoch and flat is not None and ent[5] == geo:
            if cls.frozen_log is not None and id(flat) in cls.frozen:
                cls.frozen_log.append((w, ent[4], T))        # (a capture reads this buffer: its owner keeps it current, refresh_planes)
            return ent[4]
        lib = _hip.lib()
        npb = lib.ptts_dense_planes_bytes(N, 2 * Kh)
        if ent is not None and ent[0] is w and ent[5] == geo:
            planes = ent[4]
        else:
            planes = torch.empty(NB * npb, dtype=torch.uint8, device=w.device)
        cls._build_kernel_planes(w, c, planes)
        if len(cls.w_hat) >= 8 and (id(w), sid) not in cls.w_hat:
            cls.w_hat = {k: e for k, e in cls.w_hat.items() if id(getattr(e[0], '_ptts_flat', None)) in cls.frozen}      # kernels of optimisers long gone would otherwise keep their plane sets alive
        cls.w_hat[(id(w), sid)] = (w, w._version, epoch, None, planes, geo, T)
        return planes

    @classmethod
    def _build_kernel_planes(cls, w, c, planes):
        KW, Cin, N = w.shape
        NB, Kh = c['NB'], c['Kh']
        if KW in cls.KWS:
            call('ptts_conv1d_freq_kernel_planes', ptr(w), ptr(c['Tw']), ptr(planes), NB, KW, Cin, N, Kh, stream(), tag=(NB, KW, Cin, N))
        else:
            # any other odd kernel size: the twiddle product as a GEMM (pad rows zero), then one strided split
            npb = _hip.lib().ptts_dense_planes_bytes(N, 2 * Kh)
            What = torch.zeros(c['R'] * Kh * N, dtype=torch.float32, device=w.device)
            gemm_raw(c['Tw'], w.view(KW, Cin * N), What, c['R'], Cin * N, KW, lda=KW, ldb=Cin * N, ldc=Kh * N)
            call('ptts_split3_dense_weight_strided', ptr(What), 2 * Kh * N, ptr(planes), npb, NB, N, 2 * Kh, N, 0, stream(), tag=('w', NB))

    # Kernels of a network that is FROZEN inside a replayed step (the generator inside the critic step's hipGraph): ids of their flat
    # parameter buffers.  Their planes on a stream survive clear() -- a capture then finds them and does NOT rebuild them inside the
    # graph, where they would be rebuilt on every replay although the weights change once per generator update (42 us a replay for the
    # generator's context kernel) -- and the owner of the graph refreshes them before a replay when the weights have changed.
    frozen = set()
    frozen_log = None      # a list while a capture runs: (kernel, planes buffer, T) of every frozen kernel's planes the capture reads

    @classmethod
    def refresh_planes(cls, items):
        """Rebuild, on the CURRENT stream, the planes buffers a captured graph reads for frozen kernels: items = [(w, planes, T)] as
        logged during its capture (the buffers are the graph's inputs; the cache may meanwhile hold other buffers for these kernels)."""
        for w, planes, T in items:
            KW, Cin, N = w.shape
            cls._build_kernel_planes(w, cls.const(T, KW, Cin, w.device), planes)
        return len(items)

    @classmethod
    def forward(cls, a, w, b, y):
        B, T, Cin = a.shape
        KW, _, N = w.shape
        c = cls.const(T, KW, Cin, a.device)
        NB, Kh, NS, S = c['NB'], c['Kh'], c['NS'], c['S']
        Z = B * NS
        lib = _hip.lib()
        Ap = cls.x_transform(a, KW)
        wpl = cls.kernel(w, T)
        npw = lib.ptts_dense_planes_bytes(N, 2 * Kh)
        Yh = cls._scratch('Yh', NB * 2 * Z * N * 4, a.device)                                        # [NB][2][Z][N] fp32
        call('ptts_dense_bf16x6_batched', ptr(Ap), 2 * Z * 2 * Kh, ptr(wpl), npw, None, ptr(Yh), 2 * Z * N, NB, 2 * Z, N, 2 * Kh,
             2 * Kh, N, 1 if _Flags.bf16_products else 3, stream(), tag=('freq', NB, 2 * Z, N, 2 * Kh))
        npy = lib.ptts_dense_planes_bytes(N, c['R'])
        ypl = cls._scratch('ypl', Z * npy, a.device)
        call('ptts_split3_dense_weight_strided', ptr(Yh), N, ptr(ypl), npy, Z, Z * N, c['R'], N, 0, stream(), tag=('y', Z))
        # segment z = (b, s) writes the S frames y[b, s S ...]: consecutive blocks of S N floats (T = NS S)
        call('ptts_dense_bf16x6_batched', ptr(c['E']), 0, ptr(ypl), npy, ptr(b), ptr(y), S * N, Z, S, N, c['Rp'], c['Rp'], N, 3, stream(),
             tag=('idft', Z, S, N, c['Rp']))

    @classmethod
    def has_x(cls, a, KW):
        return a is not None and cls.x_src is a and cls.x_key == (a._version, tuple(a.shape), KW, _hip.stream_id())

    @classmethod
    def wgrad(cls, a, dy, KW):
        """dW [KW, Cin, N] of the layer from the transform of its input (kept from the forward) and of dy (see the class comment, 4)."""
        B, T, Cin = a.shape
        N = dy.shape[-1]
        c = cls.const(T, KW, Cin, a.device)
        NB, Kh, NS, S, P = c['NB'], c['Kh'], c['NS'], c['S'], c['P']
        Z = B * NS
        lib = _hip.lib()
        dev = a.device
        Ap = cls.x_hat
        # planes of [Xr | -Xi ; Xi | Xr]_f as the right operand [K = 2Z][N' = 2 Kh]: once per input
        npx = lib.ptts_dense_planes_bytes(2 * Kh, 2 * Z)
        if cls.xw_src is not a or cls.xw_key != cls.x_key or cls.xw_planes is None:
            xw = cls._scratch('xw', NB * npx, dev)
            call('ptts_split3_dense_weight_strided', ptr(Ap), 2 * Z * 2 * Kh, ptr(xw), npx, NB, 2 * Kh, 2 * Z, 2 * Kh, 0, stream(), tag=('xw', NB))
            cls.xw_src, cls.xw_key, cls.xw_planes = a, cls.x_key, xw
        # DY' = DFT of the segments' gradient frames (S frames, zero-padded to P): [NB][2][Z][N]
        npd = lib.ptts_dense_planes_bytes(N, P)
        dpl = cls._scratch('dpl', Z * npd, dev)
        call('ptts_split3_frame_windows', ptr(dy), B, T, N, NS, S, 0, P, S, ptr(dpl), npd, stream(), tag=('dy', Z))
        DYh = cls._scratch('DYh', NB * 2 * Z * N * 4, dev)
        call('ptts_dense_bf16x6_batched', ptr(c['D']), 0, ptr(dpl), npd, None, ptr(DYh), N, Z, c['R'], N, P, P, Z * N, 3, stream(),
             tag=('dft_dy', Z, c['R'], N, P))
        DYt = cls._scratch('DYt', NB * N * 2 * Z * 4, dev)                                         # [NB][N][2Z] fp32
        call('ptts_transpose_batched', ptr(DYh), ptr(DYt), NB, 2 * Z, N, stream())
        Gt = cls._scratch('Gt', NB * N * 2 * Kh * 4, dev)                                          # [NB][N][2 Kh] fp32 = (Gr | -Gi)^T
        call('ptts_dense_bf16x6_batched', ptr(DYt), N * 2 * Z, ptr(cls.xw_planes), npx, None, ptr(Gt), N * 2 * Kh, NB, N, 2 * Kh, 2 * Z,
             2 * Z, 2 * Kh, 1 if _Flags.bf16_products else 3, stream(), tag=('corr', NB, N, 2 * Kh, 2 * Z))
        dw = torch.empty((KW, Cin, N), dtype=torch.float32, device=dev)
        if KW in cls.KWS:
            ws = _workspace(lib.ptts_conv1d_freq_wgrad_inverse_workspace_bytes(KW, Cin, N), dev)
            call('ptts_conv1d_freq_wgrad_inverse', ptr(Gt), ptr(c['T2f']), ptr(dw), ptr(ws), ws.numel(), NB, c['TP'], KW, Cin, N, Kh, stream(),
                 tag=(NB, KW, Cin, N))
        else:
            out2 = torch.empty(2 * KW * N * 2 * Kh, dtype=torch.float32, device=dev)
            gemm_raw(c['T2'], Gt[:NB * N * 2 * Kh * 4].view(torch.float32), out2, 2 * KW, N * 2 * Kh, NB, lda=c['NBp'], ldb=N * 2 * Kh, ldc=N * 2 * Kh)
            call('ptts_conv1d_freq_wgrad_combine', ptr(out2), ptr(dw), KW, Cin, N, Kh, stream())
        return dw

    @classmethod
    def clear(cls):
        cls.x_src = cls.x_key = cls.x_hat = None
        cls.xw_src = cls.xw_key = cls.xw_planes = None
        # (the planes of frozen networks' kernels stay valid: the owner of a graph that reads them keeps them current, refresh_planes)
        cls.w_hat = {k: (e if id(getattr(e[0], '_ptts_flat', None)) in cls.frozen else (e[0], None, None, e[3], e[4], e[5]) + tuple(e[6:]))
                     for k, e in cls.w_hat.items()}


def conv1d_fft(on):
    """Context Conv1D forward in the frequency domain (see _C1FFT): True / False, None = the default (PTTS_CONV1D_FFT)."""
    _C1FFT.enabled = _C1FFT.default if on is None else bool(on)
    _C1FFT.clear()


def conv1d_split(on):
    """Switch the bf16x6 split product of the context Conv1D forward on or off (see _C1Split); None: the default."""
    _C1Split.enabled = _C1Split.default if on is None else bool(on)
    _C1Split.clear()


class Conv1dFn(torch.autograd.Function):
    """y[b,t,:] = b + sum_k a[b,t+k-pl,:].w[k];  `a` (already activated) is given, 'same' zero padding.
    pre = (saved input, padded?, product) computed earlier for exactly these operands (see _C1Cache).
    The split (bf16x6) kernels read their own planes of `a`: no zero-padded copy of the input is made for them (it was a
    61 MB fill + copy per call at BASELINE size); the fp32 GEMM paths still take the padded frame buffer."""
    @staticmethod
    def forward(ctx, a, w, b, pre=None):
        f32c(a, 'conv1d.a'); f32c(w, 'conv1d.w')
        B, T, Cin = a.shape
        KW, Ci2, N = w.shape
        assert Ci2 == Cin
        pl = (KW - 1) // 2
        if pre is not None:
            saved, padded, y = pre
        else:
            y = torch.empty((B, T, N), dtype=torch.float32, device=a.device)
            if _C1FFT.enabled and _C1Split.enabled and (not _Flags.bf16_products or _C1FFT.bf16_one_product) and _C1FFT.eligible(a, w):
                _C1FFT.forward(a, w, b, y)
                saved, padded = a, False
            elif _C1Split.enabled and not _Flags.deterministic and _C1Split.eligible(a, w):
                xp, Cp = _C1Split.frames(a, pl, KW - 1 - pl)
                wp = _C1Split.kernel(w)
                call('ptts_conv1d_bf16x6', ptr(xp[0]), ptr(xp[1]), ptr(xp[2]), ptr(wp[0]), ptr(wp[1]), ptr(wp[2]), ptr(b), ptr(y),
                     B, T, KW, Cp, N, stream(), tag=(B, T, KW, Cp, N))
                saved, padded = a, False
            else:
                saved, padded = _pad_time(a, pl, KW - 1 - pl), True
                gemm_raw(saved, w, y, B * T, N, KW * Cin, lda=Cin, rows_per_seg=T, seg_stride=(T + KW - 1) * Cin, bias=b)
            if _C1Cache.capture:
                _C1Cache.ap = (saved, padded)
        ctx.save_for_backward(saved, w)
        ctx.padded = padded
        ctx.x_src = a
        ctx.has_b = b is not None
        ctx.dims = (B, T, Cin, KW, N, pl)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        saved, w = ctx.saved_tensors
        B, T, Cin, KW, N, pl = ctx.dims
        dy = dy.contiguous()
        da = dw = db = None
        need_b = ctx.has_b and ctx.needs_input_grad[2] and not _Flags.skip_param_grads
        want_w = ctx.needs_input_grad[1] and not _Flags.skip_param_grads

        def padded_frames():
            return saved if ctx.padded else _pad_time(saved, pl, KW - 1 - pl)

        if want_w and _C1FFT.enabled and _C1FFT.wgrad_enabled and _C1Split.enabled and (not _Flags.bf16_products or _C1FFT.bf16_one_product) and dy.is_cuda and \
                _C1FFT.has_x(ctx.x_src, KW) and not ctx.padded and (B * (T // _C1FFT.segment(T, KW))) % 2 == 0:      # (2 B NS is a reduction length: a multiple of 4)
            # in the frequency domain, from the transform of the input the forward left behind (ops._C1FFT.wgrad)
            dw = _C1FFT.wgrad(ctx.x_src, dy, KW)
        elif want_w and _C1Split.enabled and not _Flags.deterministic and dy.is_cuda and _C1Split.eligible_wgrad(KW, N):
            # the weight gradient as a bf16x6 split product over frame-major planes (csrc/split.hip)
            xt, Crows, Pp = _C1Split.frames_t(ctx.x_src, saved, ctx.padded, B, T, Cin, KW)
            yt, _ = _C1Split.transposed(dy, B, T, N, 0, T + KW - 1, Pp)
            dw = torch.empty_like(w)
            call('ptts_conv1d_wgrad_bf16x6', ptr(xt[0]), ptr(xt[1]), ptr(xt[2]), ptr(yt[0]), ptr(yt[1]), ptr(yt[2]), ptr(dw),
                 B, T, KW, Cin, N, Crows, Pp, stream(), tag=(B, T, KW, Cin, N))
        elif want_w and _C1WgradT.enabled and not _Flags.deterministic and dy.is_cuda \
                and _C1Split.eligible_wgrad(KW, N) and B * T >= 2048:
            # exact fp32 over frame-major operands (csrc/conv1d_wgrad.hip); the bias gradient comes with it
            xt, Crows, Pp = _C1WgradT.frames_t(ctx.x_src, padded_frames(), KW)
            yt = torch.empty((N, Pp), dtype=torch.float32, device=dy.device)
            call('ptts_transpose_frames', ptr(dy), ptr(yt), B, T, N, 0, T + KW - 1, N, Pp, stream(), tag=(B, T, N))
            dw = torch.empty_like(w)
            if need_b:
                db = torch.empty(N, dtype=torch.float32, device=dy.device)
            call('ptts_conv1d_wgrad_t', ptr(xt), ptr(yt), ptr(dw), ptr(db), B, T, KW, Cin, N, Crows, Pp, stream(),
                 tag=(B, T, KW, Cin, N))
        elif want_w:
            dw = torch.empty_like(w)
            if need_b and N > 4:          # bias gradient taken from the B tiles of the weight-gradient product
                db = torch.empty(N, dtype=torch.float32, device=dy.device)
            gemm_raw(padded_frames(), dy, dw, KW * Cin, N, B * T, transA=1, lda=Cin, rows_per_seg=T,
                     seg_stride=(T + KW - 1) * Cin, colsum_b=db)
        if need_b and db is None:
            db = _colsum_f32(dy.view(B * T, N))
        if ctx.needs_input_grad[0]:
            # da[b,t,:] = sum_j dyp[b,t+j,:] . wf[j],  wf[j,n,ci] = w[KW-1-j,ci,n],  dyp padded (KW-1-pl, pl)
            dyp = _pad_time(dy, KW - 1 - pl, pl)
            wf = w.flip(0).permute(0, 2, 1).contiguous()
            da = torch.empty((B, T, Cin), dtype=torch.float32, device=dy.device)
            gemm_raw(dyp, wf, da, B * T, Cin, KW * N, lda=N, rows_per_seg=T, seg_stride=(T + KW - 1) * N)
        return da, dw, db, None


def conv1d(v, w, b=None):
    a = as_tensor(v)
    c = _C1Cache
    flat = getattr(w, '_ptts_flat', None) if c.enabled else None
    if flat is not None:
        key = (a.data_ptr(), a._version, tuple(a.shape), w.data_ptr(), w._version, flat.epoch,
               None if b is None else (b.data_ptr(), b._version), _hip.stream_id())
        if not torch.is_grad_enabled():
            c.capture = True
            try:
                y = Conv1dFn.apply(a, w, b)
            finally:
                c.capture = False
            c.key, c.y = key, y
            return y
        if c.key == key and c.y is not None and c.ap is not None:
            return Conv1dFn.apply(a, w, b, (c.ap[0], c.ap[1], c.y))
    return Conv1dFn.apply(a, w, b)


# ----------------------------------------------------------------------------------------------
# BatchNormalization(axis=-1)  (networktts.py:61,118,124,132) -> per-channel affine for the consumer
# ----------------------------------------------------------------------------------------------
BN_EPS, BN_MOMENTUM = 1e-3, 0.99   # Keras defaults


class _SyncBN(object):
    """SyncBN option of the data-parallel step (cfg.train_sync_batchnorm; SURVEY 8e note 1): the per-channel sums of a
    BatchNormalization layer (sum x, sum x^2 forward; the two gradient sums backward) are all-reduced over the ranks, so
    that W ranks with B samples each normalise like one process with W*B.  Off by default: per-rank statistics."""
    world = 1


def sync_batchnorm(world):
    """world > 1: synchronise BatchNorm statistics over that many ranks (torch.distributed must be initialised); 1: off."""
    _SyncBN.world = int(world) if world else 1


def _allreduce_small(t):
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


class BatchNormTrainFn(torch.autograd.Function):
    """(z, scale, shift) from the batch statistics of z [.., C]; updates the moving statistics in place.  z is handed THROUGH the
    node (the consumer applies scale / shift to this output, not to the node's input): the consumer's gradient w.r.t. z then
    arrives here and the statistics' share c2 z + c0 is added to it in the same pass (ptts_axpby_cols) -- as two separate
    contributions autograd added them with one more pass over the map per BatchNorm layer."""
    @staticmethod
    def forward(ctx, z, gamma, beta, moving_mean, moving_var, update_moving, unbiased_moving):
        f32c(z, 'bn.z')
        C = z.shape[-1]
        rows = z.numel() // C
        dev = z.device
        scale = torch.empty(C, dtype=torch.float32, device=dev)
        shift = torch.empty(C, dtype=torch.float32, device=dev)
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        rstd = torch.empty(C, dtype=torch.float32, device=dev)
        ctx.sync = _SyncBN.world
        # inside deferred_weight_grads(): dgamma / dbeta are added straight into the parameters' gradient buffers by the kernel that
        # computes them (two AccumulateGrad add launches per BatchNorm layer less: 30 per generator step)
        ctx.gt = (grad_target(gamma), grad_target(beta)) if _Deferred.active else None
        part = getattr(z, '_ptts_bn_partials', None)
        if part is not None and ctx.sync <= 1 and part[2] == rows and part[0].shape[1] == 2 * C:
            # the convolution that produced z summed what it stored (_BNStats): only the finish is left
            call('ptts_bn_finalize_partials', ptr(part[0]), part[1], rows, C, ptr(gamma), ptr(beta), ptr(moving_mean), ptr(moving_var),
                 BN_EPS, BN_MOMENTUM, 1 if update_moving else 0, 1 if unbiased_moving else 0, ptr(scale), ptr(shift), ptr(mean), ptr(rstd),
                 stream(), tag=(rows, C))
            ctx.save_for_backward(z, gamma, mean, rstd)
            return z.view_as(z), scale, shift
        if ctx.sync <= 1 and z.data_ptr() % 16 == 0 and _hip.lib().ptts_bn_batch_stats_supported(rows, C):
            # the conv stacks' few-channel maps: statistics and affine in one launch
            ws = _workspace(_hip.lib().ptts_colstats_workspace_bytes(rows, C), dev)
            call('ptts_bn_batch_stats', ptr(z), rows, C, ptr(gamma), ptr(beta), ptr(moving_mean), ptr(moving_var), BN_EPS, BN_MOMENTUM,
                 1 if update_moving else 0, 1 if unbiased_moving else 0, ptr(scale), ptr(shift), ptr(mean), ptr(rstd),
                 ptr(ws), ws.numel(), ptr(_stream_counter(dev)), stream(), tag=(rows, C))
            ctx.save_for_backward(z, gamma, mean, rstd)
            return z.view_as(z), scale, shift
        sums = colsums(z.view(rows, C))
        if ctx.sync > 1:          # the statistics of the global batch: one tiny all-reduce (2C doubles)
            _allreduce_small(sums)
            rows = rows * ctx.sync
        call('ptts_bn_finalize', ptr(sums), rows, ptr(gamma), ptr(beta), ptr(moving_mean), ptr(moving_var),
             BN_EPS, BN_MOMENTUM, 1, 1 if update_moving else 0, 1 if unbiased_moving else 0, C,
             ptr(scale), ptr(shift), ptr(mean), ptr(rstd), stream())
        ctx.save_for_backward(z, gamma, mean, rstd)
        return z.view_as(z), scale, shift

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dz_through, dscale, dshift):
        z, gamma, mean, rstd = ctx.saved_tensors
        C = z.shape[-1]
        rows = z.numel() // C
        dev = z.device
        dscale = dscale.contiguous() if dscale is not None else torch.zeros(C, dtype=torch.float32, device=dev)
        dshift = dshift.contiguous() if dshift is not None else torch.zeros(C, dtype=torch.float32, device=dev)
        c0 = torch.empty(C, dtype=torch.float32, device=dev)
        c2 = torch.empty(C, dtype=torch.float32, device=dev)
        gt = ctx.gt
        direct = gt is not None and gt[0] is not None and gt[1] is not None and _Deferred.active and not _Flags.deterministic and \
            ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
        if direct:
            call('ptts_bn_bwd_coefs_acc', ptr(dscale), ptr(dshift), ptr(mean), ptr(rstd), ptr(gamma), rows, C,
                 ptr(gt[0]), ptr(gt[1]), ptr(c0), ptr(c2), stream())
            cur = torch.cuda.current_stream()
            if all(cur.cuda_stream != q.cuda_stream for q in _Deferred.streams):
                _Deferred.streams.append(cur)             # flush_weight_grads() joins this stream before the optimiser reads the buffer
            dgamma = dbeta = None
        else:
            dgamma = torch.empty(C, dtype=torch.float32, device=dev)
            dbeta = torch.empty(C, dtype=torch.float32, device=dev)
            call('ptts_bn_bwd_coefs', ptr(dscale), ptr(dshift), ptr(mean), ptr(rstd), ptr(gamma), rows, C,
                 ptr(dgamma), ptr(dbeta), ptr(c0), ptr(c2), stream())
        if ctx.sync > 1:
            # the terms through the statistics carry every rank's loss: their coefficients come from the all-reduced sums
            # over the global row count; dgamma / dbeta stay this rank's own share (the flat-gradient all-reduce adds them)
            both = torch.cat([dscale, dshift])
            _allreduce_small(both)
            scratch = torch.empty(2 * C, dtype=torch.float32, device=dev)
            call('ptts_bn_bwd_coefs', ptr(both[:C]), ptr(both[C:]), ptr(mean), ptr(rstd), ptr(gamma), rows * ctx.sync, C,
                 ptr(scratch[:C]), ptr(scratch[C:]), ptr(c0), ptr(c2), stream())
        dz = None
        if ctx.needs_input_grad[0]:
            dz = torch.empty_like(z)
            if dz_through is not None:
                dz_through = f32c(dz_through.contiguous(), 'bn.dz')
            call('ptts_axpby_cols', ptr(dz_through), None, ptr(z), ptr(c2), ptr(c0), ptr(dz), rows, C, stream())
        return dz, dgamma, dbeta, None, None, None, None


def batchnorm_affine(z, gamma, beta, moving_mean, moving_var, training, update_moving=True, unbiased_moving=False):
    """Returns (z', scale, shift) such that BN(z) = scale*z' + shift; z' is z handed through the autograd node (training) or z itself."""
    if training:
        return BatchNormTrainFn.apply(z, gamma, beta, moving_mean, moving_var, update_moving, unbiased_moving)
    C = z.shape[-1]
    scale = torch.empty(C, dtype=torch.float32, device=z.device)
    shift = torch.empty(C, dtype=torch.float32, device=z.device)
    call('ptts_bn_finalize', None, 1, ptr(gamma), ptr(beta), ptr(moving_mean), ptr(moving_var), BN_EPS, BN_MOMENTUM,
         0, 0, 0, C, ptr(scale), ptr(shift), None, None, stream())
    return z, scale, shift


# ----------------------------------------------------------------------------------------------
# materialised activation
# ----------------------------------------------------------------------------------------------
class AffineActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift, act, alpha):
        f32c(x, 'affine_act.x')
        C = x.shape[-1]
        rows = x.numel() // C
        y = torch.empty_like(x)
        call('ptts_affine_act', ptr(x), ptr(scale), ptr(shift), ptr(y), rows, C, act, alpha, stream())
        ctx.save_for_backward(x, y, scale, shift)
        ctx.cfg = (act, alpha)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, y, scale, shift = ctx.saved_tensors
        act, alpha = ctx.cfg
        dx, dscale, dshift = _affine_act_bwd_raw(dy.contiguous(), x, y, scale, shift, act, alpha,
                                                 want_dx=ctx.needs_input_grad[0])
        if scale is None:
            dscale = dshift = None
        return dx, dscale, dshift, None, None


def affine_act(x, scale=None, shift=None, act=None, alpha=0.3):
    return AffineActFn.apply(x.contiguous(), scale, shift, ACT_CODES[act], alpha)


# ----------------------------------------------------------------------------------------------
# gated product of a gated convolution (networktts.py:128-134: Conv2D(...) * Conv2D(..., activation=sigmoid))
# ----------------------------------------------------------------------------------------------
class GatedMulFn(torch.autograd.Function):
    """y = a * sigmoid(b) on the two pre-activations (one pass; neither the gate nor the product's operands are stored)."""
    @staticmethod
    def forward(ctx, a, b):
        f32c(a, 'gated_mul.a'); f32c(b, 'gated_mul.b')
        assert a.shape == b.shape
        y = torch.empty_like(a)
        call('ptts_gated_mul_fwd', ptr(a), ptr(b), ptr(y), a.numel(), stream(), tag=(a.numel(),))
        ctx.save_for_backward(a, b)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        dy = dy.contiguous()
        da, db = torch.empty_like(a), torch.empty_like(b)
        call('ptts_gated_mul_bwd', ptr(dy), ptr(a), ptr(b), ptr(da), ptr(db), a.numel(), stream(), tag=(a.numel(),))
        return da, db


def gated_mul(a, b):
    return GatedMulFn.apply(as_tensor(a).contiguous(), as_tensor(b).contiguous())


# ----------------------------------------------------------------------------------------------
# LSTM / Bidirectional LSTM (networktts.py:72-96)
# ----------------------------------------------------------------------------------------------
lstm_trace = None        # a list: (tag, HIP event on the recurrence's stream) around the step chains (tools/gen_timeline.py events)


def _lstm_mark(tag):
    if lstm_trace is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        lstm_trace.append((tag, ev))


class _GradGateFn(torch.autograd.Function):
    """Identity whose only purpose is its place in the autograd tape (creation order = priority of the backward pass, stream of
    creation = stream its input buffer accumulates on)."""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g


def grad_gate(x):
    return _GradGateFn.apply(x)


class _GradInjectFn(torch.autograd.Function):
    """Identity in the forward pass; in the backward pass the gradient that another, EARLIER backward pass left in `holder`
    (holder['grad'], made on another stream: holder['event']) is added to the incoming one.  The wait for that stream happens here,
    when the engine reaches this node -- not before the whole backward pass."""
    @staticmethod
    def forward(ctx, x, holder):
        ctx.holder = holder
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        h = ctx.holder
        extra = h.get('grad')
        if extra is None:
            return g, None
        if h.get('event') is not None:
            torch.cuda.current_stream().wait_event(h['event'])
        h['grad'] = None
        return (extra if g is None else g + extra), None


def grad_inject(x, holder):
    return _GradInjectFn.apply(x, holder)


class _LSTMRing(object):
    """Address-stable buffers for the recurrences when they are replayed as hipGraphs (ptts_set_lstm_graph / PTTS_LSTM_GRAPH=1: the
    graphs are keyed on the operands' addresses, and buffers taken from the caching allocator come back at other addresses in a third of
    the calls).  Two sets per shape and stream, used in turn; a set is handed out again only when every view given out of it is dead
    (weak references) -- otherwise fresh memory is used, which merely costs a capture."""
    enabled = os.environ.get('PTTS_LSTM_GRAPH', '0') == '1'
    slots = {}

    @classmethod
    def take(cls, key, shapes, dev):
        import weakref
        ring = cls.slots.setdefault(key, {'i': 0, 'sets': [None, None]})
        i = ring['i']
        ring['i'] = 1 - i
        st = ring['sets'][i]
        n = [int(torch.Size(sh).numel()) for sh in shapes]
        if st is not None and any(r() is not None for r in st['refs']):
            return [torch.empty(sh, dtype=torch.float32, device=dev) for sh in shapes]       # still in use: not this time
        if st is None:
            st = ring['sets'][i] = {'buf': torch.empty(sum(n), dtype=torch.float32, device=dev), 'refs': []}
        out, off = [], 0
        for sh, k in zip(shapes, n):
            out.append(st['buf'][off:off + k].view(sh))
            off += k
        st['refs'] = [weakref.ref(t) for t in out]
        return out


def lstm_launch(x, W, U, b, reverse=False):
    """The forward launches of LSTMFn (input projection + the T-step recurrence) with no autograd node: (h, c, gates)."""
    f32c(x, 'lstm.x'); f32c(W); f32c(U); f32c(b)
    B, T, In = x.shape
    ndir, H, G4 = U.shape
    assert G4 == 4 * H and W.shape == (In, ndir * G4)
    dev = x.device
    if _LSTMRing.enabled:
        xproj, h, c, gates = _LSTMRing.take(('fwd', B, T, ndir, H, _hip.stream_id()),
                                            [(B, T, ndir * G4), (B, T, ndir * H), (B, T, ndir * H), (B, T, ndir * G4)], dev)
    else:
        xproj = torch.empty((B, T, ndir * G4), dtype=torch.float32, device=dev)
        h = torch.empty((B, T, ndir * H), dtype=torch.float32, device=dev)
        c = torch.empty((B, T, ndir * H), dtype=torch.float32, device=dev)
        gates = torch.empty((B, T, ndir * G4), dtype=torch.float32, device=dev)
    gemm_raw(x, W, xproj, B * T, ndir * G4, In, bias=b)
    wsf = _workspace(_hip.lib().ptts_lstm_fwd_workspace_bytes(B, T, H, ndir), dev)
    _lstm_mark('fwd0')
    call('ptts_lstm_fwd', ptr(xproj), ptr(U), ptr(h), ptr(gates), ptr(c), ptr(wsf), wsf.numel(), B, T, H, ndir, int(reverse), stream())
    _lstm_mark('fwd1')
    return h, c, gates


lstm_dx_ready = None      # (data_ptr of the last LSTM backward's dx, event recorded right behind its product)


class LSTMFn(torch.autograd.Function):
    """x [B,T,In]; W [In, ndir*4H]; U [ndir,H,4H]; b [ndir*4H] -> h [B,T,ndir*H] (Keras gate order i,f,c,o)."""
    @staticmethod
    def forward(ctx, x, W, U, b, reverse, pre=None):
        # `pre`: (h, c, gates) of lstm_launch() on the same operands -- the launches went out earlier, this call only ties the result
        # into the autograd tape (layers.Model._run: a node created late has its backward chain enqueued early)
        h, c, gates = lstm_launch(x, W, U, b, reverse) if pre is None else pre
        ctx.save_for_backward(x, W, U, h, c, gates)
        ctx.reverse = int(reverse)
        # inside deferred_weight_grads(): the weight gradients are added straight into the flat gradient buffer on the stream of the
        # backward chain.  Handed to autograd instead, their AccumulateGrad nodes (made on the main stream) make the MAIN stream wait
        # for this side stream at the moment the engine reaches them -- with the chain enqueued first (Model.side_backward_first) that
        # is before the critic's backward pass, which then runs after the recurrence instead of under it.
        ctx.gt = (grad_target(W), grad_target(U), grad_target(b)) if _Deferred.active else None
        return h

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dh):
        x, W, U, h, c, gates = ctx.saved_tensors
        B, T, In = x.shape
        ndir, H, G4 = U.shape
        dev = x.device
        dh = dh.contiguous()
        if _LSTMRing.enabled:
            dh_s, dgates = _LSTMRing.take(('bwd', B, T, ndir, H, _hip.stream_id()), [tuple(dh.shape), tuple(gates.shape)], dev)
            dh_s.copy_(dh)
            dh = dh_s
        else:
            dgates = torch.empty_like(gates)
        nws = _hip.lib().ptts_lstm_bwd_workspace_bytes(B, T, H, ndir)
        ws = _workspace(nws, dev)
        _lstm_mark('bwd0')
        call('ptts_lstm_bwd', ptr(dh), ptr(U), ptr(gates), ptr(c), ptr(dgates), ptr(ws), ws.numel(),
             B, T, H, ndir, ctx.reverse, stream())
        _lstm_mark('bwd1')
        M = B * T
        dx = dW = dU = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm_raw(dgates, W, dx, M, In, ndir * G4, transB=1, ldb=ndir * G4)
            # dx is what the rest of the backward pass waits for; the weight-gradient products below (0.5 ms at [64,400], H = 256) are
            # not.  The point on this stream where dx is complete is published: a caller that runs this node on a side stream
            # (optimizertts_wgan.generator_forward_early) lets the main stream wait for THIS event instead of for the whole node.
            global lstm_dx_ready
            lstm_dx_ready = (dx.data_ptr(), torch.cuda.current_stream().record_event())
        if ctx.needs_input_grad[1]:
            dW = torch.empty_like(W)
            gemm_raw(x, dgates, dW, In, ndir * G4, M, transA=1, lda=In, rows_per_seg=M)
        if ctx.needs_input_grad[3]:
            db = _colsum_f32(dgates.view(M, ndir * G4))
        if ctx.needs_input_grad[2]:
            # dU[d] = sum_t h_prev[d]^T . dgates[d]; h_prev is h shifted by one step in the walking direction
            hprev = torch.zeros_like(h)
            for d in range(ndir):
                rev = (d == 1) if ndir == 2 else bool(ctx.reverse)
                sl = slice(d * H, (d + 1) * H)
                if T > 1:
                    if rev:
                        hprev[:, :-1, sl].copy_(h[:, 1:, sl])
                    else:
                        hprev[:, 1:, sl].copy_(h[:, :-1, sl])
            dU = torch.empty_like(U)
            for d in range(ndir):
                gemm_raw(hprev.view(M, ndir * H)[:, d * H:], dgates.view(M, ndir * G4)[:, d * G4:], dU[d],
                         H, G4, M, transA=1, lda=ndir * H, rows_per_seg=M, ldb=ndir * G4)
        gt = ctx.gt
        if gt is not None and _Deferred.active and not _Flags.deterministic and \
                all(t is not None or g is None for t, g in zip(gt, (dW, dU, db))):
            for t, g in zip(gt, (dW, dU, db)):
                if g is not None:
                    t.add_(g.view(t.shape))
            cur = torch.cuda.current_stream()
            if all(cur.cuda_stream != q.cuda_stream for q in _Deferred.streams):
                _Deferred.streams.append(cur)             # flush_weight_grads() joins this stream before the optimiser reads the buffer
            dW = dU = db = None
        return dx, dW, dU, db, None, None


def lstm(v, W, U, b, reverse=False, pre=None):
    return LSTMFn.apply(as_tensor(v).contiguous(), W, U, b, reverse, pre)


# ----------------------------------------------------------------------------------------------
# WGAN-GP pieces (optimizertts_wgan.py:44-79)
# ----------------------------------------------------------------------------------------------
def gp_interpolate(real, fake, alpha_b, out=None):
    """RandomWeightedAverage: alpha_b [B] per-sample weights (optimizertts_wgan.py:44-51)."""
    f32c(real, 'gp.real'); f32c(fake, 'gp.fake'); f32c(alpha_b, 'gp.alpha')
    assert real.shape == fake.shape and alpha_b.numel() == real.shape[0]
    if out is None:
        out = torch.empty_like(real)
    else:
        f32c(out, 'gp.out'); assert out.shape == real.shape
    B = real.shape[0]
    call('ptts_gp_interpolate', ptr(real), ptr(fake), ptr(alpha_b), ptr(out), B, real.numel() // B, stream())
    return out


class GradPenaltyFn(torch.autograd.Function):
    """mean_b (1 - ||g_b||_2)^2 (gradient_penalty_loss, optimizertts_wgan.py:53-68)."""
    @staticmethod
    def forward(ctx, g):
        f32c(g, 'gp.g')
        B = g.shape[0]
        TD = g.numel() // B
        sq = torch.empty(B, dtype=torch.float32, device=g.device)
        pen = torch.empty(1, dtype=torch.float32, device=g.device)
        coef = torch.empty(B, dtype=torch.float32, device=g.device)
        call('ptts_gp_sqnorm', ptr(g), ptr(sq), B, TD, stream())
        call('ptts_gp_penalty', ptr(sq), ptr(pen), ptr(coef), B, stream())
        ctx.save_for_backward(g, coef)
        return pen.view(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, up):
        g, coef = ctx.saved_tensors
        B = g.shape[0]
        dg = torch.empty_like(g)
        up = up.contiguous().view(1)
        call('ptts_gp_scale_rows', ptr(g), ptr(coef), ptr(up), ptr(dg), B, g.numel() // B, stream())
        return dg


def grad_penalty(g):
    return GradPenaltyFn.apply(g.contiguous())


class MeanScaledFn(torch.autograd.Function):
    """sign * mean(v): wasserstein_loss with a constant +-1 target (optimizertts_wgan.py:70-71,152-153)."""
    @staticmethod
    def forward(ctx, v, sign):
        f32c(v, 'wasserstein.v')
        out = torch.empty(1, dtype=torch.float32, device=v.device)
        call('ptts_mean_scaled', ptr(v), v.numel(), float(sign), ptr(out), stream())
        ctx.sign, ctx.shape = float(sign), v.shape
        return out.view(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, up):
        n = 1
        for s in ctx.shape:
            n *= s
        return (up * (ctx.sign / n)).expand(ctx.shape).contiguous(), None


class WassersteinPairFn(torch.autograd.Function):
    """(-mean(v[:B]), +mean(v[B:])) of the stacked critic output v [2B, T, 1]: the two Wasserstein terms of the critic loss
    (optimizertts_wgan.py:152-153) off ONE tensor -- sliced with torch, the backward of the two slices is two zero-filled
    [2B,T,1] tensors, two copies and an add; here it is the two halves of one gradient tensor, written once."""
    @staticmethod
    def forward(ctx, v, B):
        f32c(v, 'wasserstein.v')
        n = v.numel()
        h = n // v.shape[0] * B
        out = torch.empty(2, dtype=torch.float32, device=v.device)
        call('ptts_mean_scaled', ptr(v), h, -1.0, ptr(out), stream())
        call('ptts_mean_scaled', ptr(_ptr_off(v, h)), n - h, 1.0, ptr(_ptr_off(out, 1)), stream())
        ctx.cfg = (tuple(v.shape), B, h, n)
        return out[0], out[1]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, up_v, up_f):
        shape, B, h, n = ctx.cfg
        dv = torch.empty(shape, dtype=torch.float32, device=(up_v if up_v is not None else up_f).device)
        flat = dv.view(-1)
        if up_v is None: flat[:h].zero_()
        else: flat[:h].copy_((up_v * (-1.0 / h)).expand(h))
        if up_f is None: flat[h:].zero_()
        else: flat[h:].copy_((up_f * (1.0 / (n - h))).expand(n - h))
        return dv, None


def wasserstein_pair(v, B):
    return WassersteinPairFn.apply(v.contiguous(), int(B))


def wasserstein(v, sign):
    return MeanScaledFn.apply(v.contiguous(), sign)


class WLSEFn(torch.autograd.Function):
    """mean((y - yhat)^2 * w[d]) (specweighted_lse_loss :73-79; w=None gives lse_loss optimizertts.py:56-57)."""
    @staticmethod
    def forward(ctx, yhat, y, w):
        f32c(yhat, 'wlse.yhat'); f32c(y, 'wlse.y'); f32c(w)
        assert yhat.shape == y.shape
        D = y.shape[-1]
        rows = y.numel() // D
        out = torch.empty(1, dtype=torch.float32, device=y.device)
        call('ptts_wlse_fwd', ptr(y), ptr(yhat), ptr(w), ptr(out), rows, D, stream())
        ctx.save_for_backward(yhat, y, w)
        return out.view(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, up):
        yhat, y, w = ctx.saved_tensors
        D = y.shape[-1]
        rows = y.numel() // D
        d = torch.empty_like(yhat)
        up = up.contiguous().view(1)
        call('ptts_wlse_bwd', ptr(y), ptr(yhat), ptr(w), ptr(up), ptr(d), rows, D, stream())
        return d, None, None


def wlse(yhat, y, w=None):
    return WLSEFn.apply(yhat.contiguous(), y.contiguous(), w)


# ----------------------------------------------------------------------------------------------
# optimiser-side kernels
# ----------------------------------------------------------------------------------------------
def weight_clip_(p, lo, hi):
    """In-place clamp of a flat parameter buffer (north_star extra; no reference counterpart)."""
    f32c(p, 'clip.p')
    call('ptts_weight_clip', ptr(p), p.numel(), float(lo), float(hi), stream())
    return p


def adam_keras_step_(p, g, m, v, step, lr, b1, b2, eps, gscale=1.0):
    """Keras-2.2 Adam on flat buffers; `step` is a device int32 scalar incremented by the kernel."""
    for t in (p, g, m, v):
        f32c(t, 'adam')
    assert p.numel() == g.numel() == m.numel() == v.numel()
    assert step.is_cuda and step.dtype == torch.int32
    call('ptts_adam_keras_step', ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), float(b1), float(b2),
         float(eps), float(gscale), ptr(step), stream())

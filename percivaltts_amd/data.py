"""Host-side batch loading and cost helpers (numpy only) -- the callers' side of the hot path.

Same function names, arguments and results as the reference's percivaltts/data.py:34-406 for what the
training driver uses: `path:(shape)` selectors (:34-76), load (:96-133), croplen (:143-171),
croplen_weight (:173-231), batching with random-shift windows (:234-284), load_inoutset (:297-322) and
the cost helpers (:327-406).  Files are headerless float32.
"""
from __future__ import print_function

import copy
import os
import re
import time

import numpy as np

from .percivaltts import is_int, print_tty

_SEL = re.compile(r'(.*):\((.*)\)')


def getpath(path):
    m = _SEL.findall(path)
    return m[0][0] if m else path


def getpathandshape(path, shape=None):
    """'/dir/*.spec:(-1,60)' -> ('/dir/*.spec', (-1, 60)); a non-integer selector names a file whose length is used."""
    m = _SEL.findall(path)
    if not m:
        return path, shape
    path = m[0][0]
    sel = ()
    for s in m[0][1].split(','):
        if is_int(s):
            sel += (int(s),)
        else:
            sel += (np.fromfile(os.path.join(os.path.dirname(path), s), dtype=np.float32).shape[0],)
    if shape is not None:
        print('WARNING: shape has been set both as argument ({}) of the getpathandshape(.) function and in the file '
              'selector ({}) (at the end of path); The one set as argument will be used: {}'.format(shape, sel, shape))
        return path, shape
    return path, sel


def getlastdim(path):
    _, size = getpathandshape(path)
    return 1 if size is None else size[-1]


def _read(fpath, shape):
    if not os.path.isfile(fpath):
        raise ValueError('{} does not exists'.format(fpath))
    X = np.fromfile(fpath, dtype='float32')
    if shape is not None:
        X = X.reshape(shape)
    if np.isnan(X).any(): raise ValueError('ERROR: There are nan in {}'.format(fpath))
    if np.isinf(X).any(): raise ValueError('ERROR: There are inf in {}'.format(fpath))
    return X


def loadfile(fpath, fbase=None, shape=None):
    if fbase is not None:
        fpath = fpath.replace('*', fbase)
    fpath, shape = getpathandshape(fpath, shape)
    return _read(fpath, shape)


def load(dirpath, fbases, shape=None, frameshift=0.005, verbose=0, label=''):
    """List of matrices, one per file id."""
    dirpath, shape = getpathandshape(dirpath, shape)
    Xs, totlen, memsize = [], 0, 0.0
    for n, fbase in enumerate(fbases):
        if verbose > 0:
            print_tty('\r    {}Loading file {}/{} {}: ({:.2f}% done)        '.format(label, 1 + n, len(fbases), fbase, 100.0 * n / len(fbases)))
        X = _read(dirpath.replace('*', fbase), shape)
        Xs.append(X)
        totlen += X.shape[0]
        memsize += X.size * 4 / float(1024 ** 2)
    if verbose > 0:
        print_tty('\r                                                                 \r')
        print('    {}{} sentences, frames={} ({}), {} MB                     '.format(
            label, len(fbases), totlen, time.strftime('%H:%M:%S', time.gmtime(totlen * frameshift)), memsize))
    return Xs


def gettotallen(Xs, axis=0):
    return sum(x.shape[axis] for x in Xs)


def croplen(xs, axis=0):
    """Crop, sample by sample, every list of `xs` to the shortest length found among the lists (in place)."""
    if axis > 2:
        raise ValueError('Do not manage axis values bigger than 2')
    if len(set(len(x) for x in xs)) > 1:
        raise ValueError('the size of the data sets are not identical ({})'.format([len(x) for x in xs]))
    for ki in range(len(xs[0])):
        siz = min(x[ki].shape[axis] for x in xs)
        idx = [slice(None)] * axis + [slice(0, siz)]
        for x in xs:
            x[ki] = x[ki][tuple(idx)]
    return xs


def croplen_weight(xs, w, thresh=0.5, cropmode='begend', cropsize=int(0.750 / 0.005)):
    """Drop frames whose weight is below `thresh`: at both ends ('begend'), also inside when the silent gap is longer
    than `cropsize` frames ('begendbigger'), or everywhere ('all')."""
    if len(set([len(w)] + [len(x) for x in xs])) > 1:
        raise ValueError('the size of the data sets are not identical ({})'.format([len(x) for x in xs]))
    for ki in range(len(w)):
        wk = w[ki][:, 0] if w[ki].ndim > 1 else w[ki]
        keep = wk > thresh
        if cropmode == 'begend':
            on = np.where(keep)[0]
            sel = slice(int(on.min()), int(on.max()))
        elif cropmode == 'begendbigger':
            on = np.where(keep)[0]
            gaps = np.diff(on)
            for gi in np.where(gaps > 1)[0]:
                if gaps[gi] < int(cropsize):
                    keep[on[gi]:on[gi + 1]] = True
            sel = np.where(keep)[0]
        elif cropmode == 'all':
            sel = np.where(keep)[0]
        else:
            raise ValueError('unknown cropmode ' + str(cropmode))
        for x in xs:
            x[ki] = x[ki][sel,]
        w[ki] = w[ki][sel,]
    return xs, w


def batching(xs, length=None, lengthmax=None, padtype='randshift', outmask=False, rand=None):
    """Stack 2-D matrices into [B, length, feat] batches.  'randshift' takes a random window of `length` frames from
    every sample (np.random.randint, so the numpy seed makes it repeatable); 'padright' zero-pads on the right.
    `rand` (optional, one uniform number in [0, 1) per sample): the shift of sample b is floor(rand[b] * (number of possible
    shifts)) instead of a fresh np.random.randint draw -- what data parallelism uses, so that a rank that loads only its
    shard of the files takes the same windows as the process that loads the whole batch (load_inoutset)."""
    if len(set(len(x) for x in xs)) > 1:
        raise ValueError('the size of the data sets are not identical ({})'.format([len(x) for x in xs]))
    nb = len(xs[0])
    if length is None:
        lens = [xs[0][b].shape[0] for b in range(nb)]
        length = max(lens) if padtype == 'padright' else min(lens)
    if lengthmax is not None and length > lengthmax:
        length = lengthmax
    xbs = []
    for x in xs:
        feat = 1 if x[0].ndim == 1 else x[0].shape[1]
        xbs.append(np.zeros((len(x), length, feat), dtype='float32'))
    MB = np.zeros((nb, length), dtype='float32') if outmask else None
    shift = 0
    for b in range(nb):
        samplelen = xs[0][b].shape[0]
        minlen = min(samplelen, length)
        if padtype == 'randshift':
            if rand is not None:
                shift = min(int(rand[b] * ((samplelen - length) + 1)), samplelen - length)
            else:
                shift = np.random.randint(0, (samplelen - length) + 1)
        for xi, x in enumerate(xs):
            seg = x[b][shift:shift + minlen]
            xbs[xi][b, :minlen, :] = seg.reshape(minlen, -1)
        if outmask: MB[b, :minlen] = 1
    return xbs, MB


def addstop(X, value=1.0):
    X = copy.deepcopy(X)
    stop = np.zeros(X[0].shape[1] + 1)
    stop[-1] = value
    for xi in range(len(X)):
        X[xi] = np.vstack((np.concatenate((X[xi], np.zeros((X[xi].shape[0], 1))), axis=1), stop))
    return X


def _kept_len(wk, thresh, cropmode, cropsize):
    """Number of frames croplen_weight keeps of a weight track (same selection rules, no data moved)."""
    keep = wk > thresh
    if cropmode == 'begend':
        on = np.where(keep)[0]
        return int(on.max()) - int(on.min())
    if cropmode == 'begendbigger':
        on = np.where(keep)[0]
        gaps = np.diff(on)
        for gi in np.where(gaps > 1)[0]:
            if gaps[gi] < int(cropsize):
                keep[on[gi]:on[gi + 1]] = True
        return int(keep.sum())
    if cropmode == 'all':
        return int(keep.sum())
    raise ValueError('unknown cropmode ' + str(cropmode))


def batch_window_length(indir, outdir, outwdir, fid_lst, length=None, lengthmax=None, maskpadtype='padright', cropmode='begend',
                        thresh=0.5, cropsize=int(0.750 / 0.005)):
    """The window length T load_inoutset would choose for the batch `fid_lst` (reference data.py:297-322 -> croplen,
    croplen_weight, batching: with length=None it is the min ('randshift') or max ('padright') of the cropped sample lengths,
    clipped by lengthmax), found WITHOUT reading the wide files: the frame counts of the inputs and outputs come from the file
    sizes, only the one-column time-weight files are read.  Data parallelism needs it before sharding: a rank that loads only its
    shard must window it to the GLOBAL batch's T -- the shard's own min / max can differ, and with it every random shift."""
    if length is not None:
        return int(min(length, lengthmax)) if lengthmax is not None else int(length)
    lens = []
    ipath, ishape = getpathandshape(indir)
    opath, oshape = getpathandshape(outdir)
    wpath, wshape = getpathandshape(outwdir)
    def frames(path, shape, fid):
        f = path.replace('*', fid)
        if not os.path.isfile(f):
            raise ValueError('{} does not exists'.format(f))
        dim = 1
        if shape is not None:
            for d in shape[1:]: dim *= int(d)
        return os.path.getsize(f) // (4 * dim)
    for fid in fid_lst:
        w = _read(wpath.replace('*', fid), wshape)
        n = min(frames(ipath, ishape, fid), frames(opath, oshape, fid), w.shape[0])
        wk = (w[:, 0] if w.ndim > 1 else w)[:n]
        lens.append(_kept_len(wk, thresh, cropmode, cropsize))
    T = max(lens) if maskpadtype == 'padright' else min(lens)
    if lengthmax is not None and T > lengthmax:
        T = lengthmax
    return int(T)


def load_inoutset(indir, outdir, outwdir, fid_lst, inouttimesync=True, length=None, lengthmax=None,
                  maskpadtype='padright', cropmode='begend', verbose=0, rand=None):
    """Load one batch of inputs, outputs and time weights, cropped and windowed: X [B,T,ctx], Y [B,T,out], W [B,T,1]
    (reference data.py:297-322).  `rand`: see batching -- the per-sample uniforms of the random window shifts, drawn by the
    caller for the WHOLE global batch so that every rank can load its shard of `fid_lst` alone."""
    X = load(indir, fid_lst, verbose=verbose, label='Context labels: ')
    Y = load(outdir, fid_lst, verbose=verbose, label='Output features: ')
    W = load(outwdir, fid_lst, verbose=verbose, label='Time weights: ')
    if inouttimesync:
        X, Y, W = croplen([X, Y, W])
        [X, Y], W = croplen_weight([X, Y], W, cropmode=cropmode)
        [X, Y, W], _ = batching([X, Y, W], length=length, lengthmax=lengthmax, padtype=maskpadtype, rand=rand)
    else:
        X = addstop(X)
        Y, W = croplen([Y, W])
        [Y], W = croplen_weight([Y], W, cropmode=cropmode)
        Y = addstop(Y)
        [X], _ = batching([X], length=length, lengthmax=lengthmax, padtype=maskpadtype)
        [Y], _ = batching([Y], length=length, lengthmax=lengthmax, padtype=maskpadtype)
    return X, Y, W


# ---- evaluation helpers ----------------------------------------------------------------------------------
def cost_0pred_rmse(Y_val):
    """RMSE of the all-zero prediction (the worst predictor)."""
    if isinstance(Y_val, list):
        return float(np.sqrt(sum(np.sum(y ** 2) for y in Y_val) / float(sum(y.size for y in Y_val))))
    return float(np.sqrt(np.mean(Y_val ** 2)))


def _one_by_one(Xs):
    for xi in range(len(Xs[0])):
        yield xi, [np.reshape(inp[xi], [1] + list(inp[xi].shape)) for inp in Xs]


def cost_model_mfn(fn, Xs):
    """Average of fn over the samples, one utterance at a time."""
    if not isinstance(Xs[0], list):
        return 0.0
    cost = 0.0
    for _, ins in _one_by_one(Xs):
        cost += fn(*ins)
    return cost / len(Xs[0])


def cost_model_prediction_rmse(mod, Xs, Y_val, inouttimesync=True):
    if not isinstance(Xs[0], list):
        return 0.0
    cost, nbel = 0.0, 0
    for xi, ins in _one_by_one(Xs):
        ypred = mod.predict(*ins)
        cost += np.sum((Y_val[xi] - ypred[0,]) ** 2)
        nbel += ypred[0,].size
    return float(np.sqrt(cost / nbel))


def prediction_mstd(mod, Xs):
    if not isinstance(Xs[0], list):
        return 0.0
    s = 0.0
    for _, ins in _one_by_one(Xs):
        s += np.std(mod.predict(*ins)[0,])
    return s / len(Xs[0])


def prediction_rms(mod, Xs):
    if not isinstance(Xs[0], list):
        return 0.0
    s, nbel = 0.0, 0
    for _, ins in _one_by_one(Xs):
        ypred = mod.predict(*ins)
        s += np.sum(ypred[0,] ** 2)
        nbel += ypred[0,].size
    return float(np.sqrt(s / nbel))


class BatchPrefetcher(object):
    """Double-buffered batch pipeline in front of `train_on_batch`.

    The reference loads the files of a batch, builds the numpy arrays and only then trains on them, every batch
    (optimizertts.py:230-232), so a sub-15 ms device step would wait on numpy I/O and on a synchronous host-to-device
    copy.  Here a worker thread calls `make_batch(i)` (any function returning a tuple of float32 numpy arrays, e.g.
    `load_inoutset` of the i-th list of file ids) for i = 0..n-1 and issues the host-to-device copies on a dedicated
    HIP copy stream; the consumer iterates over device tensors and its stream waits on the copy's event, not on the
    host.  `depth` batches are in flight (2 = double buffering).  stage_pinned=True first copies each array into a
    reusable pinned buffer (a slot is rewritten only after its DMA completed) -- useful when the loader can fill such
    a buffer directly; for numpy arrays that already exist it measured slower than letting the runtime stage them.

    device=None (or a CPU device) gives the same iterator without pinning or streams (host tests).
    """

    _END = object()

    def __init__(self, make_batch, n, device=None, depth=2, stage_pinned=False):
        import threading
        try:
            import queue
        except ImportError:      # pragma: no cover
            import Queue as queue
        import torch
        self._torch = torch
        self._make, self._n, self._depth = make_batch, int(n), max(1, int(depth))
        self._device = torch.device(device) if device is not None else torch.device('cpu')
        self._cuda = self._device.type == 'cuda'
        if self._cuda and self._device.index is None:
            self._device = torch.device('cuda', torch.cuda.current_device())
        self._stage = bool(stage_pinned)
        self._q = queue.Queue(maxsize=self._depth)
        self._slots = [None] * (self._depth + 1)      # pinned buffers (+1: one is being filled while `depth` are queued)
        self._events = [None] * (self._depth + 1)
        self._copy_stream = torch.cuda.Stream(device=self._device) if self._cuda else None
        self._stop = False
        self.load_seconds = 0.0
        self._thread = threading.Thread(target=self._work, name='ptts-prefetch')
        self._thread.daemon = True
        self._thread.start()

    def _pinned(self, slot, k, shape):
        torch = self._torch
        bufs = self._slots[slot]
        if bufs is None:
            bufs = self._slots[slot] = {}
        need = int(np.prod(shape))
        b = bufs.get(k)
        if b is None or b.numel() < need:
            b = bufs[k] = torch.empty(need, dtype=torch.float32).pin_memory()
        return b[:need].view(*shape)

    def _work(self):
        torch = self._torch
        try:
            if self._cuda:
                torch.cuda.set_device(self._device)
            for i in range(self._n):
                if self._stop:
                    break
                t0 = time.time()
                arrays = self._make(i)
                self.load_seconds += time.time() - t0
                if not isinstance(arrays, (tuple, list)):
                    arrays = (arrays,)
                if not self._cuda:
                    item = tuple(torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)) for a in arrays)
                    self._q.put((item, None))
                    continue
                slot = i % len(self._slots)
                outs = []
                with torch.cuda.stream(self._copy_stream):
                    for k, a in enumerate(arrays):
                        a = np.ascontiguousarray(a, dtype=np.float32)
                        d = torch.empty(a.shape, dtype=torch.float32, device=self._device)
                        if self._stage:
                            if self._events[slot] is not None:
                                self._events[slot].synchronize()      # the copy that last read this pinned slot is done
                            pin = self._pinned(slot, k, a.shape)
                            np.copyto(pin.numpy(), a)                 # single-threaded memcpy, interpreter lock released
                            d.copy_(pin, non_blocking=True)
                        else:
                            # pageable source: the HIP runtime stages it through its own pinned chunks, pipelined with the
                            # DMA (measured 1.1 ms for 61 MB, against 10 ms for a host-side copy into a pinned buffer plus the
                            # DMA); the call blocks this loader thread only
                            d.copy_(torch.from_numpy(a))
                        outs.append(d)
                    ev = self._copy_stream.record_event()
                self._events[slot] = ev
                self._q.put((tuple(outs), ev))
            self._q.put((self._END, None))
        except BaseException as e:       # hand the failure to the consumer instead of dying silently
            self._q.put((e, None))

    def __iter__(self):
        return self

    def __next__(self):
        item, ev = self._q.get()
        if item is self._END:
            self._q.put((self._END, None))
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        if ev is not None:
            cur = self._torch.cuda.current_stream(self._device)
            cur.wait_event(ev)
            for t in item:
                t.record_stream(cur)       # allocated on the copy stream, consumed here
        return item

    next = __next__

    def close(self):
        self._stop = True
        try:
            while True:
                self._q.get_nowait()
        except Exception:
            pass
        self._thread.join(timeout=10.0)

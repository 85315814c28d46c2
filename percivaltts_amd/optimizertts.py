"""Training driver: epoch/batch loop, checkpoints, resume, multi-trial hyper search.

API and behaviour follow the reference's percivaltts/optimizertts.py:60-436 (OptimizerTTS): configuration
defaults (:65-101), saveTrainingState/loadTrainingState (:103-145), train_oneparamset (:149-303),
randomize_hyper (:306-322), train (:324-372) and the overridable hooks default_options / prepare /
train_on_batch / update_validation_cost / save|loadTrainingStateLossSpecific (:378-436), whose default
implementation is least-squares (Adam on lse_loss).  The arithmetic runs on HIP kernels (ops.py).
Cost/sample plots (percivaltts.py:299-381) are logging utilities outside the hot-path scope: the
`train_log_plot` switch is kept and ignored.
"""
from __future__ import print_function

import copy
import glob
import os
import pickle
import sys
import time
from collections import defaultdict

import numpy as np
import torch

from . import backend_hip
from . import data
from . import ops
from . import parallel
from .optim import KerasAdam
from .percivaltts import configuration, print_log, print_tty, time2str, proc_memresident


def lse_loss(y_true, y_pred):
    """mean((y_true - y_pred)^2) on device tensors (optimizertts.py:56-57)."""
    return ops.wlse(y_pred, y_true, None)


def _with_next(it):
    """(index, item, next item or None) over an iterator: one item of look-ahead."""
    it = iter(it)
    try:
        cur = next(it)
    except StopIteration:
        return
    i = 0
    for nx in it:
        yield i, cur, nx
        cur = nx
        i += 1
    yield i, cur, None


class OptimizerTTS:

    _model = None    # the model whose parameters are optimised
    _errtype = 'LSE'

    def __init__(self, cfgtomerge, model, errtype='LSE', **kwargs):
        self._model = model
        self._errtype = errtype

        cfg = configuration()
        # defaults common to every optimisation scheme (optimizertts.py:71-92)
        cfg.train_min_nbepochs = 200
        cfg.train_max_nbepochs = 300
        cfg.train_nbepochs_scalewdata = True
        cfg.train_cancel_nodecepochs = 50
        cfg.train_cancel_validthresh = 10.0
        cfg.train_batch_size = 5
        cfg.train_batch_padtype = 'randshift'
        cfg.train_batch_cropmode = 'begendbigger'
        cfg.train_batch_length = None
        cfg.train_batch_lengthmax = None
        cfg.train_nbtrials = 1
        cfg.train_hypers = []
        cfg = self.default_options(cfg)
        cfg.train_log_plot = True
        if cfgtomerge is not None: cfg.merge(cfgtomerge)
        for k, v in kwargs.items(): setattr(cfg, k, v)
        self.cfg = cfg

        print('Training configuration')
        self.cfg.print_content()

    # ---- training state ------------------------------------------------------------------------------------
    def saveTrainingState(self, fstate, extras=None, printfn=print):
        if extras is None: extras = dict()
        printfn('    saving training state in {} ...'.format(fstate), end='')
        sys.stdout.flush()
        self.saveTrainingStateLossSpecific(fstate)
        # beyond the reference's numpy state: torch's generators (the interpolation weights of the gradient penalty are
        # drawn on the device), so that a resumed run continues the SAME random sequence
        extras = dict(extras)
        extras['torch_rng_cpu'] = torch.get_rng_state().numpy()
        if torch.cuda.is_available():
            extras['torch_rng_cuda'] = torch.cuda.get_rng_state().cpu().numpy()
        with open(fstate + '.model.cfgextras.pkl', 'wb') as f:
            pickle.dump([self.cfg, extras, np.random.get_state()], f)
        print(' done')
        sys.stdout.flush()

    def loadTrainingState(self, fstate, printfn=print):
        printfn('    reloading parameters from {} ...'.format(fstate), end='')
        sys.stdout.flush()
        self.loadTrainingStateLossSpecific(fstate)
        with open(fstate + '.model.cfgextras.pkl', 'rb') as f:
            DATA = pickle.load(f)
        print(' done')
        sys.stdout.flush()
        saved = DATA[0]
        if self.cfg.__dict__ != saved.__dict__:
            printfn('        configurations are not the same!')
            for attr, val in self.cfg.__dict__.items():
                if attr not in saved.__dict__:
                    print('            attribute {}: is not in the saved configuration state'.format(attr))
                elif val != saved.__dict__[attr]:
                    print('            attribute {}: new state {}, saved state {}'.format(attr, val, saved.__dict__[attr]))
            for attr in saved.__dict__:
                if attr not in self.cfg.__dict__:
                    print('            attribute {}: is not in the new configuration state'.format(attr))
        return DATA

    # ---- one training run ----------------------------------------------------------------------------------
    def train_oneparamset(self, indir, outdir, wdir, fid_lst_tra, fid_lst_val, params_savefile, trialstr='', cont=None):
        stem = os.path.splitext(params_savefile)[0]

        print('Loading all validation data at once ...')
        X_vals = data.load(indir, fid_lst_val, verbose=1, label='Context labels: ')
        Y_vals = data.load(outdir, fid_lst_val, verbose=1, label='Output features: ')
        X_vals, Y_vals = data.croplen([X_vals, Y_vals])
        print('    {} validation files'.format(len(fid_lst_val)))
        print('    number of validation files / train files: {:.2f}%'.format(100.0 * float(len(fid_lst_val)) / len(fid_lst_tra)))

        print('Model initial status before training')
        worst_val = data.cost_0pred_rmse(Y_vals)
        print("    0-pred validation RMSE = {} (100%)".format(worst_val))
        print('    initial RMS of prediction = {}'.format(data.prediction_rms(self._model, [X_vals])))
        init_val = data.cost_model_prediction_rmse(self._model, [X_vals], Y_vals)
        best_val = None
        print("    initial validation RMSE = {} ({:.4f}%)".format(init_val, 100.0 * init_val / worst_val))

        nbbatches = int(len(fid_lst_tra) / self.cfg.train_batch_size)
        print('    using {} batches of {} sentences each'.format(nbbatches, self.cfg.train_batch_size))
        print('    model #parameters={}'.format(self._model.count_params()))

        nbtrainframes = sum(data.loadfile(outdir, fid).shape[0] for fid in fid_lst_tra)
        print('    Training set: {} sentences, #frames={} ({})'.format(
            len(fid_lst_tra), nbtrainframes, time.strftime('%H:%M:%S', time.gmtime(nbtrainframes * self._model.vocoder.shift))))
        print('    #parameters/#frames={:.2f}'.format(float(self._model.count_params()) / nbtrainframes))
        if self.cfg.train_nbepochs_scalewdata and self.cfg.train_batch_lengthmax is not None:
            # an epoch only sees train_batch_lengthmax frames per sentence: rescale the epoch counts accordingly
            epochcoef = nbtrainframes / float(self.cfg.train_batch_lengthmax * len(fid_lst_tra))
            print('    scale number of epochs wrt number of frames')
            self.cfg.train_min_nbepochs = int(self.cfg.train_min_nbepochs * epochcoef)
            self.cfg.train_max_nbepochs = int(self.cfg.train_max_nbepochs * epochcoef)
            print('        train_min_nbepochs={}'.format(self.cfg.train_min_nbepochs))
            print('        train_max_nbepochs={}'.format(self.cfg.train_max_nbepochs))

        self.prepare()

        costs = defaultdict(list)
        epochs_modelssaved, epochs_durs = [], []
        nbnodecepochs, generator_updates, epochstart = 0, 0, 1
        if cont and len(glob.glob(stem + '-trainingstate-last.h5*')) > 0:
            print('    reloading previous training state ...')
            savedcfg, extras, rngstate = self.loadTrainingState(stem + '-trainingstate-last.h5')
            np.random.set_state(rngstate)
            if 'torch_rng_cpu' in extras:
                torch.set_rng_state(torch.as_tensor(extras['torch_rng_cpu'], dtype=torch.uint8))
            if 'torch_rng_cuda' in extras and torch.cuda.is_available():
                torch.cuda.set_rng_state(torch.as_tensor(extras['torch_rng_cuda'], dtype=torch.uint8))
            costs = extras['costs']
            epochs_modelssaved = extras['epochs_modelssaved']
            epochs_durs = extras['epochs_durs']
            generator_updates = extras['generator_updates']
            if hasattr(self, 'generator_updates'):
                self.generator_updates = generator_updates      # the critic_runs schedule continues where it stopped
            epochstart = extras['epoch'] + 1
            same = all(getattr(savedcfg, k) == getattr(self.cfg, k)
                       for k in ('train_min_nbepochs', 'train_max_nbepochs', 'train_cancel_nodecepochs'))
            if same:
                best_val = extras['best_val']
                nbnodecepochs = extras['nbnodecepochs']

        print_log("    start training ...")
        epoch = -1
        for epoch in range(epochstart, 1 + self.cfg.train_max_nbepochs):
            timeepochstart = time.time()
            rndidx = np.arange(int(nbbatches * self.cfg.train_batch_size))   # restart from the ordered state: repeatable after a reload
            np.random.shuffle(rndidx)
            rndidxb = np.split(rndidx, nbbatches)
            costs_tra_batches, load_times, train_times = [], [], []
            world, rank = getattr(self, 'world', 1), getattr(self, 'rank', 0)
            # data parallelism: a rank reads only the files of ITS shard of the global batch.  The random
            # window shifts are drawn for the whole batch, up front and in batch order (every rank holds the same numpy RNG
            # state), as uniforms that load_inoutset turns into shifts: W ranks take the windows one process would take.
            shift_rand = [np.random.random_sample(int(self.cfg.train_batch_size)) for _ in range(nbbatches)] if world > 1 else None
            def make_batch(batchid, rndidxb=rndidxb, shift_rand=shift_rand):
                ids = rndidxb[batchid]
                rnd = None
                length = self.cfg.train_batch_length
                if world > 1:
                    # the window length is a property of the GLOBAL batch (min / max of its cropped sample lengths when
                    # train_batch_length is None, the default): fixed before sharding from the file sizes and the one-column
                    # weight files, so that every rank windows its shard to the same T and draws the shifts one process would
                    length = data.batch_window_length(indir, outdir, wdir, [fid_lst_tra[bidx] for bidx in ids], length=length,
                                                      lengthmax=self.cfg.train_batch_lengthmax, maskpadtype=self.cfg.train_batch_padtype,
                                                      cropmode=self.cfg.train_batch_cropmode)
                    lo, hi = parallel.shard_batch(len(ids), world, rank)
                    ids, rnd = ids[lo:hi], shift_rand[batchid][lo:hi]
                fid_lst_trab = [fid_lst_tra[bidx] for bidx in ids]
                X_trab, Y_trab, W_trab = data.load_inoutset(
                    indir, outdir, wdir, fid_lst_trab, length=length,
                    lengthmax=self.cfg.train_batch_lengthmax, maskpadtype=self.cfg.train_batch_padtype,
                    cropmode=self.cfg.train_batch_cropmode, rand=rnd)
                return X_trab, Y_trab                        # already this rank's shard: only it was read and crosses PCIe

            # batches are loaded, pinned and copied to the device two ahead of the step that consumes them
            prefetch = data.BatchPrefetcher(make_batch, nbbatches, device=self.device, depth=2)
            for batchid, (X_trab, Y_trab), nxt in _with_next(self._closing(prefetch)):
                t0 = time.time()
                self.hint_next_batch(*(nxt if nxt is not None else (None, None)))     # (the prefetcher has it on the device already)
                print_tty('\r    Training batch {}/{}'.format(1 + batchid, nbbatches))
                load_times.append(prefetch.load_seconds - sum(load_times))
                print_tty(' (iter load: {:.6f}s); training '.format(load_times[-1]))

                t1 = time.time()
                cost_tra = self.train_on_batch(batchid, X_trab, Y_trab)
                train_times.append(time.time() - t1)

                if cost_tra is not None:
                    print_tty('err={:.4f} (iter train: {:.4f}s)                  '.format(cost_tra, train_times[-1]))
                    if np.isnan(cost_tra):
                        print_log('    previous costs: {}'.format(costs_tra_batches))
                        print_log('    E{} Batch {}/{} train cost = {}'.format(epoch, 1 + batchid, nbbatches, cost_tra))
                        raise ValueError('ERROR: Training cost is nan!')
                    costs_tra_batches.append(cost_tra)
            print_tty('\r                                                           \r')
            costs['model_training'].append(np.mean(costs_tra_batches) if costs_tra_batches else float('nan'))

            # data parallelism: every rank has accumulated BatchNorm moving statistics from its own shards; validation and
            # the checkpoints use their mean over the ranks, so that all ranks validate (and rank 0 saves) the same model
            self._sync_moving_statistics()
            cost_val = self.update_validation_cost(costs, X_vals, Y_vals)

            print_log("    E{}/{} {}  cost_tra={:.6f} (load:{}s train:{}s)  cost_val={:.6f} ({:.4f}% RMSE)  {} MiB GPU {} MiB RAM".format(
                epoch, self.cfg.train_max_nbepochs, trialstr, costs['model_training'][-1], time2str(np.sum(load_times)),
                time2str(np.sum(train_times)), cost_val, 100 * costs['model_rmse_validation'][-1] / worst_val,
                backend_hip.gpu_memused(), proc_memresident()))
            sys.stdout.flush()

            if np.isnan(cost_val): raise ValueError('ERROR: Validation cost is nan!')

            main = parallel.is_main()                   # files are written by rank 0 only (the replicas are identical)
            if main:
                self._model.save(stem + '-last.h5', printfn=print_log, extras={'cost_val': cost_val})

            if epoch >= self.cfg.train_min_nbepochs:   # no model is trusted before train_min_nbepochs
                if (best_val is None) or (cost_val < best_val):
                    best_val = cost_val
                    if main:
                        self._model.save(params_savefile, printfn=print_log, extras={'cost_val': cost_val},
                                         infostr='(E{} C{:.4f})'.format(epoch, best_val))
                    epochs_modelssaved.append(epoch)
                    nbnodecepochs = 0
                else:
                    nbnodecepochs += 1

            epochs_durs.append(time.time() - timeepochstart)
            med = np.median(epochs_durs[-10:])
            print_log('    ET: {}   max TT: {}s   train ~time left: {}'.format(
                time2str(epochs_durs[-1]), time2str(med * self.cfg.train_max_nbepochs),
                time2str(med * (self.cfg.train_max_nbepochs - epoch))))

            if main:
                self.saveTrainingState(stem + '-trainingstate-last.h5', printfn=print_log, extras={
                    'cost_val': cost_val, 'best_val': best_val, 'costs': costs, 'epochs_modelssaved': epochs_modelssaved,
                    'epochs_durs': epochs_durs, 'nbnodecepochs': nbnodecepochs,
                    'generator_updates': getattr(self, 'generator_updates', generator_updates), 'epoch': epoch})
            parallel.barrier()                          # nobody runs ahead (or resumes) before the files are complete

            if nbnodecepochs >= self.cfg.train_cancel_nodecepochs:
                print_log('WARNING: validation error did not decrease for {} epochs. Early stop!'.format(self.cfg.train_cancel_nodecepochs))
                break

        if best_val is None: raise ValueError('No model has been saved during training!')
        return {'epoch_stopped': epoch, 'worst_val': worst_val,
                'best_epoch': epochs_modelssaved[-1] if len(epochs_modelssaved) > 0 else -1, 'best_val': best_val}

    def _sync_moving_statistics(self):
        if parallel.world_size() > 1:
            parallel.average_buffers_([b for b in self._model.kerasmodel.buffers()])

    @classmethod
    def randomize_hyper(cls, cfg):
        """Draw each (name, lo, hi) of cfg.train_hypers uniformly (integers when both bounds are ints) into a COPY of cfg."""
        cfg = copy.copy(cfg)
        if len(cfg.train_hypers) < 1: return cfg, ''
        parts = []
        for hyper in cfg.train_hypers:
            if isinstance(hyper[1], int) and isinstance(hyper[2], int):
                setattr(cfg, hyper[0], np.random.randint(hyper[1], hyper[2]))
            else:
                setattr(cfg, hyper[0], np.random.uniform(hyper[1], hyper[2]))
            parts.append(hyper[0] + '=' + str(getattr(cfg, hyper[0])))
        return cfg, ','.join(parts)

    def train(self, indir, outdir, wdir, fid_lst_tra, fid_lst_val, params_savefile, cont=None):
        stem = os.path.splitext(params_savefile)[0]
        if self.cfg.train_nbtrials > 1:
            if parallel.is_main():
                self._model.save(stem + '-init.h5', printfn=print_log)
            parallel.barrier()
        try:
            trials = []
            for triali in range(1, 1 + self.cfg.train_nbtrials):
                print('\nStart trial {} ...'.format(triali))
                train_rets, cfg = None, self.cfg
                try:
                    trialstr = 'trial' + str(triali)
                    if len(self.cfg.train_hypers) > 0:
                        cfg, hyperstr = self.randomize_hyper(self.cfg)
                        trialstr += ',' + hyperstr
                        print('    randomized hyper-parameters: ' + trialstr)
                    if self.cfg.train_nbtrials > 1:
                        self._model.load(stem + '-init.h5')
                    t0 = time.time()
                    basecfg, self.cfg = self.cfg, cfg
                    try:
                        train_rets = self.train_oneparamset(indir, outdir, wdir, fid_lst_tra, fid_lst_val, params_savefile, trialstr=trialstr, cont=cont)
                    finally:
                        self.cfg = basecfg
                    cont = None
                    print_log('Total trial run time: {}s'.format(time2str(time.time() - t0)))
                except KeyboardInterrupt:
                    raise
                except ValueError:
                    if len(self.cfg.train_hypers) > 0:
                        print_log('WARNING: Training crashed!')
                        import traceback
                        traceback.print_exc()
                    else:
                        print_log('ERROR: Training crashed!')
                        raise
                if self.cfg.train_nbtrials > 1 and train_rets is not None:
                    line = [triali] + [getattr(cfg, f[0]) for f in self.cfg.train_hypers]
                    line += [train_rets[k] for k in sorted(train_rets.keys())]
                    header = 'trials ' + ' '.join(f[0] for f in self.cfg.train_hypers) + ' ' + ' '.join(sorted(train_rets.keys()))
                    trials.append(line)
                    if parallel.is_main():
                        np.savetxt(stem + '-trials.txt', np.vstack(trials), header=header)
        except KeyboardInterrupt:
            print_log('WARNING: Training interrupted by user!')
        print_log('Finished')

    # ---- hooks: least-squares training by default -----------------------------------------------------------
    def default_options(self, cfg):
        cfg.train_lse_learningrate_log10 = -3.39794   # 10**-3.39794 = 0.0004
        cfg.train_lse_adam_beta1 = 0.9
        cfg.train_lse_adam_beta2 = 0.999
        cfg.train_lse_adam_epsilon_log10 = -8
        return cfg

    def prepare(self):
        print('    Prepare LSE training')
        self.device = self._model.to_device()
        self.world, self.rank = parallel.init()
        self.opti = KerasAdam(self._model.kerasmodel, self.device, lr=10 ** self.cfg.train_lse_learningrate_log10,
                              beta_1=self.cfg.train_lse_adam_beta1, beta_2=self.cfg.train_lse_adam_beta2,
                              epsilon=10 ** self.cfg.train_lse_adam_epsilon_log10)
        print('    optimizer: Adam')

    @staticmethod
    def _closing(prefetch):
        """Iterate a BatchPrefetcher and stop its worker thread whatever ends the loop (NaN guard, KeyboardInterrupt)."""
        try:
            for item in prefetch:
                yield item
        finally:
            prefetch.close()

    def hint_next_batch(self, X_next, Y_next):
        """The training loop names the batch the next train_on_batch will get (or None); optimisers that can use it override this."""
        pass

    def _local_shard(self, X, Y):
        """Host (numpy) batches are the global batch: keep this rank's rows.  Device tensors come from
        data.BatchPrefetcher, whose loader already cut the shard before it crossed PCIe."""
        if getattr(self, 'world', 1) > 1 and not torch.is_tensor(X):
            lo, hi = parallel.shard_batch(X.shape[0], self.world, self.rank)
            return X[lo:hi], Y[lo:hi]
        return X, Y

    def _to_dev(self, a):
        if torch.is_tensor(a):            # already resident (data.BatchPrefetcher)
            return a.to(device=self.device, dtype=torch.float32)
        return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(self.device)

    def train_on_batch(self, batchid, X_trab, Y_trab):
        X_trab, Y_trab = self._local_shard(X_trab, Y_trab)
        X, Y = self._to_dev(X_trab), self._to_dev(Y_trab)
        self.opti.zero_grad()
        pred = self._model.kerasmodel(X, training=True)
        loss = lse_loss(Y, pred)
        loss.backward()
        self.opti.step(parallel.allreduce_sum_(self.opti.flat.grad))
        return float(np.sqrt(float(loss.item())))   # a cost related to the generator's error, whatever the loss type

    def update_validation_cost(self, costs, X_vals, Y_vals):
        costs['model_rmse_validation'].append(data.cost_model_prediction_rmse(self._model, [X_vals], Y_vals))
        return costs['model_rmse_validation'][-1]

    def saveTrainingStateLossSpecific(self, fstate):
        self.opti.save(fstate + '.optimizer.npz')
        np.savez(fstate + '.model.weights.npz', *self._model.kerasmodel.get_weights())

    def loadTrainingStateLossSpecific(self, fstate):
        self.opti.load(fstate + '.optimizer.npz')
        with np.load(fstate + '.model.weights.npz') as z:
            self._model.kerasmodel.set_weights([z['arr_{}'.format(i)] for i in range(len(z.files))])

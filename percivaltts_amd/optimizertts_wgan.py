"""WGAN-GP optimiser of the acoustic model -- THE hot path of this build.

Same class, hooks, configuration keys and step semantics as the reference's
percivaltts/optimizertts_wgan.py:82-328:

  critic step (:115-154, every batch):   L_D = mean(-D(y,x)) + mean(D(G(x),x)) + lambda*mean_b(1-||dD(x^,x)/dx^||_2)^2,
        x^ = a*y + (1-a)*G(x), a~U[0,1) per sample; G frozen; Adam(1e-4, .5, .9) on the critic.
  generator step (:157-213, when batchid % critic_runs == 0, critic_runs = 10|5 :225-228):
        WGAN:    L_G = mean(-D(G(x),x));     WLSWGAN: L_G = mean(w)*mean(-D(G(x),x)) + mean((y-G(x))^2*(1-w));
        D frozen; Adam(1e-3, .5, .9) on the generator.

What is different is where the work runs: every forward, backward and second-order sweep is a HIP kernel
reached through ops.py; the three critic evaluations share one evaluation of the context branch
(mathematically identical to the reference's three calls); in the critic step only the generator's spectral
branch is computed because the critic slices the spectrum out of its input (networks_critic.py:58) -- f0 and
noise-mask columns of G(x) cannot influence L_D (cfg.train_wgan_prune_dead_branches, on by default; results
are identical, the reference's TF graph cannot see through its concat+slice).  The whole device step can be
captured once and replayed as a hipGraph (cfg.train_wgan_hipgraph).  Under torch.distributed each rank takes its
shard of the minibatch and the flat gradient buffer of the network being updated is all-reduced before Adam.
"""
from __future__ import print_function

import numpy as np
import time

import torch

from . import backend_hip
from . import data
from . import layers as kl
from . import ops
from . import optimizertts
from . import parallel
from .backend_hip import nonlin_sigmoidparm
from .optim import KerasAdam


# ---- module-level losses, same names as the reference (:44-79), operating on device tensors ---------------
class RandomWeightedAverage(object):
    """x^ = a*real + (1-a)*fake with a ~ U[0,1) of shape [B,1,1] (:44-51); `alpha` can be injected for parity tests."""
    def __init__(self, batchsize=None):
        self.batchsize = batchsize

    def __call__(self, inputs, alpha=None):
        real, fake = inputs
        if alpha is None:
            alpha = torch.rand(real.shape[0], device=real.device, dtype=torch.float32)
        return ops.gp_interpolate(real.contiguous(), fake.contiguous(), alpha.reshape(-1).contiguous())


def gradient_penalty_loss(y_true, y_pred, averaged_samples):
    """mean_b (1 - ||d sum(y_pred) / d averaged_samples||_2)^2 (:53-68); differentiable w.r.t. the critic weights."""
    with ops.input_grad_only():
        g = torch.autograd.grad(y_pred, averaged_samples, grad_outputs=torch.ones_like(y_pred), create_graph=True)[0]
    return ops.grad_penalty(g)


def wasserstein_loss(valid_true, valid_pred):
    """mean(valid_true * valid_pred) for a constant target of +-1 (:70-71,152-153)."""
    return ops.wasserstein(valid_pred, float(valid_true))


def specweighted_lse_loss(y_true, y_pred, specweight):
    """mean((y_true - y_pred)^2 * specweight) (:73-79)"""
    return ops.wlse(y_pred, y_true, specweight)


def freq2fwspecidx(freq, fs, nbbnds):
    """Index of the first frequency-warped band whose centre lies above `freq`.

    Stand-in for sp.freq2fwspecidx of the pulsemodel submodule, which is absent from the reference checkout
    (optimizertts_wgan.py:192; SURVEY.md 8c: parity unpinned).  Documented warping of this build: band centres
    uniformly spaced on the mel scale mel(f) = 1127 ln(1 + f/700) from 0 to fs/2.  Override with
    cfg.train_wgan_critic_LSWGANtransidx."""
    mel = lambda f: 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)
    centres = 700.0 * (np.exp(np.linspace(0.0, mel(fs / 2.0), nbbnds) / 1127.0) - 1.0)
    above = np.where(centres > freq)[0]
    return int(above[0]) if len(above) > 0 else int(nbbnds - 1)


class _DeviceCosts(list):
    """costs_tra_critic_batches of the reference (a list of floats, :219) whose new entries stay on the device: a critic
    step then costs no host synchronisation; the floats are fetched -- all at once -- when the list is read."""
    def __init__(self):
        list.__init__(self)
        self._pending = []

    def append_device(self, t):
        # a copy, not a view: under cfg.train_wgan_hipgraph every replay returns the SAME static output tensor of the
        # captured graph, and pending views of it would all read the last batch's loss
        self._pending.append(t.detach().reshape(()).clone())

    def _sync(self):
        if self._pending:
            vals = torch.stack(self._pending).cpu().tolist()
            self._pending = []
            list.extend(self, [float(v) for v in vals])

    def __len__(self): self._sync(); return list.__len__(self)
    def __iter__(self): self._sync(); return list.__iter__(self)
    def __getitem__(self, i): self._sync(); return list.__getitem__(self, i)
    def __repr__(self): self._sync(); return list.__repr__(self)


class OptimizerTTSWGAN(optimizertts.OptimizerTTS):

    costs_tra_critic_batches = []
    generator_updates = 0

    def __init__(self, cfgtomerge, model, errtype='WGAN', critic=None, **kwargs):
        optimizertts.OptimizerTTS.__init__(self, cfgtomerge, model, errtype, **kwargs)
        self.critic = critic
        self.costs_tra_critic_batches = _DeviceCosts()
        self.generator_updates = 0

    def default_options(self, cfg):
        cfg.train_wgan_critic_learningrate_log10 = -4
        cfg.train_wgan_critic_adam_beta1 = 0.5
        cfg.train_wgan_critic_adam_beta2 = 0.9
        cfg.train_wgan_gen_learningrate_log10 = -3
        cfg.train_wgan_gen_adam_beta1 = 0.5
        cfg.train_wgan_gen_adam_beta2 = 0.9
        cfg.train_wgan_pg_lambda = 10
        cfg.train_wgan_LScoef = 0.25                  # if >0, mix LSE and WGAN losses
        cfg.train_wgan_validation_ltm_winlen = 20
        cfg.train_wgan_critic_LSWGANtransfreqcutoff = 4000
        cfg.train_wgan_critic_LSWGANtranscoef = 1.0 / 8.0
        cfg.train_wgan_critic_use_WGAN_incnoisefeature = False
        # build extensions (no reference counterpart)
        cfg.train_wgan_critic_LSWGANtransidx = None   # None: freq2fwspecidx(cutoff) of this build
        cfg.train_wgan_weight_clip = None             # c > 0: clamp critic weights to [-c, c] after each critic update
        cfg.train_wgan_prune_dead_branches = True
        cfg.train_wgan_hipgraph = False              # True: every step replayed as a hipGraph; 'auto': only for batches of at most train_wgan_hipgraph_maxframes frames (launch-bound steps such as the reference's B = 10, run.py:125-126)
        cfg.train_wgan_hipgraph_maxframes = 8192
        cfg.train_wgan_parallel_streams = False      # the critic evaluations on separate HIP streams
        cfg.train_wgan_side_backward_first = False   # generator step: the BLSTM's autograd node created last (its launches go out first), so that its backward chain is enqueued first.  Measured: the chain then ends 1 ms earlier, the step does not (the main stream's backward becomes the tail): off
        cfg.train_wgan_stack_real_fake = True        # critic(real) and critic(fake) as one stacked 2B pass (exact: no BatchNorm)
        cfg.train_wgan_graph_frozen_planes = True    # a replayed critic step reads the frozen generator's context-kernel planes from a buffer refreshed per generator update instead of rebuilding them in every replay
        cfg.train_wgan_fake_ahead = False            # (measured, off) the frozen generator's sample of the NEXT critic-only batch drawn on a side stream beside this batch's critic step: cycle 29.55 -> 29.85 ms -- the step's graph then transforms the context input a second time, and the overlap does not pay for it
        cfg.train_wgan_generator_lookahead = True    # a generator step's forward launched one batch ahead when the caller names the next batch (hint_next_batch / device_step(nxt=...))
        cfg.train_wgan_pair_forward = True           # critic step: the forward of the stacked real / fake batch (2B) and of x^ (B) as ONE launch per layer over 3B rows (their backward passes stay separate)
        cfg.train_wgan_ctx_stream = False            # critic step: the context branch on a side stream beside the spectral stacks (one fork / join per pass; measured, see DESIGN)
        cfg.train_wgan_feed_spectra = True           # the critic is fed at its spectral slice (real / fake / interpolated spectra built directly; False: whole 86-column samples through the slice, as the reference's graph does)
        cfg.train_wgan_reuse_ctx_conv = True         # generator step reuses the critic step's G-context-Conv1D product (same batch)
        cfg.train_wgan_early_critic = True           # generator step: critic starts on the spectral branch, BLSTM joins for the LS term
        cfg.train_wgan_hoist_side_backward = True    # ... and the BLSTM branch's BACKWARD too (its output is read by the least-squares term only: the branch is cut out of the tape, run on its own, its gradient injected at the cut).  Measured + 1 % (three A/B pairs, fp32 and bf16): both chains then run under the critic step
        cfg.train_wgan_batch_graph = False           # 'tune': a batch that trains both networks may be replayed as ONE hipGraph (BLSTM fork kept), if that times faster.  It does not: 29.6 ms against 14.0 for the separate steps (cross-stream edges of a graph replay at half speed on this runtime) -- off, so that the timing runs are not made either
        cfg.train_wgan_hoist_generator = True        # a batch that trains both: G's forward (it does not depend on the critic) is launched BEFORE the critic step -- its BLSTM chain runs under that step -- and the critic step takes its fake sample from it
        cfg.train_wgan_graph_critic = None           # 'on' / 'off': pin the critic step's form (hipGraph replay / eager launches) whatever train_wgan_hipgraph would choose
        cfg.train_wgan_graph_generator = None        # ... and the generator step's
        cfg.train_wgan_graph_split = False           # hipGraph of forward + backward only, update launched eagerly (what data parallelism uses; settable for tests)
        cfg.train_wgan_async_update = None           # all-reduce + Adam on a communication stream, overlapped with the next forward that does not need the weights (None: on when world > 1)
        cfg.train_sync_batchnorm = False             # data parallelism: BatchNorm statistics over all ranks (SyncBN) instead of per rank
        cfg.train_wgan_bf16_products = False         # BASELINE configs[2]: ONE bf16 product per position in the split GEMM kernels (context Conv1D, Dense, LSTM projections) instead of the six of the fp32 split; fp32 accumulation and master weights (ops.bf16_products)
        cfg.train_wgan_split_bf16 = None             # context Conv1D (forward + weight gradient) and Dense products as bf16x6 split products (fp32 arithmetic on the bf16 matrix cores: ops._C1Split, ops._DenseSplit); None: the defaults (on; PTTS_CONV1D_SPLIT=0 / PTTS_DENSE_SPLIT=0 turn them off), False: fp32 MFMA kernels
        return cfg

    # ---------------------------------------------------------------------------------------------------------
    def _wls_weights(self):
        """Per-feature least-squares weights and the WGAN term weight (:186-213)."""
        voc, cfg = self._model.vocoder, self.cfg
        transidx = cfg.train_wgan_critic_LSWGANtransidx

        def sig(n):
            c = transidx if transidx is not None else freq2fwspecidx(cfg.train_wgan_critic_LSWGANtransfreqcutoff, voc.fs, n)
            return nonlin_sigmoidparm(np.arange(n, dtype=np.float32), c, cfg.train_wgan_critic_LSWGANtranscoef)

        els = [np.zeros(1)]                                                           # f0
        els.append(np.ones(voc.specsize()) if cfg.train_wgan_LScoef == 0.0 else sig(voc.specsize()))
        if voc.noisesize() > 0 and cfg.train_wgan_critic_use_WGAN_incnoisefeature:
            els.append(np.ones(voc.noisesize()) if cfg.train_wgan_LScoef == 0.0 else sig(voc.noisesize()))
        else:
            els.append(np.zeros(voc.noisesize()))
        if voc.vuvsize() > 0:
            els.append(np.zeros(1))
        w = np.hstack(els) * (1.0 - cfg.train_wgan_LScoef)
        return (1.0 - w), float(np.mean(w))

    def prepare(self):
        print('    Prepare {} training...'.format(self._errtype))
        cfg = self.cfg
        self.device = dev = self._model.to_device()
        self.world, self.rank = parallel.init()

        generator = self._model.kerasmodel
        critic = self.critic.model
        critic.to(dev)
        print('    critic architecture:')
        critic.summary()
        self.critic_net = critic

        print('    compiling critic')
        self.critic_opti = KerasAdam(critic, dev, lr=10 ** cfg.train_wgan_critic_learningrate_log10,
                                     beta_1=cfg.train_wgan_critic_adam_beta1, beta_2=cfg.train_wgan_critic_adam_beta2, epsilon=1e-7)
        print('        optimizer: Adam')
        print('    compiling generator')
        self.gen_opti = KerasAdam(generator, dev, lr=10 ** cfg.train_wgan_gen_learningrate_log10,
                                  beta_1=cfg.train_wgan_gen_adam_beta1, beta_2=cfg.train_wgan_gen_adam_beta2, epsilon=1e-7)
        print('        optimizer: Adam')
        if self.world > 1:   # identical replicas to start from
            parallel.broadcast_(self.critic_opti.flat.flat)
            parallel.broadcast_(self.gen_opti.flat.flat)
        ops.sync_batchnorm(self.world if bool(getattr(cfg, 'train_sync_batchnorm', False)) else 1)
        ops.bf16_products(bool(getattr(cfg, 'train_wgan_bf16_products', False)))

        # kept for API compatibility: Keras needed target arrays, the kernels take the signs directly
        self.wgan_valid = -np.ones((cfg.train_batch_size, 1, 1))
        self.wgan_fake = np.ones((cfg.train_batch_size, 1, 1))
        self.wgan_dummy = np.zeros((cfg.train_batch_size, 1, 1))

        # the spectral branch alone, for the critic step
        self._gen_spec = None
        node_spec = getattr(self._model, 'node_spec', None)
        if cfg.train_wgan_prune_dead_branches and node_spec is not None and self.critic.cfgarch is not None:
            self._gen_spec = kl.Model(inputs=generator.inputs[0], outputs=node_spec)

        if self._errtype == 'WGAN':
            print('        use WGAN optimization')
            self._w_ls, self._wgan_weight = None, 1.0
        elif self._errtype == 'WLSWGAN':
            print('        use WLSWGAN optimization')
            w_ls, ww = self._wls_weights()
            self._w_ls = torch.as_tensor(w_ls, dtype=torch.float32, device=dev).contiguous()
            self._wgan_weight = ww
        else:
            raise ValueError('unknown error type ' + str(self._errtype))

        # these two names are what the reference exposes after prepare()
        generator.parallel_branches = bool(cfg.train_wgan_parallel_streams)
        generator.side_backward_first = bool(getattr(cfg, 'train_wgan_side_backward_first', False))
        self.critic_model = self.critic_net
        self.generator_model = generator
        self._graphs = {}
        self._graph_choice = {}      # cfg.train_wgan_hipgraph = 'tune': (kind, shapes) -> replay the step as a hipGraph?
        self._graph_tuning = {}      # ... and the two timings behind the choice
        self._pending = {}           # 'critic' / 'generator' -> event of an optimiser update still running on the communication stream
        self._comm = None

    # ---- device-side losses ----------------------------------------------------------------------------------
    def _fake_sample(self, X, training):
        """G(x) with the generator frozen: batch statistics in training mode but NO moving-average update
        (hard parts).  With the dead branches pruned: the spectral branch alone, [B,T,spec] -- the only columns
        of G(x) the critic reads (networks_critic.py:57-59); critic_loss takes either form."""
        memo = {'freeze_bn_stats': True}
        self._wait_update('generator')
        with torch.no_grad():
            if self._gen_spec is not None:
                return self._gen_spec(X, training=training, memo=memo)
            return self._model.kerasmodel(X, training=training, memo=memo)

    def _spec_of(self, t):
        """The spectral columns of a sample [B,T,out] (a view), or the tensor itself if it already is the spectral slice [B,T,spec]."""
        voc = self._model.vocoder
        F = voc.specsize()
        if t.shape[-1] == F and voc.featuressize() != F:
            return t.reshape(t.shape[0], t.shape[1], F)
        return t[:, :, 1:1 + F]

    def critic_loss(self, X, Y, alpha=None, training=True, fake=None):
        """Total critic loss and its three parts on device tensors X [B,T,ctx], Y [B,T,out]; `fake`: G(x) as [B,T,out] or just its
        spectral columns [B,T,spec]."""
        if fake is None:
            fake = self._fake_sample(X, training)
        self._wait_update('critic')
        streams = bool(getattr(self.cfg, 'train_wgan_parallel_streams', False))
        node = getattr(self.critic, 'node_spec_in', None)
        B = Y.shape[0]
        if node is not None and getattr(self.cfg, 'train_wgan_feed_spectra', True):
            # The critic reads the spectral columns only (networks_critic.py:57-59).  It is fed AT its slice: real and fake spectra are
            # copied once into the two halves of one [2B,T,spec] tensor (no 86-column fake sample, no concatenation, no slice copies),
            # and x^ is interpolated between those halves.  d D(x^) / d x^ is zero in the columns the critic does not read, so the
            # penalty's norm over [T,spec] IS the reference's norm over [T,out] (optimizertts_wgan.py:53-68).
            F = self._model.vocoder.specsize()
            # (one buffer of 3B samples: the stacked pair and x^ lie back to back, so that a layer's forward over both evaluations is
            # ONE launch over 3B rows -- Model.forward_multi_at(pair=True), ops.Conv2dPairFn)
            spec3 = torch.empty((3 * B, Y.shape[1], F), dtype=torch.float32, device=Y.device)
            spec2 = spec3[:2 * B]
            spec2[:B].copy_(self._spec_of(Y))
            spec2[B:].copy_(self._spec_of(fake))
            if alpha is None:
                alpha = torch.rand(B, device=Y.device, dtype=torch.float32)
            x_hat = ops.gp_interpolate(spec2[:B], spec2[B:], alpha.reshape(-1).contiguous(), out=spec3[2 * B:]).detach().requires_grad_(True)
            feed = {self.critic.input_ctx: X}
            if getattr(self.cfg, 'train_wgan_stack_real_fake', True) and self._critic_is_per_sample():
                # no BatchNorm in the critic: critic(real) and critic(fake) are one pass over the stacked 2B batch (half the launches,
                # one weight-gradient product per layer instead of two; the context branch stays shared, at B)
                both, v_hat = self.critic_net.forward_multi_at(node, [spec2, x_hat], feed, training=training, parallel_streams=streams,
                                                                shared_stream=bool(getattr(self.cfg, 'train_wgan_ctx_stream', False)),
                                                                pair=bool(getattr(self.cfg, 'train_wgan_pair_forward', True)))
                l_valid, l_fake = ops.wasserstein_pair(both, B)
            else:
                valid, fake_v, v_hat = self.critic_net.forward_multi_at(node, [spec2[:B], spec2[B:], x_hat], feed, training=training,
                                                                        parallel_streams=streams)
                l_valid = wasserstein_loss(-1.0, valid)
                l_fake = wasserstein_loss(+1.0, fake_v)
            gp = gradient_penalty_loss(None, v_hat, x_hat)
            total = l_valid + l_fake + float(self.cfg.train_wgan_pg_lambda) * gp
            return total, (l_valid, l_fake, gp)
        if fake.shape[-1] != Y.shape[-1]:          # a custom critic fed whole samples: the spectral branch embedded in zeros
            full = torch.zeros_like(Y)
            full[:, :, 1:1 + fake.shape[-1]] = fake
            fake = full
        x_hat = RandomWeightedAverage(X.shape[0])([Y, fake], alpha).requires_grad_(True)
        if getattr(self.cfg, 'train_wgan_stack_real_fake', True) and self._critic_is_per_sample():
            both, v_hat = self.critic_net.forward_multi(0, [torch.cat([Y, fake], 0), x_hat], [X], training=training,
                                                        parallel_streams=streams)
            valid, fake_v = both[:B], both[B:]
        else:
            valid, fake_v, v_hat = self.critic_net.forward_multi(0, [Y, fake, x_hat], [X], training=training,
                                                                   parallel_streams=streams)
        l_valid = wasserstein_loss(-1.0, valid)
        l_fake = wasserstein_loss(+1.0, fake_v)
        gp = gradient_penalty_loss(None, v_hat, x_hat)
        total = l_valid + l_fake + float(self.cfg.train_wgan_pg_lambda) * gp
        return total, (l_valid, l_fake, gp)

    def _critic_is_per_sample(self):
        """True when no critic layer couples the samples of a batch (no BatchNormalization): stacking evaluations is exact."""
        ok = getattr(self, '_critic_per_sample', None)
        if ok is None:
            from . import layers as _layers
            ok = not any(isinstance(l, _layers.BatchNormalization) for l in self.critic_net.layers_list)
            self._critic_per_sample = ok
        return ok

    def _can_split_generator(self):
        m = self._model.kerasmodel
        return getattr(self, '_gen_spec', None) is not None and getattr(self._model, 'node_spec', None) is not None and m.single_output and \
            bool(getattr(self.cfg, 'train_wgan_early_critic', True))

    def generator_forward_early(self, X, Y=None, training=True):
        """The generator's forward up to (not including) its final concatenation, with the autograd tape: everything of the generator
        step that does not depend on the critic.  `device_step` launches it BEFORE the critic step of a batch that trains both networks
        (reference optimizertts_wgan.py:225-240: critic step, then generator step, on the same batch; G is not touched by the critic's
        update, so the values are the same): the BLSTM's 400-step chain on its side stream then runs under the critic step instead of in
        front of the generator step's critic evaluation, and the critic step's fake sample -- G's spectral branch on the same batch and
        weights -- is taken from it instead of being computed again."""
        m = self._model.kerasmodel
        self._wait_update('generator')
        feed = {id(m.inputs[0]): X}
        early_bwd = Y is not None and self._errtype == 'WLSWGAN' and bool(getattr(m, 'parallel_branches', False)) and \
            bool(getattr(self.cfg, 'train_wgan_hoist_side_backward', True))
        values = m._run(feed, training, None, hold={id(m.outputs[0])}, cut_side=early_bwd)
        self._gen_cuts = []
        if early_bwd and values.get('__cuts__') and '__side__' in values:
            # The side branch's BACKWARD as well: its output (f0) is read by the least-squares term only -- the critic slices the
            # spectral columns -- and that term is a mean of per-element squares, so d loss / d f0 needs nothing but f0 and Y.
            # The term is evaluated on a tensor with the branch's output in its columns and zeros elsewhere (the value is
            # discarded, the gradient w.r.t. the branch's columns is the true one), on the branch's stream, and backpropagated to
            # the cut: the whole 2 x 400-step recurrence chain then runs under the critic step.  The leaves' gradients join the
            # rest of the graph in _generator_grads.
            on_side, pending, side, cur = values['__side__']
            out_node = m.outputs[0]
            sp = [p for p in out_node.parents if id(p) in on_side]
            if sp and all(id(p) in values for p in out_node.parents if id(p) in on_side):
                with torch.cuda.stream(side):
                    cols = []
                    for p in out_node.parents:
                        if id(p) in on_side:
                            cols.append(kl.to_tensor(values[id(p)]))
                        else:
                            cols.append(torch.zeros(X.shape[0], X.shape[1], int(p.shape[-1]), dtype=torch.float32, device=X.device))
                    pred_tmp = torch.cat(cols, dim=-1)
                    l_tmp = specweighted_lse_loss(Y, pred_tmp, self._w_ls)
                    ops.lstm_dx_ready = None
                    l_tmp.backward()
                    for p in sp:                               # the branch is done with: the join sees constants
                        values[id(p)] = kl.to_tensor(values[id(p)]).detach()
                    ev = side.record_event()
                    # The main stream needs the branch's dx only: the recurrence's weight-gradient products behind it (0.5 ms) may
                    # still run on the side stream while the trunk's backward goes on.  The LSTM's backward publishes the event right
                    # behind its dx product; it is the leaf's gradient if autograd handed that very tensor over (no copy behind it).
                    dxr = ops.lstm_dx_ready
                    for t, leaf, holder in values['__cuts__']:
                        if leaf.grad is not None:
                            early = dxr is not None and len(values['__cuts__']) == 1 and leaf.grad.data_ptr() == dxr[0]
                            holder['grad'], holder['event'] = leaf.grad, (dxr[1] if early else ev)
                            self._gen_cuts.append(t)
                    self._side_tail_event = ev                 # (the weight-gradient products: joined before the optimiser reads the buffer)
        values.pop('__cuts__', None)
        return feed, values

    def fake_from_early(self, X, pre):
        """The critic step's fake sample from the hoisted forward: the spectral branch [B,T,spec], detached (the other columns are not read)."""
        _, values = pre
        spec = kl.to_tensor(values[id(self._model.node_spec)]).detach()
        return spec.reshape(spec.shape[0], spec.shape[1], -1)

    def generator_loss(self, X, Y, training=True, pre=None):
        m = self._model.kerasmodel
        node_spec = getattr(self._model, 'node_spec', None)
        self._wait_update('generator')
        if self._can_split_generator():
            # The critic reads the spectral columns only (the condition under which the critic step prunes the other
            # branches).  It is therefore fed the spectral branch as soon as that exists, while the latency-bound f0
            # branch (BLSTM, side stream) still runs; the final concatenation -- the join with the side stream -- is
            # needed by the least-squares term alone and comes last.  Same values.  (cfg.train_wgan_side_backward_first creates
            # the BLSTM's autograd node last -- launches first, layers.Model._run -- so that its backward chain is enqueued
            # first: the chain then ends 1 ms earlier but the step does not, tools/gen_timeline.py events; off.)
            out_node = m.outputs[0]
            if pre is not None:
                feed, values = pre
            else:
                feed = {id(m.inputs[0]): X}
                values = m._run(feed, training, None, hold={id(out_node)})
            ops._lstm_mark('G_spec_fwd_end')
            spec = kl.to_tensor(values[id(node_spec)])
            if ops.lstm_trace is not None and spec.requires_grad:
                spec.register_hook(lambda g: ops._lstm_mark('critic_bwd_end'))
            voc = self._model.vocoder
            self._wait_update('critic')          # the critic's update of this batch may still be in flight: G's forward above did not need it
            cnode = getattr(self.critic, 'node_spec_in', None)
            if cnode is not None and getattr(self.cfg, 'train_wgan_feed_spectra', True):
                # the critic fed at its spectral slice (see critic_loss): no zero-padded 86-column sample, no slice copy, and the
                # gradient comes back as [B,T,spec] straight into the generator's last convolution
                valid = self.critic_net.forward_multi_at(cnode, [spec.reshape(spec.shape[0], spec.shape[1], voc.specsize())],
                                                         {self.critic.input_ctx: X}, training=training)[0]
            else:
                feat = torch.zeros(X.shape[0], X.shape[1], voc.featuressize(), dtype=torch.float32, device=X.device)
                feat[:, :, 1:1 + voc.specsize()] = spec
                valid = self.critic_net(feat, X, training=training)
            l_w = wasserstein_loss(-1.0, valid)
            ops._lstm_mark('critic_fwd_end')
            values = m._run(feed, training, None, values=values)
            pred = kl.to_tensor(values[id(out_node)])
            ops._lstm_mark('join')
        else:
            pred = m(X, training=training)
            self._wait_update('critic')
            valid = self.critic_net(pred, X, training=training)
            l_w = wasserstein_loss(-1.0, valid)
        if self._errtype == 'WGAN':
            return l_w, (l_w, None)
        l_ls = specweighted_lse_loss(Y, pred, self._w_ls)
        return self._wgan_weight * l_w + l_ls, (l_w, l_ls)

    # ---- device-side steps (no host synchronisation) -----------------------------------------------------------
    # ---- optimiser updates: all-reduce of the flat gradient + Adam, optionally on a communication stream -------------------
    def _async(self):
        a = getattr(self.cfg, 'train_wgan_async_update', None)
        return (self.world > 1) if a is None else bool(a)

    def _wait_update(self, kind):
        """Make the current stream wait for a still-running update of that network's weights (no-op otherwise)."""
        ev = self._pending.pop(kind, None) if getattr(self, '_pending', None) else None
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def wait_updates(self):
        self._wait_update('critic'); self._wait_update('generator')

    def _update(self, kind):
        """All-reduce (sum over ranks; the 1/world goes into Adam's gscale) and the Adam step of one network.  With
        cfg.train_wgan_async_update both run on a communication stream behind an event of the backward pass: the compute
        stream goes straight on to the next forward that does not read these weights (the frozen generator's sample of the
        next critic step; the generator's forward of the generator step) and waits only where it needs them."""
        opti = self.critic_opti if kind == 'critic' else self.gen_opti

        def run():
            # explicit ordering, the same for RCCL and gloo: the collective is started (RCCL: on its own stream, behind everything
            # already queued on the current one), work.wait() makes the current stream (gloo: the host) wait for the sums, and only
            # then is Adam queued -- nothing here depends on which stream a blocking all_reduce would have synchronised with
            work, gscale = parallel.allreduce_sum_async(opti.flat.grad)
            if work is not None:
                work.wait()
            opti.step(gscale)
            if kind == 'critic' and self.cfg.train_wgan_weight_clip:
                c = float(self.cfg.train_wgan_weight_clip)
                opti.clip_weights(-c, c)

        if not self._async() or torch.cuda.is_current_stream_capturing():
            run()
            return
        if self._comm is None:
            from . import layers
            self._comm = layers.side_streams(1, 'comm')[0]
        # the backward pass's last kernel -> event -> communication stream; the compute stream goes on at once
        done = torch.cuda.current_stream().record_event()
        self._comm.wait_event(done)
        with torch.cuda.stream(self._comm):
            run()
            self._pending[kind] = self._comm.record_event()

    def _critic_grads(self, X, Y, alpha=None, fake=None):
        if fake is None:
            fake = self._fake_sample(X, True)      # first: it does not need the critic's weights (a pending update may still run)
        self._wait_update('critic')
        self.critic_opti.zero_grad()
        with ops.deferred_weight_grads():       # the Dense layers' weight gradients run as one grouped launch at exit
            total, _ = self.critic_loss(X, Y, alpha, training=True, fake=fake)
            total.backward()
        return total.detach()

    def critic_step(self, X, Y, alpha=None, fake=None):
        total = self._critic_grads(X, Y, alpha, fake)
        self._update('critic')
        return total

    def _generator_grads(self, X, Y, pre=None):
        self._wait_update('generator')
        if pre is None:
            self.gen_opti.zero_grad()          # (a hoisted forward may already have run a side branch's backward: device_step zeroed before it)
        cps = self.critic_opti.flat.params
        for p in cps: p.requires_grad_(False)      # frozen critic (:160-161)
        try:
            with ops.deferred_weight_grads():
                if pre is not None:
                    ops.deferred_attach(getattr(self, '_gen_deferred', None))
                    self._gen_deferred = None
                total, _ = self.generator_loss(X, Y, training=True, pre=pre)
                ops._lstm_mark('loss')
                cuts = getattr(self, '_gen_cuts', None) if pre is not None else None
                if cuts:
                    # the injection nodes at the side branch's cut add the gradient its early backward left behind; as extra roots (with a
                    # zero gradient) they are reached even if no later layer consumed them
                    torch.autograd.backward([total] + cuts, [None] + [torch.zeros_like(t) for t in cuts])
                    self._gen_cuts = None
                else:
                    total.backward()
                ops._lstm_mark('bwd_enqueued')
            ops._lstm_mark('wgrads_flushed')
        finally:
            for p in cps: p.requires_grad_(True)
        return total.detach()

    def generator_step(self, X, Y, pre=None):
        total = self._generator_grads(X, Y, pre)
        self._update('generator')
        return total

    # ---- training state kept across the extra step executions of a capture / tuning run ------------------------------------
    def _state_snapshot(self):
        """Everything a training step changes: both networks' weights, Adam moments and step counters, the layers' buffers
        (BatchNorm moving averages) and the device's random stream."""
        self.wait_updates()
        snap = {'opt': [], 'buf': [], 'rng': torch.cuda.get_rng_state(), 'gen_updates': self.generator_updates}
        for o in (self.critic_opti, self.gen_opti):
            snap['opt'].append((o, o.flat.flat.clone(), o.m.clone(), o.v.clone(), o.step_count.clone()))
        for net in (self.critic_net, self._model.kerasmodel):
            for b in net.buffers():
                snap['buf'].append((b, b.clone()))
        return snap

    def _state_restore(self, snap):
        self.wait_updates()
        for o, w, m, v, t in snap['opt']:
            o.flat.flat.copy_(w); o.m.copy_(m); o.v.copy_(v); o.step_count.copy_(t)
            o.flat.epoch += 1                     # weights changed behind every weight-keyed cache
        for b, val in snap['buf']:
            b.copy_(val)
        torch.cuda.set_rng_state(snap['rng'])
        self.generator_updates = snap['gen_updates']
        ops.clear_caches()

    def _tune_graph(self, kind, X, Y):
        """cfg.train_wgan_hipgraph = 'tune': time the step of this kind and shape eagerly (side streams on) and as a hipGraph
        replay (single stream, no host work), keep the faster.  The timing runs are real steps on the first batch, so the training
        state is put back afterwards: the run that follows is the one an untuned run would have made."""
        key = (kind, tuple(X.shape), tuple(Y.shape))
        if key in self._graph_choice:
            return self._graph_choice[key]
        snap = self._state_snapshot()
        def timed(fn, n=3):
            torch.cuda.synchronize()
            t = time.time()
            for _ in range(n): fn()
            torch.cuda.synchronize()
            return (time.time() - t) / n
        eager = (lambda: self.critic_step(X, Y)) if kind == 'critic' else (lambda: self.generator_step(X, Y))
        graph = lambda: self._graphed(kind, X, Y)
        for _ in range(2): eager()          # both forms warm (allocator, weight-keyed caches, the capture itself) before either is timed
        ok = 1.0
        try:
            graph()
        except Exception as e:              # a capture this runtime / collective backend refuses: the step stays eager, and the line says why
            ok = 0.0
            self._graph_tuning[key] = {'graph': False, 'capture_error': repr(e)[:300]}
            torch.cuda.synchronize()
        if self.world > 1:                  # every rank must take the same branch below (the timed runs contain collectives)
            ok = 1.0 if parallel.max_over_ranks(1.0 - ok, self.device) == 0.0 else 0.0
        if ok == 0.0:
            self._graphs = {k: v for k, v in self._graphs.items() if k[0] != kind}
            self._state_restore(snap)
            self._graph_choice[key] = False
            self._graph_tuning.setdefault(key, {'graph': False, 'capture_error': 'on another rank'})
            return False
        t_eager, t_graph = [], []
        for _ in range(2):                  # alternating: clocks and caches drift over the first seconds of a process
            t_eager.append(timed(eager)); t_graph.append(timed(graph))
        t_eager, t_graph = min(t_eager), min(t_graph)
        if self.world > 1:
            t_eager, t_graph = parallel.max_over_ranks(t_eager, self.device), parallel.max_over_ranks(t_graph, self.device)
        self._state_restore(snap)
        # a replay costs the host one launch: inside the training loop that time goes to the launches of whatever else is in flight (the
        # generator's forward hoisted in front of the critic step), so a step that replays within 8 % of its eager time is replayed --
        # measured in the loop, the eager choice at equal isolated times was 5 % slower (3.51 against 3.70 M frames/s)
        # (the generator step the other way round: replayed, its forward cannot be hoisted in front of the critic step -- worth 1.8 ms)
        choice = bool(t_graph < 1.08 * t_eager) if kind == 'critic' else bool(t_graph < 0.85 * t_eager)
        self._graph_choice[key] = choice
        self._graph_tuning[key] = {'eager_ms': t_eager * 1e3, 'graph_ms': t_graph * 1e3, 'graph': choice}
        return self._graph_choice[key]

    # hipGraph replay of a whole step: static input buffers, one capture per (kind, shape)
    def _graphed(self, kind, X, Y, alpha=None, fake=None):
        """hipGraph replay of a step.  One process: the whole step (forward, backward, Adam) is one graph.  Data parallel: the
        graph ends with the backward pass -- the gradient all-reduce cannot be captured -- and the update follows eagerly.
        `fake` (critic step): the fake sample is an INPUT of the graph (a second graph per shape, without the frozen generator's
        forward) -- the batches on which the generator's forward was hoisted in front of the critic step."""
        whole = self.world == 1 and not bool(getattr(self.cfg, 'train_wgan_graph_split', False))
        with_fake = kind == 'critic' and fake is not None
        key = (kind, tuple(X.shape), tuple(Y.shape), whole, with_fake)
        ent = self._graphs.get(key)
        if ent is None:
            snap = self._state_snapshot()       # the warm-up below runs real steps: the state is put back before the first replay
            sX, sY = X.clone(), Y.clone()
            sA = torch.rand(X.shape[0], device=X.device, dtype=torch.float32)
            sF = fake.detach().clone() if with_fake else None
            if kind == 'batch':
                assert whole, 'the whole-batch graph is a one-process form'
                fn = lambda: self._batch_steps(sX, sY, sA, True, False, False)
            elif whole:
                fn = (lambda: self.critic_step(sX, sY, sA, sF)) if kind == 'critic' else (lambda: self.generator_step(sX, sY))
            else:
                fn = (lambda: self._critic_grads(sX, sY, sA, sF)) if kind == 'critic' else (lambda: self._generator_grads(sX, sY))
            from . import layers
            # the graph is captured on one stream: the evaluations' side streams would become cross-stream edges of the capture
            saved_streams = (self.cfg.train_wgan_parallel_streams, getattr(self._model.kerasmodel, 'parallel_branches', False))
            if kind == 'batch':
                # one fork: the generator's BLSTM branch stays on its side stream (its chain runs beside the critic step); the critic's
                # three evaluations go on one stream (a capture with their cross-stream edges replayed at half the speed)
                self.cfg.train_wgan_parallel_streams = False
            elif not bool(getattr(self.cfg, 'train_wgan_graph_streams', False)):
                self.cfg.train_wgan_parallel_streams = False
                self._model.kerasmodel.parallel_branches = False
            side = layers.side_streams(1, 'capture')[0]
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):     # warm-up outside capture (allocator, workspace growth)
                    fn()
                    if not whole:
                        self._update(kind)
                    # an asynchronous update of the warm-up must not be left pending: a wait recorded outside the capture
                    # would be popped by the first _wait_update() inside it and be no dependency of the graph
                    self.wait_updates()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            # ... except those of the FROZEN generator inside a critic step: its context kernel's frequency-domain planes (42 us to
            # rebuild) change once per generator update, not per replay -- the capture reads the planes the warm-up left on this stream
            # and every replay is preceded by ops._C1FFT.refresh_planes when the weights have changed (below)
            frozen_prev = ops._C1FFT.frozen
            frozen_items = []
            if kind == 'critic' and bool(getattr(self.cfg, 'train_wgan_graph_frozen_planes', True)):
                ops._C1FFT.frozen = {id(self.gen_opti.flat)}
                ops._C1FFT.frozen_log = frozen_items
            ops.clear_caches()         # every derived operand (bf16 planes, Toeplitz tables) must be rebuilt inside the graph
            try:
                with torch.cuda.graph(g, stream=side):      # the stream of the warm-up: stream-keyed operand caches (weight planes) keep their entries and are refreshed grouped
                    out = fn()
            finally:
                ops._C1FFT.frozen_log = None
                ops.clear_caches()         # ... and the graph's private copies are not for eager code
                ops._C1FFT.frozen = frozen_prev
                self.cfg.train_wgan_parallel_streams, self._model.kerasmodel.parallel_branches = saved_streams
            if not hasattr(self, '_graph_frozen'):
                self._graph_frozen = {}
            # the buffers this graph reads for the frozen generator's kernels, and the state of its weights they were built from
            self._graph_frozen[key] = {'items': frozen_items, 'epoch': self.gen_opti.flat.epoch}
            ent = (g, sX, sY, sA, out, sF)
            self._graphs[key] = ent
            self._state_restore(snap)
        g, sX, sY, sA, out, sF = ent
        sX.copy_(X); sY.copy_(Y)
        if sF is not None:
            sF.copy_(fake)
        if kind in ('critic', 'batch'):
            if alpha is None: sA.uniform_(0.0, 1.0)
            else: sA.copy_(alpha.reshape(-1))
        self.wait_updates()            # the replay reads (and, in one process, writes) both networks' weights
        fz = getattr(self, '_graph_frozen', {}).get(key)
        if fz is not None and fz['items'] and fz['epoch'] != self.gen_opti.flat.epoch:
            # the frozen generator's kernel planes the graph reads (see the capture): rebuilt if the generator's weights have changed
            ops._C1FFT.refresh_planes(fz['items'])
            fz['epoch'] = self.gen_opti.flat.epoch
        g.replay()
        if whole:
            # the replayed Adam / clip kernels changed the weights behind every weight-keyed cache (bf16 planes, Toeplitz
            # tables): the Python-side epoch bump of KerasAdam.step() is not part of the graph
            if kind in ('critic', 'batch'): self.critic_opti.flat.epoch += 1
            if kind in ('generator', 'batch'): self.gen_opti.flat.epoch += 1
        else:
            self._update(kind)
        return out

    def _use_graph(self, X, kind=None, Y=None):
        """cfg.train_wgan_hipgraph: True / False; 'auto' = replay the step as a hipGraph when it is launch-bound (few frames:
        about 1 000 launches of a few microseconds each against 8 ms of host enqueue time at the reference's B = 10); 'tune' = as
        'auto' below the frame threshold, above it measured per step kind on the first batch (`_tune_graph`): a step whose kernels
        are short against the host's enqueue time (the bf16 critic step) replays faster than it launches, one that lives on
        overlapping streams (the generator step's BLSTM branch) does not."""
        pin = getattr(self.cfg, 'train_wgan_graph_' + kind, None) if kind in ('critic', 'generator') else None
        if pin is not None:             # 'on' / 'off' (bench.py --graph-critic / --graph-generator): the timed program is pinned,
            return pin in (True, 'on')  # not left to the wall-clock comparison of 'tune' (two runs of one tree time ONE program)
        g = self.cfg.train_wgan_hipgraph
        if g in ('auto', 'tune'):
            if X.shape[0] * X.shape[1] <= int(getattr(self.cfg, 'train_wgan_hipgraph_maxframes', 8192)):
                return True
            if g == 'tune' and kind is not None:
                # (more than one rank: the graph holds forward + backward only, the collective and Adam follow eagerly --
                # _graphed's split form; the timings are maxed over the ranks so that every rank makes the same choice)
                return self._tune_graph(kind, X, Y)
            return False
        return bool(g)

    def hint_next_batch(self, X_next, Y_next):
        """The batch the NEXT train_on_batch will get (device tensors, the very objects that will be passed), or None.  With it a
        generator step's forward is launched ONE BATCH AHEAD (cfg.train_wgan_generator_lookahead, see device_step)."""
        self._next_batch = None if X_next is None else (X_next, Y_next)

    def device_step(self, batchid, X, Y, alpha=None, nxt=None):
        """One `train_on_batch` worth of device work on resident tensors; returns (critic_loss, generator_loss|None)
        as device scalars.

        `nxt` = (X, Y) of the next batch (the tensors the next call will be given), optional.  The reference trains the generator on
        every critic_runs-th batch (optimizertts_wgan.py:225-240); its forward depends on the generator's weights -- untouched since
        the previous generator step -- and on that batch's labels only.  Knowing the next batch, the forward (and, as with the hoist
        inside a batch, the BLSTM branch's backward) of a generator step is launched BEFORE the critic step of the batch in front of
        it: the two 400-step recurrence chains, 5.7 ms on their own, then run under TWO critic steps instead of one, and the
        generator step itself is left with the critic's evaluation, the backward pass and the update.  Same weights, same inputs,
        same arithmetic: the results are those of the plain order (tested)."""
        critic_runs = 10 if (self.generator_updates < 25) or (self.generator_updates % 500 == 0) else 5   # (:225-228)
        gen_too = batchid % critic_runs == 0
        # a kernel of an earlier step that gave up on a hand-off has left its code in the device status word: an error at the step
        # boundary (a host memory load, no synchronisation; graph replays bypass the per-call checks of the C ABI), not a bad step
        from . import _hip
        _hip.check_status()
        # the generator step that follows on the same batch reuses the generator's context-Conv1D product of the critic
        # step's fake sample (same input, same not-yet-updated kernel): ops._C1Cache, valid inside this call only
        ops.bf16_products(bool(getattr(self.cfg, 'train_wgan_bf16_products', False)))
        split = getattr(self.cfg, 'train_wgan_split_bf16', None)
        if split is not None:
            if bool(split) != ops._C1Split.enabled:
                ops.conv1d_split(split)
            ops.dense_split(split)
        graph_c = self._use_graph(X, 'critic', Y)
        graph_g = gen_too and self._use_graph(X, 'generator', Y)
        use_graph = graph_c or graph_g
        ops.conv1d_cache(gen_too and not use_graph and bool(getattr(self.cfg, 'train_wgan_reuse_ctx_conv', True)))
        try:
            if gen_too and not graph_g and self._use_batch_graph(X, Y, graph_c):
                # the whole train_on_batch of this batch (hoisted generator forward, critic step, generator step) as ONE hipGraph with
                # the BLSTM branch's fork / join kept: no host time at all
                lc, lg = self._graphed('batch', X, Y, alpha)
                self.generator_updates += 1
                return lc, lg
            if nxt is None:
                nxt, self._next_batch = getattr(self, '_next_batch', None), None
            next_gen = nxt is not None and not gen_too and (batchid + 1) % critic_runs == 0
            lc, lg = self._batch_steps(X, Y, alpha, gen_too, graph_c, graph_g, nxt if next_gen else None,
                                       nxt if (nxt is not None and not gen_too and not next_gen) else None)
            if gen_too:
                self.generator_updates += 1
        finally:
            ops.conv1d_cache(False)
        return lc, lg

    def _batch_steps(self, X, Y, alpha, gen_too, graph_c, graph_g, nxt=None, nxt_critic=None):
        """The steps of one train_on_batch: critic step, and the generator step when `gen_too` (its forward hoisted in front of the
        critic step -- or of the previous batch's critic step, see device_step -- by generator_forward_early)."""
        pre = fake = None
        hoist = bool(getattr(self.cfg, 'train_wgan_hoist_generator', True)) and self._can_split_generator()
        ahead, self._ahead = getattr(self, '_ahead', None), None
        if ahead is not None and not (gen_too and not graph_g and hoist and ahead['X'] is X and ahead['Y'] is Y and
                                      ahead['epoch'] == self.gen_opti.flat.epoch):
            self._drop_ahead(ahead)
            ahead = None
        if gen_too and not graph_g and hoist:
            if ahead is not None:
                # launched one batch ago (below): nothing of the generator's forward is left to do
                pre, self._gen_deferred, self._gen_cuts = ahead['pre'], ahead['deferred'], ahead['cuts']
            else:
                # G's forward first (see generator_forward_early); inside deferred_weight_grads() so that its layers note their
                # gradient targets as they do inside the generator step
                self.gen_opti.zero_grad()
                with ops.deferred_weight_grads():
                    pre = self.generator_forward_early(X, Y, training=True)
                    self._gen_deferred = ops.deferred_detach()      # (joined and flushed by the generator step's own context)
            fake = self.fake_from_early(X, pre)
        fa, self._fake_ahead = getattr(self, '_fake_ahead', None), None
        if fa is not None and fake is None and fa['X'] is X and fa['epoch'] == self.gen_opti.flat.epoch:
            # the frozen generator's sample of this batch was drawn one batch ago on a side stream (below)
            cur = torch.cuda.current_stream()
            cur.wait_event(fa['event'])
            fake = fa['fake']
            fake.record_stream(cur)
        if nxt_critic is not None and bool(getattr(self.cfg, 'train_wgan_fake_ahead', False)):
            self._fake_sample_ahead(nxt_critic[0])
        if nxt is not None and hoist and bool(getattr(self.cfg, 'train_wgan_generator_lookahead', True)) and \
                not self._use_graph(nxt[0], 'generator', nxt[1]):
            # the NEXT batch trains the generator: its forward goes out now, in front of this batch's critic step
            Xn, Yn = nxt
            self._wait_update('generator')          # (the previous generator update reads the gradient buffer zeroed here)
            self.gen_opti.zero_grad()
            # the forward moves the BatchNorm moving averages: kept (one launch) for the case that another batch comes (_drop_ahead)
            bufs = [b for b in self._model.kerasmodel.buffers() if b.numel() > 0]
            kept = torch.cat([b.detach().reshape(-1) for b in bufs]) if bufs else None
            with ops.deferred_weight_grads():
                pre_n = self.generator_forward_early(Xn, Yn, training=True)
                dfr = ops.deferred_detach()
            self._ahead = {'X': Xn, 'Y': Yn, 'pre': pre_n, 'deferred': dfr, 'cuts': self._gen_cuts, 'epoch': self.gen_opti.flat.epoch,
                           'buffers': (bufs, kept)}
            self._gen_cuts = None
        ops._lstm_mark('critic_step_begin')
        if graph_c:
            lc = self._graphed('critic', X, Y, alpha, fake)
        else:
            lc = self.critic_step(X, Y, alpha) if fake is None else self.critic_step(X, Y, alpha, fake)
        ops._lstm_mark('critic_step_end')
        lg = None
        if gen_too:
            if graph_g:
                lg = self._graphed('generator', X, Y)
            else:
                lg = self.generator_step(X, Y) if pre is None else self.generator_step(X, Y, pre)
        return lc, lg

    def _fake_sample_ahead(self, Xn):
        """The next batch trains the critic alone and the generator is not updated before it: the frozen generator's sample of THAT batch
        (reference optimizertts_wgan.py:126-131 -- it depends on the generator's weights and the batch's labels only) is drawn now, on a
        side stream, beside this batch's critic step instead of in front of the next one.  The critic step's launches are single-round
        and latency-bound: a second, independent chain of launches fills the compute units they leave idle.  The
        step's own stream -- and its hipGraph, which then takes the sample as an input -- stays a single stream."""
        cur = torch.cuda.current_stream()
        self._wait_update('generator')                  # on the step's stream (the pending update is known to ONE waiter) ...
        side = getattr(self, '_fake_stream', None)
        if side is None:
            side = self._fake_stream = kl.side_streams(2)[0]       # (an existing one: every stream created costs every later launch)
        side.wait_stream(cur)                           # ... and the side stream behind it
        with torch.cuda.stream(side):
            f = self._fake_sample(Xn, True)
            ev = side.record_event()
        self._fake_ahead = {'X': Xn, 'fake': f, 'event': ev, 'epoch': self.gen_opti.flat.epoch}

    def _drop_ahead(self, ahead):
        """A generator forward launched one batch ahead for a batch that did not come (the caller named another one, the weights were
        restored, the step kind changed): everything it changed is put back -- the BatchNorm moving averages it moved, the gradient
        buffer its side branch's early backward pass added into (behind that branch's stream) -- and its queued weight-gradient
        products are forgotten."""
        cur = torch.cuda.current_stream()
        for q in ahead['deferred'][2]:
            cur.wait_stream(q)
        bufs, kept = ahead['buffers']
        off = 0
        with torch.no_grad():
            for b in bufs:
                b.copy_(kept[off:off + b.numel()].view_as(b))
                off += b.numel()
        self.gen_opti.zero_grad()

    def _use_batch_graph(self, X, Y, graph_c):
        """cfg.train_wgan_hipgraph = 'tune' (one process): is a batch that trains both networks replayed as ONE graph?  Timed on the first
        such batch against the separate steps (critic step as chosen, eager generator step with its forward hoisted)."""
        if self.cfg.train_wgan_hipgraph != 'tune' or self.world != 1 or not bool(getattr(self.cfg, 'train_wgan_batch_graph', False)):
            return False
        if X.shape[0] * X.shape[1] <= int(getattr(self.cfg, 'train_wgan_hipgraph_maxframes', 8192)):
            return False
        key = ('batch', tuple(X.shape), tuple(Y.shape))
        if key in self._graph_choice:
            return self._graph_choice[key]
        snap = self._state_snapshot()
        gu = self.generator_updates
        def timed(fn, n=3):
            torch.cuda.synchronize()
            t = time.time()
            for _ in range(n): fn()
            torch.cuda.synchronize()
            return (time.time() - t) / n
        sep = lambda: self._batch_steps(X, Y, None, True, graph_c, False)
        one = lambda: self._graphed('batch', X, Y)
        for _ in range(2): sep()
        one()
        t_sep, t_one = [], []
        for _ in range(2):
            t_sep.append(timed(sep)); t_one.append(timed(one))
        t_sep, t_one = min(t_sep), min(t_one)
        self._state_restore(snap)
        self.generator_updates = gu
        self._graph_choice[key] = bool(t_one < t_sep)
        self._graph_tuning[key] = {'separate_steps_ms': t_sep * 1e3, 'one_graph_ms': t_one * 1e3, 'graph': bool(t_one < t_sep)}
        return self._graph_choice[key]

    # ---- the reference's hooks --------------------------------------------------------------------------------------
    def train_on_batch(self, batchid, X_trab, Y_trab):
        X_trab, Y_trab = self._local_shard(X_trab, Y_trab)
        X, Y = self._to_dev(X_trab), self._to_dev(Y_trab)
        nb = getattr(self, '_next_batch', None)
        if nb is not None and torch.is_tensor(nb[0]) and nb[0].is_cuda:
            nb = (self._to_dev(nb[0]), self._to_dev(nb[1]))           # (identity for resident float32 tensors: the objects stay the same)
        else:
            nb = None                                                  # host arrays: no look-ahead (they would be copied twice)
        self._next_batch = None
        lc, lg = self.device_step(batchid, X, Y, nxt=nb)
        # the critic's loss stays on the device (no host synchronisation on the 4 of 5 batches that do not train the generator)
        self.costs_tra_critic_batches.append_device(lc)
        return None if lg is None else float(lg.item())

    def update_validation_cost(self, costs, X_vals, Y_vals):
        self.wait_updates()
        costs['model_rmse_validation'].append(data.cost_model_prediction_rmse(self._model, [X_vals], Y_vals))

        def gen_cost(x, y):
            with torch.no_grad():
                return float(self.generator_loss(self._to_dev(x), self._to_dev(y), training=False)[0].item())

        def critic_cost(y, x):
            total, _ = self.critic_loss(self._to_dev(x), self._to_dev(y), None, training=False)
            return float(total.item())

        costs['model_validation'].append(data.cost_model_mfn(gen_cost, [X_vals, Y_vals]))
        critic_batches = [c for c in self.costs_tra_critic_batches]      # (fetches the device-side entries)
        costs['critic_training'].append(np.mean(critic_batches))
        costs['critic_validation'].append(data.cost_model_mfn(critic_cost, [Y_vals, X_vals]))
        costs['critic_validation_ltm'].append(np.mean(costs['critic_validation'][-self.cfg.train_wgan_validation_ltm_winlen:]))
        cost_val = costs['critic_validation_ltm'][-1]

        if np.mean(critic_batches) <= 0.0:
            print('Average critic loss is negative: Training is likely to take ages to converge or not converge at all. ')
        self.costs_tra_critic_batches = _DeviceCosts()
        return cost_val

    def saveOptimizer(self, optimizer, fname):
        optimizer.save(fname)

    def loadOptimizer(self, optimizer, fname):
        try:
            optimizer.load(fname)
        except ValueError:
            print('Restoring optimizer failed from ' + fname + '. Fresh optimizer used instead (i.e. momentums, etc., might be wrong)')

    def saveTrainingStateLossSpecific(self, fstate):
        self.wait_updates()
        self.saveOptimizer(self.gen_opti, fstate + '.generator.optimizer.npz')
        self.saveOptimizer(self.critic_opti, fstate + '.critic.optimizer.npz')
        np.savez(fstate + '.model.weights.npz', *self._model.kerasmodel.get_weights())
        # the reference does not save the critic (its author's TODO at :310); without it a resumed run restarts D from scratch
        np.savez(fstate + '.critic.weights.npz', *self.critic_net.get_weights())

    def loadTrainingStateLossSpecific(self, fstate):
        self.loadOptimizer(self.gen_opti, fstate + '.generator.optimizer.npz')
        self.loadOptimizer(self.critic_opti, fstate + '.critic.optimizer.npz')
        with np.load(fstate + '.model.weights.npz') as z:
            self._model.kerasmodel.set_weights([z['arr_{}'.format(i)] for i in range(len(z.files))])
        import os
        if os.path.exists(fstate + '.critic.weights.npz'):
            with np.load(fstate + '.critic.weights.npz') as z:
                self.critic_net.set_weights([z['arr_{}'.format(i)] for i in range(len(z.files))])
        # the weights changed behind every weight-keyed cache (bf16 planes, Toeplitz tables)
        self.gen_opti.flat.epoch += 1
        self.critic_opti.flat.epoch += 1

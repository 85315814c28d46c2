"""The WGAN critic D(features, context) -> per-frame score; reference: percivaltts/networks_critic.py:34-96.

Spectral slice -> L x (Conv2D 5x5, C filters, bias, LeakyReLU .3) -> flatten; context -> Conv1D(k) -> 2 FC;
concat -> 3 FC -> Dense(1).  bn=False everywhere, so every layer carries a bias.  f0 and the noise mask are
not seen by the critic (networks_critic.py:57-59).  The three graph handles `input_features`, `input_ctx`,
`output` keep their names because OptimizerTTSWGAN.prepare reads them (optimizertts_wgan.py:115,126,165);
`model` is the executable module.
"""
from __future__ import print_function

from . import layers as kl
from .networktts import pFC, pCNN1D


class Critic:

    input_features = None
    input_ctx = None
    output = None

    vocoder = None
    ctxsize = -1
    cfgarch = None

    def __init__(self, vocoder, ctxsize, cfgarch):
        self.vocoder, self.ctxsize, self.cfgarch = vocoder, ctxsize, cfgarch
        bn = False

        self.input_features = kl.Input(shape=(None, vocoder.featuressize()), name='input_features')

        l_spec = kl.SliceLast(1, 1 + vocoder.specsize())(self.input_features)
        # the optimiser feeds the critic at this node (kl.Model.forward_multi_at): the real / fake / interpolated SPECTRA, so that the
        # 86-column samples the slice would cut them out of are never built (the critic reads nothing else: :57-59)
        self.node_spec_in = l_spec

        if cfgarch.arch_gen_nbcnnlayers > 0:
            # build extension (BASELINE configs[2]; the reference is fp32): cfgarch.arch_critic_bf16
            #   True      the whole stack in bf16 storage / bf16 products / fp32 accumulation, ONE launch per pass with the maps
            #             between the layers in the LDS (kl.Conv2DStack -> csrc/conv2d_chain.hip); 4 filters of 5x5, <= 8 layers,
            #             <= 68 bins -- other geometries take the layer-wise form below
            #   'layers'  the round-2 form: the maps between the 4 -> 4 channel layers (and their gradients) as bf16, one launch
            #             per layer and pass; the first layer (1 -> 4) and the map handed to the dense layers stay fp32
            # master weights and every weight gradient are fp32 either way
            bf16 = getattr(cfgarch, 'arch_critic_bf16', False)
            L = cfgarch.arch_gen_nbcnnlayers
            std = cfgarch.arch_gen_nbfilters == 4 and cfgarch.arch_gen_winlen == 5 and cfgarch.arch_spec_freqlen == 5
            if bf16 and bf16 != 'layers' and std and L <= 8 and vocoder.specsize() <= 68:
                l_spec = kl.Conv2DStack(L, cfgarch.arch_gen_nbfilters, [cfgarch.arch_gen_winlen, cfgarch.arch_spec_freqlen], alpha=0.3)(l_spec)
            else:
                l_spec = kl.Reshape([vocoder.specsize(), 1])(l_spec)
                for li in range(L):
                    conv = kl.Conv2D(cfgarch.arch_gen_nbfilters, [cfgarch.arch_gen_winlen, cfgarch.arch_spec_freqlen])
                    if bf16 and li >= 1 and std:
                        conv.bf16 = 'out16' if li < L - 1 else 'out32'
                    l_spec = conv(l_spec)
                    l_spec = kl.LeakyReLU(alpha=0.3)(l_spec)
            l_spec = kl.Reshape([l_spec.shape[-2] * l_spec.shape[-1]])(l_spec)
        else:
            for _ in range(3):
                l_spec = pFC(l_spec, cfgarch.arch_hiddenwidth, bn=bn)

        self.input_ctx = kl.Input(shape=(None, self.ctxsize), name='input_ctx')
        l_ctx = self.input_ctx
        for _ in range(cfgarch.arch_ctx_nbcnnlayers):
            l_ctx = pCNN1D(l_ctx, cfgarch.arch_hiddenwidth, cfgarch.arch_ctx_winlen, bn=bn)
        l_ctx = pFC(l_ctx, cfgarch.arch_hiddenwidth, bn=bn)
        l_ctx = pFC(l_ctx, cfgarch.arch_hiddenwidth, bn=bn)

        l_post = kl.Concatenate(name='lo_concatenation')([l_spec, l_ctx])
        for _ in range(3):
            l_post = pFC(l_post, cfgarch.arch_hiddenwidth, bn=bn)

        self.output = kl.Dense(1, activation=None)(l_post)

        self.model = kl.Model(inputs=[self.input_features, self.input_ctx], outputs=self.output)

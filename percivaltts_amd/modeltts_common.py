"""The acoustic models of the reference (percivaltts/modeltts_common.py:36-126) on the HIP layer graph."""
from __future__ import print_function

from . import layers as kl
from . import modeltts
from . import networktts


class Generic(modeltts.ModelTTS):
    """Stack of `network_generic` layers + the vocoder's output heads (modeltts_common.py:36-58)."""
    def __init__(self, ctxsize, vocoder, fmodel=None, layertypes=['FC', 'FC', 'FC'], nameprefix=None, cfgarch=None):
        modeltts.ModelTTS.__init__(self, ctxsize, vocoder)
        if nameprefix is None: nameprefix = ''
        l_in = kl.Input(shape=(None, ctxsize), name=nameprefix + 'input.conditional')
        l_out = networktts.network_generic(l_in, layertypes=layertypes, cfgarch=cfgarch)
        l_out = networktts.network_final(l_out, vocoder, mlpg_wins=vocoder.mlpg_wins)
        self.kerasmodel = kl.Model(inputs=l_in, outputs=l_out)
        if fmodel is not None:
            self.load(fmodel)
        self.kerasmodel.summary()


class DCNNF0SpecNoiseFeatures(modeltts.ModelTTS):
    """The PercivalTTS model (modeltts_common.py:61-126): context Conv1D + 2 FC, then
    f0: BLSTM -> Dense(1);  spec: Dense(F) -> L x (Conv2D + BN + LeakyReLU) -> Conv2D(1);  noise: L//2 FC -> sigmoid."""
    def __init__(self, ctxsize, vocoder, cfgarch, nameprefix=None):
        modeltts.ModelTTS.__init__(self, ctxsize, vocoder)
        if nameprefix is None: nameprefix = ''

        l_in = kl.Input(shape=(None, ctxsize), name=nameprefix + 'input.conditional')

        l_ctx = l_in
        for _ in range(cfgarch.arch_ctx_nbcnnlayers):
            l_ctx = networktts.pCNN1D(l_ctx, cfgarch.arch_hiddenwidth, cfgarch.arch_ctx_winlen)
        l_ctx = networktts.pFC(l_ctx, cfgarch.arch_hiddenwidth)
        l_ctx = networktts.pFC(l_ctx, cfgarch.arch_hiddenwidth)

        # F0
        l_f0 = networktts.pBLSTM(l_ctx, width=cfgarch.arch_hiddenwidth)
        l_f0.stream = 1           # independent, latency-bound branch: may run on a side HIP stream (layers.Model._run)
        l_f0 = kl.Dense(1, activation=None, use_bias=True)(l_f0)
        l_f0.stream = 1

        # Spec
        l_spec = kl.Dense(vocoder.specsize(), use_bias=True)(l_ctx)   # projection
        l_spec = kl.Reshape([vocoder.specsize(), 1])(l_spec)          # channels after the spectral dimension
        gated = bool(getattr(cfgarch, 'arch_gen_gated', False))       # build extension (BASELINE config 5)
        dils = getattr(cfgarch, 'arch_gen_dilations', None)
        for li in range(cfgarch.arch_gen_nbcnnlayers):
            if gated:
                kw = {'dil_t': dils[li % len(dils)] if dils else 1, 'causal': bool(getattr(cfgarch, 'arch_gen_causal', False))}
                l_spec = networktts.pGCNN2D(l_spec, cfgarch.arch_gen_nbfilters, cfgarch.arch_gen_winlen, cfgarch.arch_spec_freqlen, **kw)
            else:
                l_spec = networktts.pCNN2D(l_spec, cfgarch.arch_gen_nbfilters, cfgarch.arch_gen_winlen, cfgarch.arch_spec_freqlen)
        # (the causal variant keeps its output layer causal too, so that the whole spectral stack looks only backwards)
        l_spec = kl.Conv2D(1, [cfgarch.arch_gen_winlen, cfgarch.arch_spec_freqlen], use_bias=True, activation=None,
                           causal=gated and bool(getattr(cfgarch, 'arch_gen_causal', False)))(l_spec)
        l_spec = kl.Reshape([l_spec.shape[-2]])(l_spec)

        # NM
        l_nm = l_ctx
        for _ in range(cfgarch.arch_gen_nbcnnlayers // 2):   # integer division as in the Python-2 reference (:119)
            l_nm = networktts.pFC(l_nm, cfgarch.arch_hiddenwidth)
        l_nm = kl.Dense(vocoder.noisesize(), activation='sigmoid')(l_nm)

        l_out = kl.Concatenate()([l_f0, l_spec, l_nm])

        # handles for the critic step, which only needs the spectral branch (the critic slices it, networks_critic.py:58)
        self.node_spec, self.node_f0, self.node_nm = l_spec, l_f0, l_nm

        self.kerasmodel = kl.Model(inputs=l_in, outputs=l_out)
        self.kerasmodel.summary()

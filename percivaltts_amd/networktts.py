"""Layer builders of the acoustic models, same names and signatures as the reference's networktts.py:36-225,
built on percivaltts_amd.layers (HIP kernels) instead of tf.keras.

Each builder takes a graph Node and returns a Node, exactly like the Keras functional API the reference uses,
so modeltts_common / networks_critic read like the originals.  pFC / pCNN1D / pCNN2D are
Linear -> (BatchNorm) -> LeakyReLU(0.3); the BatchNorm-affine and the LeakyReLU are not executed as
layers of their own but handed to the next linear kernel (see ops.Lazy).
"""
from __future__ import print_function

from . import layers as kl
from . import vocoders
from .layers import GaussianNoiseInput   # re-exported: networktts.GaussianNoiseInput (modeltts.py:119)


def pFC(input, width, bn=True, **kwargs):
    """Dense(use_bias = not bn) -> BN -> LeakyReLU(.3)   (networktts.py:59-63)"""
    fc = kl.Dense(width, use_bias=not bn, **kwargs)
    fc.bn_follows = bool(bn)          # (the layer's launch also sums its output's columns for the BatchNormalization behind it: kl.Dense.compute)
    output = fc(input)
    if bn: output = kl.BatchNormalization()(output)
    return kl.LeakyReLU(alpha=0.3)(output)


def pDO(input, rate=0.2, batch_size=5):
    """Dropout with one mask per (sample, feature) shared along time (networktts.py:65-70)"""
    return kl.Dropout(rate=rate)(input)


def pLSTM(input, width, bn=False, cudnn=False, **kwargs):
    """kl.LSTM(tanh, recurrent sigmoid, return_sequences) (networktts.py:72-81); the CuDNN variant of the
    reference only differs by a second bias vector, there is a single implementation here."""
    if bn: print('WARNING: Batch normalisation can be unstable with LSTM layers')
    return kl.LSTM(width, bidirectional=False)(input)


def pRawLSTM(input, width, bn=False, **kwargs):
    return pLSTM(input, width, bn=bn, cudnn=False, **kwargs)


def pBLSTM(input, width, bn=False, cudnn=False, **kwargs):
    """kl.Bidirectional(kl.LSTM) with concat merge (networktts.py:85-96)"""
    if bn: print('WARNING: Batch normalisation can be unstable with BLSTM layers')
    return kl.LSTM(width, bidirectional=True)(input)


def pRawBLSTM(input, width, bn=False, **kwargs):
    return pBLSTM(input, width, bn=bn, cudnn=False, **kwargs)


def pGRU(input, width, bn=False, **kwargs):
    if bn: print('WARNING: Batch normalisation is not working for GRU layers (bug?)')
    return kl.GRU(width, bidirectional=False)(input)


def pBGRU(input, width, bn=False, **kwargs):
    if bn: print('WARNING: Batch normalisation is not working for BGRU layers (bug?)')
    return kl.GRU(width, bidirectional=True)(input)


def pCNN1D(input, nbfilters, winlen, bn=True, **kwargs):
    """Conv1D over time, 'same' -> BN -> LeakyReLU(.3) (networktts.py:116-120)"""
    output = kl.Conv1D(nbfilters, winlen, use_bias=not bn, **kwargs)(input)
    if bn: output = kl.BatchNormalization()(output)
    return kl.LeakyReLU(alpha=0.3)(output)


def pCNN2D(input, nbfilters, winlen, freqlen, bn=True, **kwargs):
    """Conv2D over (time, frequency), 'same' -> BN -> LeakyReLU(.3) (networktts.py:122-126)"""
    conv = kl.Conv2D(nbfilters, [winlen, freqlen], use_bias=not bn, **kwargs)
    conv.bn_follows = bool(bn)        # (the layer's launch also sums its outputs for the BatchNormalization behind it: kl.Conv2D.compute)
    output = conv(input)
    if bn: output = kl.BatchNormalization()(output)
    return kl.LeakyReLU(alpha=0.3)(output)


def pGCNN2D(input, nbfilters, winlen, freqlen, bn=True, **kwargs):
    """Gated conv: conv_a(x) * sigmoid(conv_b(x)) -> BN -> LeakyReLU (networktts.py:128-134).
    kwargs dil_t / causal are build extensions (BASELINE config 5); the defaults are the reference's layer."""
    output = kl.Conv2D(nbfilters, [winlen, freqlen], use_bias=not bn, **kwargs)(input)
    gate = kl.Conv2D(nbfilters, [winlen, freqlen], use_bias=not bn, **kwargs)(input)      # its sigmoid lives in the product
    output = kl.GatedMultiply()([output, gate])
    if bn: output = kl.BatchNormalization()(output)
    return kl.LeakyReLU(alpha=0.3)(output)


def network_generic(input, layertypes=['FC', 'FC', 'FC'], bn=True, cfgarch=None):
    """Stack of layers described by strings / lists / callables (networktts.py:136-179)."""
    simple = {
        'FC': lambda x: pFC(x, width=cfgarch.arch_hiddenwidth, bn=bn),
        'DO': lambda x: pDO(x, 0.2, batch_size=cfgarch.train_batch_size),
        'LSTM': lambda x: pLSTM(x, width=cfgarch.arch_hiddenwidth, bn=bn),
        'RawLSTM': lambda x: pRawLSTM(x, width=cfgarch.arch_hiddenwidth, bn=bn),
        'BLSTM': lambda x: pBLSTM(x, width=cfgarch.arch_hiddenwidth, bn=bn),
        'RawBLSTM': lambda x: pRawBLSTM(x, width=cfgarch.arch_hiddenwidth, bn=bn),
        'GRU': lambda x: pGRU(x, width=cfgarch.arch_hiddenwidth, bn=bn),
        'BGRU': lambda x: pBGRU(x, width=cfgarch.arch_hiddenwidth, bn=bn),
        'RND': lambda x: GaussianNoiseInput(width=cfgarch.arch_hiddenwidth)(x),
    }
    parametrised = {
        'FC': lambda x, a: pFC(x, a[1], bn=bn),
        'CNN1D': lambda x, a: pCNN1D(x, a[1], a[2], bn=bn),
        'CNN2D': lambda x, a: pCNN2D(x, a[1], a[2], a[3], bn=bn),
        'RND': lambda x, a: GaussianNoiseInput(width=a[1])(x),
    }
    l_out = input
    for lt in layertypes:
        if isinstance(lt, str) and lt in simple:
            l_out = simple[lt](l_out)
        elif isinstance(lt, (list, tuple)) and len(lt) > 0 and lt[0] in parametrised:
            l_out = parametrised[lt[0]](l_out, lt)
        elif callable(lt):
            l_out = lt(l_out)
        else:
            raise ValueError('Unknown layer type ' + str(lt))
    return l_out


def network_final(l_in, vocoder, mlpg_wins=None):
    """Output heads per vocoder (networktts.py:192-225): linear f0+spec, sigmoid noise mask, optional delta heads."""
    heads = []
    nwins = len(mlpg_wins) if mlpg_wins is not None else 0
    if isinstance(vocoder, vocoders.VocoderPML):
        heads.append(kl.Dense(1 + vocoder.spec_size, activation=None, name='lo_f0spec')(l_in))
        heads.append(kl.Dense(vocoder.nm_size, activation='sigmoid', name='lo_nm')(l_in))
        if nwins > 0:
            heads.append(kl.Dense(1 + vocoder.spec_size, activation=None, name='lo_delta_f0spec')(l_in))
            heads.append(kl.Dense(vocoder.nm_size, activation='tanh', name='lo_delta_nm')(l_in))
            if nwins > 1:
                heads.append(kl.Dense(1 + vocoder.spec_size, activation=None, name='lo_deltadelta_f0spec')(l_in))
                heads.append(kl.Dense(vocoder.nm_size, activation=('tanh_saturated', 2.0), name='lo_deltadelta_nm')(l_in))
    elif isinstance(vocoder, vocoders.VocoderWORLD):
        heads.append(kl.Dense(vocoder.featuressizeraw(), name='lo_f0specaper')(l_in))
        if nwins > 0:
            heads.append(kl.Dense(vocoder.featuressizeraw(), activation=None, name='lo_delta_f0specaper')(l_in))
            if nwins > 1:
                heads.append(kl.Dense(vocoder.featuressizeraw(), activation=None, name='lo_deltadelta_f0specaper')(l_in))
    if len(heads) == 1:
        return heads[0]
    return kl.Concatenate(name='lo_concatenation')(heads)

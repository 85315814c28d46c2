"""Experiment script with the shape of the reference's percivaltts/run.py:57-230: a module-level `cfg`,
`build_model()`, `training(cont)`, `generate()`.  The corpus-preparation stages of the reference
(features_extraction / contexts_extraction: vocoder analysis and HTS label normalisation, run.py:147-180) need
the pulsemodel/merlin front ends and are outside this build; `synthesize_corpus()` writes a synthetic corpus of
the same on-disk format (headerless float32 `path:(-1,D)` files + file_id_list.scp) so that the training stages run
unchanged.  Point `cfg.inpath / cfg.outpath / cfg.wpath / cfg.fileids` at real composed features to train on them.
"""
from __future__ import print_function

import os
import sys

import numpy as np

from percivaltts_amd import *  # noqa: F401,F403  (configuration, readids, print_log, ...)
from percivaltts_amd import modeltts_common, networks_critic, optimizertts, optimizertts_wgan, vocoders

print_log('Global configurations')
cfg = configuration()

# Corpus
cp = os.environ.get('PERCIVAL_CORPUS', os.path.join(os.getcwd(), 'corpus_synthetic')) + '/'
cfg.fileids = cp + 'file_id_list.scp'
cfg.id_valid_start = 40
cfg.id_valid_nb = 4
cfg.id_test_nb = 4

ctxsize = 416 + 9
cfg.inpath = cp + 'label_state_align_bin' + str(ctxsize) + '_norm_minmaxm11/*.lab:(-1,' + str(ctxsize) + ')'

cfg.vocoder_fs = 16000
cfg.vocoder_shift = 0.005
mlpg_wins = None
vocoder = vocoders.VocoderPML(cfg.vocoder_fs, cfg.vocoder_shift, spec_size=129, nm_size=33, mlpg_wins=mlpg_wins)

errtype = 'WLSWGAN'   # 'LSE', 'WGAN', 'WLSWGAN'

out_size = vocoder.featuressize()
cfg.outpath = cp + 'wav_PML_cmp_lf0_fwlspec' + str(vocoder.specsize()) + '_fwnm' + str(vocoder.noisesize()) + '_nmnoscale/*.cmp:(-1,' + str(out_size) + ')'
cfg.wpath = cp + 'label_state_align_weights/*.w:(-1,1)'

# Model architecture (run.py:114-120)
cfg.arch_hiddenwidth = 256
cfg.arch_ctx_nbcnnlayers = 1
cfg.arch_ctx_winlen = 21
cfg.arch_gen_nbcnnlayers = 8
cfg.arch_gen_nbfilters = 4
cfg.arch_gen_winlen = 5
cfg.arch_spec_freqlen = 5

# Training (run.py:122-137)
cfg.fparams_fullset = 'model.h5'
cfg.train_batch_size = 10
cfg.train_batch_lengthmax = int(2.0 / 0.005)
cfg.train_wgan_LScoef = 0.25
cfg.train_min_nbepochs = 250
cfg.train_max_nbepochs = 300
cfg.train_cancel_nodecepochs = 25
cfg.train_wgan_critic_LSWGANtransfreqcutoff = 4000
cfg.train_wgan_critic_LSWGANtranscoef = 1.0 / 8.0
cfg.train_wgan_critic_use_WGAN_incnoisefeature = False


def synthesize_corpus(nfiles=48, minlen=450, maxlen=700, seed=123):
    """Synthetic composed features: labels ~ U(-1,1), f0+spec ~ N(0,1), noise mask ~ U(0,1), weights 1 with silent ends."""
    rng = np.random.RandomState(seed)
    fids = ['syn_{:04d}'.format(i) for i in range(nfiles)]
    for path in (cfg.inpath, cfg.outpath, cfg.wpath):
        makedirs(os.path.dirname(path.split(':')[0]))
    with open(cfg.fileids, 'w') as f:
        f.write('\n'.join(fids) + '\n')
    for fid in fids:
        n = int(rng.randint(minlen, maxlen))
        (rng.rand(n, ctxsize) * 2 - 1).astype(np.float32).tofile(cfg.inpath.split(':')[0].replace('*', fid))
        y = rng.randn(n, out_size).astype(np.float32)
        y[:, 1 + vocoder.specsize():] = rng.rand(n, vocoder.noisesize())
        y.tofile(cfg.outpath.split(':')[0].replace('*', fid))
        w = np.ones((n, 1), dtype=np.float32)
        w[:20] = 0.0; w[-20:] = 0.0
        w.tofile(cfg.wpath.split(':')[0].replace('*', fid))
    return fids


def build_model():
    return modeltts_common.DCNNF0SpecNoiseFeatures(ctxsize, vocoder, cfg)    # DCNN in the paper


def training(cont=False):
    fids = readids(cfg.fileids)
    fid_lst_tra = fids[:cfg.id_train_nb()]
    fid_lst_val = fids[cfg.id_valid_start:cfg.id_valid_start + cfg.id_valid_nb]
    mod = build_model()
    if errtype == 'LSE': opti = optimizertts.OptimizerTTS(cfg, mod)
    else:                opti = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype=errtype, critic=networks_critic.Critic(vocoder, ctxsize, cfg))
    opti.train(cfg.inpath, cfg.outpath, cfg.wpath, fid_lst_tra, fid_lst_val, cfg.fparams_fullset, cont=cont)
    del mod


def generate(fparams=None):
    fparams = cfg.fparams_fullset if fparams is None else fparams
    fids = readids(cfg.fileids)
    mod = build_model()
    mod.load(fparams)
    fid_lst_test = fids[cfg.id_valid_start + cfg.id_valid_nb:cfg.id_valid_start + cfg.id_valid_nb + cfg.id_test_nb]
    mod.generate_cmp(cfg.inpath, os.path.splitext(fparams)[0] + '-gen/*.cmp', fid_lst_test)


if __name__ == "__main__":
    if not os.path.exists(cfg.fileids):
        synthesize_corpus()
    training(cont='--continue' in sys.argv)
    generate()

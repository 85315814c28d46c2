"""ModelTTS: the acoustic-model wrapper (reference: percivaltts/modeltts.py:46-205).

Holds `.ctxsize .vocoder .kerasmodel`; `kerasmodel` keeps its name for drop-in compatibility but is a
percivaltts_amd.layers.Model running on HIP kernels.  predict / count_params / save / load follow
modeltts.py:68-130; the model file trio keeps its stems (`.arch.json`, `.weights.npz` in place of `.weights.h5`
because h5py is not part of this image, `.cfgextras.pkl`).  generate_wav needs the vocoder DSP and stays
out of scope.
"""
from __future__ import print_function

import os
import pickle
import sys

import numpy as np
import torch

from . import backend_hip
from . import data
from . import networktts


class ModelTTS:

    ctxsize = -1
    vocoder = None
    kerasmodel = None

    def __init__(self, ctxsize, vocoder, kerasmodel=None):
        print("Building the TTS-dedicated model")
        self.ctxsize = ctxsize
        self.vocoder = vocoder
        if kerasmodel is not None:
            self.kerasmodel = kerasmodel
            self.kerasmodel.summary()

    def to_device(self):
        dev = backend_hip.device()
        p = next(self.kerasmodel.parameters(), None)
        if p is None or p.device != dev:
            self.kerasmodel.to(dev)
        return dev

    def predict(self, x):
        """Inference forward (BatchNorm uses its moving statistics): numpy [B,T,ctx] -> numpy [B,T,out]."""
        dev = self.to_device()
        with torch.no_grad():
            xt = torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).to(dev)
            y = self.kerasmodel(xt, training=False)
        return y.cpu().numpy()

    def count_params(self):
        return self.kerasmodel.count_params()

    def save(self, fmodel, cfg=None, extras=None, printfn=print, infostr=''):
        if extras is None: extras = dict()
        printfn('    saving parameters in {} ...'.format(fmodel), end='')
        sys.stdout.flush()
        with open(fmodel + '.arch.json', 'w') as f:
            f.write(self.kerasmodel.to_json())
        ws = self.kerasmodel.weights()
        np.savez(fmodel + '.weights.npz', **{'w{:04d}'.format(i): t.detach().cpu().numpy() for i, (_, t) in enumerate(ws)})
        with open(fmodel + '.cfgextras.pkl', 'wb') as f:
            pickle.dump([cfg, extras], f)
        print(' done ' + infostr)
        sys.stdout.flush()

    def load(self, fmodel, printfn=print, compile=True):
        printfn('    reloading parameters from {} ...'.format(fmodel), end='')
        sys.stdout.flush()
        if self.kerasmodel is None:
            raise ValueError('the architecture has to be rebuilt from source before loading weights '
                             '(the .arch.json of this build is descriptive only)')
        with np.load(fmodel + '.weights.npz') as z:
            arrays = [z['w{:04d}'.format(i)] for i in range(len(z.files))]
        self.kerasmodel.set_weights(arrays)
        with open(fmodel + '.cfgextras.pkl', 'rb') as f:
            DATA = pickle.load(f)
        print(' done')
        sys.stdout.flush()
        return DATA

    def generate_cmp(self, inpath, outpath, fid_lst):
        """Write the raw network output of each file as headerless float32 [T,out] (modeltts.py:133-141)."""
        if not os.path.isdir(os.path.dirname(outpath)): os.mkdir(os.path.dirname(outpath))
        X = data.load(inpath, fid_lst, verbose=1, label='Context labels: ')
        for vi in range(len(fid_lst)):
            CMP = self.predict(np.reshape(X[vi], [1] + [s for s in X[vi].shape]))[0,]
            CMP.astype('float32').tofile(outpath.replace('*', fid_lst[vi]))

    def generate_wav(self, *args, **kwargs):
        raise NotImplementedError('waveform synthesis needs the vocoder DSP (pulsemodel/pyworld), outside this build')

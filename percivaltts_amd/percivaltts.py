"""Pipeline-wide helpers of the MI355X build: the `configuration` attribute bag and a few logging utilities.

Mirrors the API surface of the reference's percivaltts/percivaltts.py:51-168 that the WGAN hot path and
its callers touch (configuration :57-104, readids :51-54, print_log/print_tty :107-117, time2str :123-129,
is_int :132-136, makedirs :146-153, weights_normal_ortho :157-168, proc_memresident :173-179).  Python 3.
"""
from __future__ import print_function

import errno
import os
import random as rn
import runpy
import sys
import time

import numpy as np


def numpy_force_random_seed(seed=123):
    """The reference seeds numpy and `random` with 123 at import (percivaltts.py:30-33)."""
    np.random.seed(seed)
    rn.seed(seed)


numpy_force_random_seed()


def readids(fileids):
    """Non-empty, stripped lines of a file-id list."""
    with open(fileids) as f:
        return [ln.strip() for ln in f if ln.strip()]


class configuration(object):
    """Mutable bag of attributes carried along the pipeline (`cfg`)."""

    def __eq__(self, other):
        return isinstance(other, configuration) and self.__dict__ == other.__dict__

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def id_train_nb(self):
        """Size of the training set: the largest multiple of the batch size below id_valid_start."""
        return self.train_batch_size * int(self.id_valid_start // self.train_batch_size)

    def _public_items(self):
        for key in sorted(self.__dict__):
            if not key.startswith('__'):
                yield key, self.__dict__[key]

    def print_content(self):
        hypers = set(h[0] for h in getattr(self, 'train_hypers', []) or [])
        for key, val in self._public_items():
            note = '    (Attention! hyper-parameter optimized during multi-training)' if key in hypers else ''
            print('    {:<30}{}{}'.format(key, val, note))
        print('')

    def mergefiles(self, filenames):
        """Run python file(s) and take their globals as configuration values."""
        if not isinstance(filenames, (list, tuple)):
            filenames = [filenames]
        for fname in filenames:
            for k, v in runpy.run_path(fname).items():
                setattr(self, k, v)

    def merge(self, cfgtoadd):
        for k, v in cfgtoadd.__dict__.items():
            if not k.startswith('__'):
                setattr(self, k, v)


def datetime2str(sec=None):
    return time.strftime('%Y-%m-%d %H:%M:%S', time.gmtime(sec))


def time2str(sec=None):
    days = int(sec // 86400)
    hms = time.strftime('%H:%M:%S', time.gmtime(sec))
    return '{}d{}'.format(days, hms) if days > 0 else hms


def print_log(txt, end='\n'):
    print(datetime2str() + ': ' + txt, end=end)
    sys.stdout.flush()


def print_tty(txt, end=''):
    if sys.stdout.isatty():
        print(txt, end=end)
        sys.stdout.flush()


def is_int(v):
    s = str(v).strip()
    if s in ('0', '+0', '-0'):
        return True
    s = s.lstrip('+-')
    if '.' in s:
        whole, _, frac = s.partition('.')
        return whole.isdigit() and set(frac) <= {'0'} and '..' not in str(v)
    return s.isdigit()


def makedirs(path):
    try:
        os.makedirs(path)
    except OSError as e:
        if e.errno != errno.EEXIST:
            raise


def weights_normal_ortho(insiz, outsiz, std, rng, dtype):
    """Orthogonal initialisation from the SVD of a Gaussian matrix."""
    a = rng.normal(0.0, std, size=(insiz, outsiz))
    u, _, vt = np.linalg.svd(a, full_matrices=False)
    q = u if u.shape == (insiz, outsiz) else vt
    return np.asarray(q.reshape((insiz, outsiz)), dtype=dtype)


def proc_memresident():
    """Resident memory of this process [MiB], -1 if unknown."""
    try:
        with open('/proc/self/status') as f:
            for ln in f:
                if ln.startswith('VmRSS:'):
                    return int(ln.split()[1]) // 1024
    except (IOError, OSError):
        pass
    return -1

"""Device backend glue (counterpart of the reference's backend_tensorflow.py:27-69,107-109)."""
import numpy as np
import torch

from . import _hip


def hip_available():
    return torch.cuda.is_available()


def device():
    """The one GPU this process owns (LOCAL_RANK picks it under torch.distributed.run; the reference picked a free
    NVIDIA GPU with GPUtil, run.py:28-31)."""
    import os
    if not torch.cuda.is_available():
        raise _hip.HipLibraryError('no MI355X visible: percivaltts_amd has no CPU compute path')
    idx = int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()
    torch.cuda.set_device(idx)
    return torch.device('cuda', idx)


def set_random_seed(seed=123):
    np.random.seed(seed)
    torch.manual_seed(seed)


def gpu_memused():
    """MiB of device memory held by this process (tf_gpu_memused, backend_tensorflow.py:59-69)."""
    if not torch.cuda.is_available():
        return -1
    return int(torch.cuda.max_memory_allocated() // (1024 * 1024))


def nonlin_sigmoidparm(x, c=0.0, f=1.0):
    """Parametrised sigmoid: centre c, slope f (backend_tensorflow.py:107-109)."""
    return 1.0 / (1.0 + np.exp(-(np.asarray(x) - c) * f))

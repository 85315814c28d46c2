"""percivaltts_amd -- MI355X-native WGAN-GP training hot path of percivaltts behind the reference's Python API.

`from percivaltts_amd import *` gives what `from percivaltts import *` gives the reference's scripts
(configuration, readids, print_log, ...; percivaltts/percivaltts.py); the sub-modules keep the reference's names:
modeltts, modeltts_common, networktts, networks_critic, optimizertts, optimizertts_wgan, data, vocoders.
All arithmetic runs in libpercival_hip.so (include/percival_hip.h); there is no CPU compute path.
"""
from .percivaltts import (configuration, readids, print_log, print_tty, datetime2str, time2str, is_int, makedirs,
                          weights_normal_ortho, proc_memresident, numpy_force_random_seed)

__all__ = ['configuration', 'readids', 'print_log', 'print_tty', 'datetime2str', 'time2str', 'is_int', 'makedirs',
           'weights_normal_ortho', 'proc_memresident', 'numpy_force_random_seed']
__version__ = '0.1.0'

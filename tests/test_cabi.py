"""CPU tests: the C-ABI library loads (no compute) and exports every symbol include/percival_hip.h declares,
and the ctypes signature table of percivaltts_amd/_hip.py covers exactly that set."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'percival_hip.h')


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ptts_\w+)\s*\(', src)))


def test_header_declares_the_hot_path_entry_points():
    syms = declared_symbols()
    for must in ('ptts_conv2d_fwd', 'ptts_conv2d_bwd', 'ptts_gemm', 'ptts_gp_interpolate', 'ptts_gp_sqnorm',
                 'ptts_gp_penalty', 'ptts_weight_clip', 'ptts_adam_keras_step', 'ptts_lstm_fwd', 'ptts_lstm_bwd',
                 'ptts_colstats', 'ptts_bn_finalize', 'ptts_version', 'ptts_device_arch', 'ptts_last_error'):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from percivaltts_amd import _hip
    if not os.path.exists(_hip.LIB_PATH):
        pytest.skip('libpercival_hip.so not built (run __graft_entry__.build())')
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), 'missing export ' + s
    assert set(_hip.SIGNATURES) == set(declared_symbols())
    lib.ptts_device_arch.restype = ctypes.c_char_p
    assert lib.ptts_device_arch() == b'gfx950'


def test_missing_library_or_cpu_tensor_fails_loudly(monkeypatch):
    import torch
    from percivaltts_amd import _hip, ops
    with pytest.raises(_hip.HipLibraryError):
        ops.gp_interpolate(torch.zeros(2, 3, 4), torch.zeros(2, 3, 4), torch.zeros(2))   # CPU tensors: no fallback
    monkeypatch.setattr(_hip, '_lib', None)
    monkeypatch.setattr(_hip, 'LIB_PATH', '/nonexistent/libpercival_hip.so')
    with pytest.raises(_hip.HipLibraryError):
        _hip.lib()


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, 'percivaltts_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt, f

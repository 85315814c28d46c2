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


def test_device_status_word_is_sticky_and_decoded():
    """The host side of the device status word (include/percival_hip.h: ptts_device_status*): a code a kernel would store -- written
    here through the word's host address, no GPU involved -- makes ptts_device_status return PTTS_EDEVICE with a message naming the
    kernel family, stays until ptts_device_status_clear(), and _hip.check_status() raises."""
    from percivaltts_amd import _hip
    if not os.path.exists(_hip.LIB_PATH):
        pytest.skip('libpercival_hip.so not built (run __graft_entry__.build())')
    lib = _hip.lib()
    word = ctypes.cast(lib.ptts_device_status_word(), ctypes.POINTER(ctypes.c_uint))
    _hip.clear_status()
    out = ctypes.c_uint(99)
    assert lib.ptts_device_status(ctypes.byref(out)) == 0 and out.value == 0
    _hip.check_status()
    for slot, code, text in ((0, 1, 'conv2d'), (1, 2, 'LSTM')):
        word[slot] = code
        assert lib.ptts_device_status(ctypes.byref(out)) == -4 and out.value == code        # PTTS_EDEVICE
        assert text in _hip.last_error()
        assert lib.ptts_device_status(None) == -4                                           # sticky
        with pytest.raises(_hip.HipLibraryError, match='hand-off|hidden state'):
            _hip.check_status()
        _hip.clear_status()
        assert lib.ptts_device_status(ctypes.byref(out)) == 0 and out.value == 0
    word[0] = 1; word[1] = 2
    assert lib.ptts_device_status(ctypes.byref(out)) == -4 and out.value == 3
    buf = ctypes.create_string_buffer(400)
    assert lib.ptts_device_status_message(3, buf, 400) == 0 and b'conv2d' in buf.value and b'LSTM' in buf.value
    assert lib.ptts_device_status_message(0, buf, 400) == 0 and buf.value == b'ok'
    _hip.clear_status()


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

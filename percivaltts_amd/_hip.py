"""ctypes binding of libpercival_hip.so (the C ABI declared in include/percival_hip.h).

There is NO fallback: if the library is missing or a kernel reports an error the call raises.
The reference reaches its arithmetic through tf.keras (percivaltts/backend_tensorflow.py:38-39
opens the TF session); this module is the counterpart for the MI355X build.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PTTS_LIB_PATH') or os.path.join(_HERE, 'lib', 'libpercival_hip.so')      # (PTTS_LIB_PATH: an A/B build of the library, tools/ab_define.sh)

c_f = ctypes.c_float
c_i = ctypes.c_int
c_ll = ctypes.c_longlong
c_p = ctypes.c_void_p
c_sz = ctypes.c_size_t

class WGradDesc(ctypes.Structure):
    """struct ptts_wgrad_desc of include/percival_hip.h (one weight-gradient product of a grouped launch)."""
    _fields_ = [('A', c_p), ('B', c_p), ('C', c_p), ('colsum_b', c_p), ('in_scale', c_p), ('in_shift', c_p), ('mask_src', c_p),
                ('M', c_i), ('N', c_i), ('K', c_i), ('lda', c_ll), ('ldb', c_ll), ('ldc', c_ll), ('in_mode', c_i), ('alpha', c_f)]


class DenseWgradReduceDesc(ctypes.Structure):
    """struct ptts_dense_wgrad_reduce_desc of include/percival_hip.h (the partial rows of one weight-gradient product)."""
    _fields_ = [('partials', c_p), ('split', c_i), ('Kin', c_i), ('N', c_i), ('ldc', c_ll), ('C', c_p), ('colsum_b', c_p)]


class DenseSplitDesc(ctypes.Structure):
    """struct ptts_dense_split_desc of include/percival_hip.h (one weight of a grouped plane split)."""
    _fields_ = [('w', c_p), ('planes', c_p), ('ldw', c_ll), ('K', c_i), ('N', c_i), ('transposed', c_i), ('reserved', c_i)]


class Conv2dReduceDesc(ctypes.Structure):
    """struct ptts_conv2d_reduce_desc of include/percival_hip.h (one queued conv2d backward pass)."""
    _fields_ = [('partials', c_p), ('nblocks', c_i), ('npart', c_i), ('nw', c_i), ('cout', c_i), ('dw', c_p), ('dbias', c_p)]


# name -> (restype, argtypes); must mirror include/percival_hip.h exactly
SIGNATURES = {
    'ptts_version': (ctypes.c_char_p, []),
    'ptts_device_arch': (ctypes.c_char_p, []),
    'ptts_last_error': (ctypes.c_char_p, []),
    'ptts_device_status': (c_i, [c_p]),
    'ptts_device_status_clear': (c_i, []),
    'ptts_device_status_word': (c_p, []),
    'ptts_device_status_message': (c_i, [ctypes.c_uint, ctypes.c_char_p, c_sz]),
    'ptts_set_deterministic': (c_i, [c_i]),
    'ptts_get_deterministic': (c_i, []),
    'ptts_set_bf16_products': (c_i, [c_i]),
    'ptts_get_bf16_products': (c_i, []),
    'ptts_conv2d_fwd': (c_i, [c_p] * 7 + [c_i] * 10 + [c_f, c_p]),
    'ptts_conv2d_bwd_workspace_bytes': (c_sz, [c_i] * 8),
    'ptts_conv2d_bwd': (c_i, [c_p] * 11 + [c_p, c_sz] + [c_i] * 10 + [c_f, c_p]),
    'ptts_gemm': (c_i, [c_p] * 4 + [c_i] * 3 + [c_i, c_ll, c_ll, c_ll, c_i, c_ll, c_ll, c_i, c_p, c_p, c_p, c_f, c_i, c_p, c_p, c_p]),
    'ptts_gemm_wgrad_grouped': (c_i, [c_p, c_i, c_p]),
    'ptts_split3_frames': (c_i, [c_p] * 4 + [c_i] * 6 + [c_p]),
    'ptts_split3_weight_t': (c_i, [c_p] * 4 + [c_i] * 4 + [c_p]),
    'ptts_conv1d_bf16x6': (c_i, [c_p] * 8 + [c_i] * 5 + [c_p]),
    'ptts_split3_frames_t': (c_i, [c_p] * 4 + [c_i] * 6 + [c_ll, c_p]),
    'ptts_conv1d_wgrad_bf16x6': (c_i, [c_p] * 7 + [c_i] * 6 + [c_ll, c_p]),
    'ptts_transpose_frames': (c_i, [c_p] * 2 + [c_i] * 6 + [c_ll, c_p]),
    'ptts_conv1d_wgrad_t': (c_i, [c_p] * 4 + [c_i] * 6 + [c_ll, c_p]),
    'ptts_conv2d_bwd_partials': (c_i, [c_p] * 5 + [c_p, c_sz, c_p] + [c_i] * 10 + [c_f, c_p]),
    'ptts_conv2d_reduce_grouped': (c_i, [c_p, c_i, c_p]),
    'ptts_conv2d_mfma_debug': (c_i, [c_i, c_p]),
    'ptts_conv2d_mfma_table_bytes': (c_sz, [c_i]),
    'ptts_conv2d_mfma_tables': (c_i, [c_p, c_p, c_p] + [c_i] * 5 + [c_p]),
    'ptts_conv2d_mfma_tables_grouped': (c_i, [c_p, c_p, c_p, c_i, c_i, c_p]),
    'ptts_conv2d_mfma_supported': (c_i, [c_i] * 6),
    'ptts_conv2d_mfma_fwd': (c_i, [c_p] * 8 + [c_i] * 7 + [c_f] + [c_i] * 3 + [c_p]),
    'ptts_conv2d_mfma_fwd_stats_supported': (c_i, [c_i, c_i, c_i]),
    'ptts_conv2d_mfma_fwd_stats': (c_i, [c_p] * 6 + [c_i] * 6 + [c_f] + [c_p, c_i, c_p, c_p]),
    'ptts_conv2d_mfma_wgrad_workspace_bytes': (c_sz, [c_i, c_i]),
    'ptts_conv2d_mfma_wgrad_partials': (c_i, [c_p] * 4 + [c_sz, c_p, c_p] + [c_i] * 7 + [c_f] + [c_i] * 3 + [c_p]),
    'ptts_conv2d_mfma_bwd_fused_workspace_bytes': (c_sz, [c_i, c_i]),
    'ptts_conv2d_mfma_bwd_fused_supported': (c_i, [c_i, c_i, c_i]),
    'ptts_conv2d_mfma_bwd_fused_affine': (c_i, [c_p] * 5 + [c_sz, c_p, c_p] + [c_i] * 5 + [c_f, c_p, c_p, c_p]),
    'ptts_conv2d_mfma_bwd_fused': (c_i, [c_p] * 6 + [c_sz, c_p, c_p] + [c_i] * 6 + [c_f, c_p]),
    'ptts_conv2d_chain_supported': (c_i, [c_i] * 6),
    'ptts_conv2d_chain_debug': (c_i, [c_p]),
    'ptts_conv2d_chain_tables_bytes': (c_sz, []),
    'ptts_conv2d_chain_partials_bytes': (c_sz, [c_i]),
    'ptts_conv2d_chain_map_elems': (c_ll, [c_i] * 3),
    'ptts_conv2d_chain_tables': (c_i, [c_p, c_p, c_p, c_i, c_i, c_p]),
    'ptts_conv2d_chain_fwd': (c_i, [c_p, c_ll, c_p, c_p, c_p] + [c_i] * 4 + [c_f, c_p]),
    'ptts_conv2d_chain_bwd': (c_i, [c_p, c_i, c_p, c_ll, c_p, c_p, c_p, c_p, c_sz, c_p, c_p] + [c_i] * 5 + [c_f, c_p]),
    'ptts_conv2d_chain_bwd_data': (c_i, [c_p, c_i, c_p, c_p, c_p, c_p, c_p] + [c_i] * 4 + [c_f, c_p]),
    'ptts_conv2d_chain_second': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_sz, c_p, c_p] + [c_i] * 5 + [c_f, c_p]),
    'ptts_dense_planes_bytes': (c_sz, [c_i, c_i]),
    'ptts_split3_dense_weight': (c_i, [c_p, c_ll, c_i, c_i, c_i, c_p, c_p]),
    'ptts_split3_dense_weight_grouped': (c_i, [c_p, c_i, c_p]),
    'ptts_split3_frame_windows': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_ll, c_p]),
    'ptts_split3_dense_weight_strided': (c_i, [c_p, c_ll, c_p, c_ll, c_i, c_ll, c_i, c_i, c_i, c_p]),
    'ptts_dense_bf16x6_supported': (c_i, [c_i, c_i, c_i, c_ll, c_ll]),
    'ptts_dense_bf16x6': (c_i, [c_p] * 4 + [c_i] * 3 + [c_ll, c_ll, c_i, c_p, c_p, c_p, c_f, c_i, c_p, c_p]),
    'ptts_dense_bf16x6_stats_rows': (c_i, [c_i, c_i]),
    'ptts_dense_bf16x6_stats': (c_i, [c_p] * 4 + [c_i] * 3 + [c_ll, c_ll, c_i, c_p, c_p, c_f, c_p, c_i, c_p, c_p]),
    'ptts_dense_bf16x6_bwd_affine': (c_i, [c_p] * 3 + [c_i] * 3 + [c_ll, c_ll, c_p, c_p, c_p, c_f, c_p, c_i, c_p, c_p]),
    'ptts_partial_rows_sum': (c_i, [c_p, c_i, c_i, c_p, c_p]),
    'ptts_dense_bf16x6_res': (c_i, [c_p] * 4 + [c_i] * 3 + [c_ll, c_ll, c_i, c_p, c_p, c_p, c_f, c_p, c_i, c_ll, c_p, c_p]),
    'ptts_conv1d_freq_kernel_planes': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'ptts_transpose_batched': (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    'ptts_conv1d_freq_wgrad_combine': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    'ptts_conv1d_freq_wgrad_inverse_workspace_bytes': (c_sz, [c_i, c_i, c_i]),
    'ptts_conv1d_freq_wgrad_inverse': (c_i, [c_p, c_p, c_p, c_p, c_sz, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'ptts_dft_mirror': (c_i, [c_p, c_i, c_i, c_i, c_i, c_p]),
    'ptts_dense_bf16x6_batched': (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_p, c_ll, c_i, c_i, c_i, c_i, c_ll, c_ll, c_i, c_p]),
    'ptts_dense_wgrad_bf16x6_supported': (c_i, [c_i, c_i, c_i, c_ll, c_ll]),
    'ptts_dense_wgrad_workspace_bytes': (c_sz, [c_i, c_i, c_i]),
    'ptts_dense_wgrad_bf16x6_partials': (c_i, [c_p] * 6 + [c_sz, c_p] + [c_i] * 3 + [c_ll] * 2 + [c_i, c_f, c_p]),
    'ptts_dense_wgrad_reduce_grouped': (c_i, [c_p, c_i, c_p]),
    'ptts_dense_wgrad_bf16x6': (c_i, [c_p] * 8 + [c_sz] + [c_i] * 3 + [c_ll] * 3 + [c_i, c_f, c_p]),
    'ptts_colstats_workspace_bytes': (c_sz, [c_ll, c_i]),
    'ptts_colstats': (c_i, [c_p, c_ll, c_i, c_i, c_p, c_p, c_p, c_f, c_p, c_p, c_sz, c_p]),
    'ptts_bn_finalize': (c_i, [c_p, c_ll, c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    'ptts_bn_batch_stats_supported': (c_i, [c_ll, c_i]),
    'ptts_bn_finalize_partials': (c_i, [c_p, c_i, c_ll, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    'ptts_bn_batch_stats': (c_i, [c_p, c_ll, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_sz, c_p, c_p]),
    'ptts_bn_bwd_coefs': (c_i, [c_p] * 5 + [c_ll, c_i] + [c_p] * 4 + [c_p]),
    'ptts_bn_bwd_coefs_acc': (c_i, [c_p] * 5 + [c_ll, c_i] + [c_p] * 4 + [c_p]),
    'ptts_affine_act': (c_i, [c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_f, c_p]),
    'ptts_affine_act_bwd': (c_i, [c_p] * 7 + [c_p, c_sz, c_ll, c_i, c_i, c_f, c_p]),
    'ptts_gated_mul_fwd': (c_i, [c_p] * 3 + [c_ll, c_p]),
    'ptts_gated_mul_bwd': (c_i, [c_p] * 5 + [c_ll, c_p]),
    'ptts_axpby_cols': (c_i, [c_p] * 6 + [c_ll, c_i, c_p]),
    'ptts_gp_interpolate': (c_i, [c_p] * 4 + [c_i, c_ll, c_p]),
    'ptts_gp_sqnorm': (c_i, [c_p, c_p, c_i, c_ll, c_p]),
    'ptts_gp_penalty': (c_i, [c_p, c_p, c_p, c_i, c_p]),
    'ptts_gp_scale_rows': (c_i, [c_p] * 4 + [c_i, c_ll, c_p]),
    'ptts_mean_scaled': (c_i, [c_p, c_ll, c_f, c_p, c_p]),
    'ptts_wlse_fwd': (c_i, [c_p] * 4 + [c_ll, c_i, c_p]),
    'ptts_wlse_bwd': (c_i, [c_p] * 5 + [c_ll, c_i, c_p]),
    'ptts_weight_clip': (c_i, [c_p, c_ll, c_f, c_f, c_p]),
    'ptts_adam_keras_step': (c_i, [c_p] * 4 + [c_ll] + [c_f] * 5 + [c_p, c_p]),
    'ptts_lstm_fwd_workspace_bytes': (c_sz, [c_i] * 4),
    'ptts_lstm_fwd': (c_i, [c_p] * 5 + [c_p, c_sz] + [c_i] * 5 + [c_p]),
    'ptts_lstm_bwd_workspace_bytes': (c_sz, [c_i] * 4),
    'ptts_lstm_bwd': (c_i, [c_p] * 5 + [c_p, c_sz] + [c_i] * 5 + [c_p]),
    'ptts_set_lstm_graph': (c_i, [c_i]),
    'ptts_lstm_graph_stats': (c_i, [c_p] * 3),
    'ptts_lstm_graph_clear': (c_i, []),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def lib():
    """Load the library once; raise loudly if it is absent (no CPU path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                'libpercival_hip.so not found at {}: run `python -c "import __graft_entry__ as g; g.build()"` '
                '(or `make -C percivaltts_amd/csrc`). There is no CPU fallback.'.format(LIB_PATH))
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)   # AttributeError if the symbol is missing -> loud
            fn.restype = res
            fn.argtypes = args
        if os.environ.get('PTTS_DETERMINISTIC', '0') == '1':
            l.ptts_set_deterministic(1)
        _lib = l
    return _lib


def last_error():
    return lib().ptts_last_error().decode('utf8', 'replace')


class KernelTimer(object):
    """HIP-event timing of every C-ABI call made while it is active (bench.py's roofline leg).  Events are recorded on
    the stream the kernels are launched on (torch's current stream).  Not usable during graph capture."""
    active = None

    def __init__(self):
        self.records = []   # (name, tag, start_event, end_event)

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None

    def durations_ms(self):
        torch.cuda.synchronize()
        return [(n, t, s.elapsed_time(e)) for n, t, s, e in self.records]


def call(name, *args, **kw):
    """Invoke an int-returning entry point and raise on a non-zero status.  `tag` labels the call for KernelTimer."""
    timer = KernelTimer.active
    if timer is not None:
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
    rc = getattr(lib(), name)(*args)
    if timer is not None:
        e.record()
        timer.records.append((name, kw.get('tag'), s, e))
    if rc != 0:
        raise HipLibraryError('{} failed (rc={}): {}'.format(name, rc, last_error()))


def check_status():
    """Raise HipLibraryError if a kernel has reported a failed hand-off since the last clear_status() (the sticky device status
    word of include/percival_hip.h).  A host memory load, no synchronisation: the optimiser calls it at every step boundary."""
    rc = lib().ptts_device_status(None)
    if rc != 0:
        raise HipLibraryError('device status (rc={}): {}'.format(rc, last_error()))


def clear_status():
    lib().ptts_device_status_clear()


def stream_id():
    """Raw handle (an integer) of the current HIP stream of the current device.  torch.cuda.current_stream() builds a Stream object
    through several layers of Python (7 us a call, 150 calls per training step: a millisecond of a host-bound step); the raw getter
    is the same lookup without them."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def stream():
    return ctypes.c_void_p(stream_id())


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be fp32/fp64/int32 CUDA memory."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipLibraryError('percivaltts_amd kernels need device tensors (got a {} tensor); there is no CPU path'
                              .format(t.device))
    return ctypes.c_void_p(t.data_ptr())


def f32c(t, name='tensor'):
    """Validate: CUDA, float32, contiguous."""
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise HipLibraryError('{}: expected a contiguous float32 device tensor, got {} {} contiguous={}'.format(
            name, t.device, t.dtype, t.is_contiguous()))
    return t

"""ctypes binding of libsrgan_hip.so (include/srgan_hip.h).  There is no fallback: if the library is
missing or a call fails, an exception is raised."""
import ctypes
import os

from ._build import LIBRARY, source_id

c_float_p = ctypes.c_void_p
i32, i64, f32, vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    """Mirror of ``srgan_conv_desc``."""
    _fields_ = [(name, i32) for name in ('N', 'C', 'H', 'W', 'K', 'R', 'S', 'stride_h', 'stride_w', 'pad_h', 'pad_w',
                                         'OH', 'OW')] + [('x_batch_stride', i64), ('y_batch_stride', i64), ('compute_dtype', i32)]


class BnRelu(ctypes.Structure):
    """Mirror of ``srgan_bn_relu``: frozen batch-norm statistics and parameters of a fused norm -> relu -> conv."""
    _fields_ = [(name, vp) for name in ('mean', 'inv_std', 'gamma', 'beta')]


class BnReduceJob(ctypes.Structure):
    """Mirror of ``srgan_bn_reduce_job``."""
    _fields_ = [('partial_offset', i64), ('tiles', i32), ('channels', i32), ('inv_std', vp), ('g_gamma', vp), ('g_beta', vp)]


class Capabilities(ctypes.Structure):
    """Mirror of ``srgan_capabilities_t``."""
    _fields_ = [('abi_version', i32), ('struct_bytes', i32), ('arch', ctypes.c_char * 16), ('dtypes', ctypes.c_uint32),
                ('features', ctypes.c_uint32), ('workspace_bytes', i64), ('max_tensor_elements', i64)]


SIGNATURES = {
    'srgan_version': ([], ctypes.c_int),
    'srgan_build_id': ([], ctypes.c_char_p),
    'srgan_last_error': ([], ctypes.c_char_p),
    'srgan_conv2d_fwd': ([ctypes.POINTER(ConvDesc), vp, vp, vp, vp, ctypes.c_int, vp], ctypes.c_int),
    'srgan_conv2d_bwd_data': ([ctypes.POINTER(ConvDesc), vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, vp],
                              ctypes.c_int),
    'srgan_conv2d_bwd_weight': ([ctypes.POINTER(ConvDesc), vp, vp, vp, ctypes.c_int, ctypes.c_int, vp], ctypes.c_int),
    'srgan_conv2d_bnrelu_supported': ([ctypes.POINTER(ConvDesc), ctypes.c_int], ctypes.c_int),
    'srgan_conv2d_fwd_bnrelu': ([ctypes.POINTER(ConvDesc), vp, ctypes.POINTER(BnRelu), vp, vp, vp, vp], ctypes.c_int),
    'srgan_conv2d_fwd_bnrelu_into_zeros': ([ctypes.POINTER(ConvDesc), vp, ctypes.POINTER(BnRelu), vp, vp, vp, vp],
                                           ctypes.c_int),
    'srgan_conv2d_fwd_bnrelu_splits': ([ctypes.POINTER(ConvDesc)], ctypes.c_int),
    'srgan_bn_conv_tangent_weights': ([vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp], ctypes.c_int),
    'srgan_conv2d_bwd_data_bnrelu': ([ctypes.POINTER(ConvDesc), vp, vp, ctypes.POINTER(BnRelu), vp, vp, vp, vp,
                                      ctypes.c_int, vp], ctypes.c_int),
    'srgan_conv2d_bwd_data_bnrelu_tiles': ([ctypes.POINTER(ConvDesc)], ctypes.c_int64),
    'srgan_conv2d_bwd_data_bnrelu_partials': ([ctypes.POINTER(ConvDesc), vp, vp, ctypes.POINTER(BnRelu), vp, vp, vp,
                                               ctypes.c_int, vp], ctypes.c_int),
    'srgan_bn_partial_reduce_batched': ([vp, i32, i32, i32, vp, vp], ctypes.c_int),
    'srgan_conv2d_bwd_weight_bnrelu': ([ctypes.POINTER(ConvDesc), vp, ctypes.POINTER(BnRelu), vp, vp, ctypes.c_int, vp],
                                       ctypes.c_int),
    'srgan_gemm_f32': ([i32, i32, i32, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i32, ctypes.c_int, ctypes.c_int,
                        vp], ctypes.c_int),
    'srgan_gemm': ([i32, i32, i32, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i32, ctypes.c_int, ctypes.c_int,
                    ctypes.c_int, vp], ctypes.c_int),
    'srgan_ew_unary': ([ctypes.c_int, vp, vp, i64, f32, f32, vp], ctypes.c_int),
    'srgan_ew_binary': ([ctypes.c_int, vp, vp, vp, i64, f32, vp], ctypes.c_int),
    'srgan_fill': ([vp, i64, f32, vp], ctypes.c_int),
    'srgan_chan_affine': ([vp, vp, vp, vp, vp, vp, i32, i32, i64, vp], ctypes.c_int),
    'srgan_chan_affine_act': ([vp, vp, vp, vp, vp, vp, ctypes.c_int, vp, i32, i32, i64, vp], ctypes.c_int),
    'srgan_chan_affine_act_strided': ([vp, vp, vp, vp, vp, vp, ctypes.c_int, vp, i32, i32, i64, i64, i64, i64,
                                       ctypes.c_int, vp], ctypes.c_int),
    'srgan_bn_act_bwd': ([vp, vp, vp, vp, vp, vp, ctypes.c_int, vp, vp, vp, i32, i32, i64, i64, i64, i64, ctypes.c_int,
                          ctypes.c_int, vp],
                         ctypes.c_int),
    'srgan_chan_reduce': ([vp, vp, vp, vp, vp, i32, i32, i64, ctypes.c_int, vp], ctypes.c_int),
    'srgan_row_max': ([vp, vp, i32, i32, vp], ctypes.c_int),
    'srgan_nearest_bin_onehot': ([vp, vp, vp, i32, i32, vp], ctypes.c_int),
    'srgan_copy_channels': ([vp, i32, i32, vp, i32, i32, i32, i32, i64, ctypes.c_int, vp], ctypes.c_int),
    'srgan_bn_relu_maxpool_fwd': ([vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp], ctypes.c_int),
    'srgan_bn_relu_maxpool_bwd_supported': ([i32, i32, i32, i32, i32, i32, i32], ctypes.c_int),
    'srgan_bn_relu_maxpool_bwd': ([vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
                                  ctypes.c_int),
    'srgan_bn_relu_avgpool2_supported': ([i32, i32, i32, i32], ctypes.c_int),
    'srgan_bn_relu_avgpool2_fwd': ([vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp], ctypes.c_int),
    'srgan_bn_relu_avgpool2_bwd': ([vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp], ctypes.c_int),
    'srgan_maxpool2d_fwd': ([vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp], ctypes.c_int),
    'srgan_maxpool2d_bwd': ([vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp], ctypes.c_int),
    'srgan_pool_scatter': ([vp, vp, vp, i32, i64, i64, vp], ctypes.c_int),
    'srgan_pool_gather': ([vp, vp, vp, i32, i64, i64, vp], ctypes.c_int),
    'srgan_avgpool2d_fwd': ([vp, vp, i32, i32, i32, i32, i32, i32, i32, vp], ctypes.c_int),
    'srgan_avgpool2d_bwd': ([vp, vp, i32, i32, i32, i32, i32, i32, i32, vp], ctypes.c_int),
    'srgan_gp_interpolate': ([vp, vp, vp, vp, i32, i64, vp], ctypes.c_int),
    'srgan_crowd_map_l1_fwd': ([vp, vp, vp, i32, i32, i64, vp], ctypes.c_int),
    'srgan_crowd_map_l1_bwd': ([vp, vp, vp, vp, i32, i32, i64, vp], ctypes.c_int),
    'srgan_profile_begin': ([], ctypes.c_int),
    'srgan_profile_end': ([ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                           ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)], ctypes.c_int),
    'srgan_profile_report': ([ctypes.c_char_p, ctypes.c_int64], ctypes.c_int64),
    'srgan_profile_bytes': ([ctypes.POINTER(ctypes.c_double)], ctypes.c_int),
    'srgan_profile_mixed': ([ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)], ctypes.c_int),
    'srgan_capabilities': ([ctypes.POINTER(Capabilities), i32], ctypes.c_int),
    'srgan_workspace_bytes': ([], ctypes.c_int64),
    'srgan_set_workspace': ([vp, i64, vp], ctypes.c_int),
    'srgan_split_is_ordered': ([vp], ctypes.c_int),
    'srgan_crowd_density_label': ([vp, i32, i32, i32, f32, vp, i32, vp, vp, vp], ctypes.c_int),
    'srgan_crowd_iknn_map': ([vp, i32, i32, i32, i32, f32, f32, vp, vp], ctypes.c_int),
    'srgan_crowd_extract_patches': ([vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp], ctypes.c_int),
    'srgan_bn_conv_tangent_weights_job': ([vp, vp, vp, vp, vp, i64, i32, i32, i32, vp], ctypes.c_int),
    'srgan_bn_conv_tangent_weights_grouped': ([vp, i32, i32, i32, vp, vp, vp], ctypes.c_int),
    'srgan_wgrad_group_plan': ([ctypes.POINTER(ConvDesc), ctypes.POINTER(BnRelu), i64, i64, vp, i64, i32, i64, i64, vp, ctypes.POINTER(i32),
                                ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i64)], ctypes.c_int),
    'srgan_wgrad_group_run': ([vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, i64, i64, i64, i64, vp], ctypes.c_int),
    'srgan_adam_step': ([vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp], ctypes.c_int),
    'srgan_adam_step_counted': ([vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, vp], ctypes.c_int),
    'srgan_pack_bf16': ([vp, vp, i64, vp], ctypes.c_int),
    'srgan_unpack_bf16': ([vp, vp, i64, vp], ctypes.c_int),
    'srgan_h_pack': ([vp, vp, vp, f32, i32, i32, i64, ctypes.c_int, vp], ctypes.c_int),
    'srgan_h_unpack': ([vp, vp, i32, i32, i64, ctypes.c_int, vp], ctypes.c_int),
    'srgan_h_add': ([vp, vp, vp, i64, ctypes.c_int, vp], ctypes.c_int),
    'srgan_h_channel_sums': ([vp, vp, i32, i32, i64, ctypes.c_int, vp], ctypes.c_int),
    'srgan_h_maxpool2': ([vp, vp, vp, i64, i32, i32, ctypes.c_int, ctypes.c_int, vp], ctypes.c_int),
    'srgan_h_maxpool2_bwd': ([vp, vp, vp, i64, i32, i32, ctypes.c_int, f32, ctypes.c_int, vp], ctypes.c_int),
    'srgan_h_conv_weight_slots': ([i32, i32, i32, i32], ctypes.c_int64),
    'srgan_h_pack_conv_weights': ([vp, vp, i32, i32, i32, i32, ctypes.c_int, ctypes.c_int, vp], ctypes.c_int),
    'srgan_h_conv3x3': ([vp, vp, vp, vp, f32, ctypes.c_int, vp, i32, i32, i32, i32, i32, i32, ctypes.c_int, vp], ctypes.c_int),
    'srgan_h_conv3x3_wgrad': ([vp, vp, vp, i32, i32, i32, i32, i32, ctypes.c_int, vp], ctypes.c_int),
    'srgan_h_pack_matrix': ([vp, vp, i64, i64, i64, i64, i64, i64, i32, i32, ctypes.c_int, vp], ctypes.c_int),
    'srgan_h_gemm': ([vp, vp, vp, vp, f32, ctypes.c_int, vp, i32, i32, i32, i32, i32, ctypes.c_int, vp], ctypes.c_int),
    'srgan_h_linear_wgrad': ([vp, vp, vp, i32, i32, i32, i64, i64, i64, i64, i32, i32, ctypes.c_int, vp], ctypes.c_int),
    'srgan_h_k4s2_weight_slots': ([i32, i32, ctypes.c_int, ctypes.c_int], ctypes.c_int64),
    'srgan_h_pack_k4s2_weights': ([vp, vp, i32, i32, ctypes.c_int, ctypes.c_int, vp], ctypes.c_int),
    'srgan_h_pack_job_bytes': ([], ctypes.c_int32),
    'srgan_h_pack_job_conv_weights': ([vp, i64, vp, vp, i32, i32, i32, i32, ctypes.c_int, ctypes.c_int], ctypes.c_int64),
    'srgan_h_pack_job_k4s2_weights': ([vp, i64, vp, vp, i32, i32, ctypes.c_int, ctypes.c_int, vp], ctypes.c_int64),
    'srgan_h_pack_batched': ([vp, i32, i64, vp], ctypes.c_int),
    'srgan_h_conv4x4s2': ([vp, vp, vp, vp, f32, ctypes.c_int, vp, i32, i32, i32, i32, i32, ctypes.c_int, vp], ctypes.c_int),
    'srgan_h_conv_transpose4x4s2': ([vp, vp, vp, vp, f32, ctypes.c_int, vp, i32, i32, i32, i32, i32, ctypes.c_int, vp], ctypes.c_int),
    'srgan_h_k4s2_wgrad': ([vp, vp, vp, i32, i32, i32, i32, i32, ctypes.c_int, ctypes.c_int, vp], ctypes.c_int),
    'srgan_comm_available': ([], ctypes.c_int),
    'srgan_comm_unique_id': ([vp], ctypes.c_int),
    'srgan_comm_init': ([ctypes.POINTER(vp), i32, i32, vp], ctypes.c_int),
    'srgan_comm_world_size': ([vp, ctypes.POINTER(i32)], ctypes.c_int),
    'srgan_comm_destroy': ([vp], ctypes.c_int),
    'srgan_all_reduce_sum': ([vp, vp, vp, i64, i32, vp], ctypes.c_int),
    'srgan_reduce_scatter_sum': ([vp, vp, vp, i64, i32, vp], ctypes.c_int),
    'srgan_all_gather': ([vp, vp, vp, i64, i32, vp], ctypes.c_int),
    'srgan_broadcast': ([vp, vp, i64, i32, i32, vp], ctypes.c_int),
}


EINVAL, EUNSUPPORTED, ERANGE = -1, -2, -3      # negative status codes of include/srgan_hip.h


class HipLibraryError(RuntimeError):
    """A failed library call; ``status`` is the C ABI's return code (< 0 argument errors, > 0 hipError_t)."""

    def __init__(self, message, status=None):
        super().__init__(message)
        self.status = status


_library = None


def library():
    """The loaded shared object with typed entry points; raises if it has not been built."""
    global _library
    if _library is None:
        if not os.path.exists(LIBRARY):
            raise HipLibraryError(f'{LIBRARY} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                                  '(there is no CPU fallback)')
        import torch  # noqa: F401  -- first: loads PyTorch-ROCm's HIP runtime (SONAME libamdhip64.so.7) that we bind to
        lib = ctypes.CDLL(LIBRARY)
        for name, (argtypes, restype) in SIGNATURES.items():
            function = getattr(lib, name)      # AttributeError if the ABI lost a symbol
            function.argtypes = argtypes
            function.restype = restype
        built_from = lib.srgan_build_id().decode()
        if built_from != source_id() and os.environ.get('SRGAN_ALLOW_STALE_LIBRARY') != '1':
            raise HipLibraryError(f'{LIBRARY} was built from other kernel sources (library {built_from}, tree '
                                  f'{source_id()}): rebuild with `python -c "import __graft_entry__ as g; g.build()"`')
        _library = lib
    return _library


def capabilities():
    out = Capabilities()
    check(library().srgan_capabilities(ctypes.byref(out), ctypes.sizeof(out)), 'srgan_capabilities')
    return out


_workspaces = {}      # (device index, stream handle) -> the torch tensor registered as that stream's split-K workspace


def stream_handle(stream=None):
    """The hipStream_t of ``stream`` (default: torch's current stream), with the caller-owned split-K workspace of
    include/srgan_hip.h registered for it on first use (one block of srgan_workspace_bytes() per device and stream; the
    library itself never allocates).  Called once per launch: the raw-handle query avoids building a Stream object."""
    import torch
    if stream is None:
        device = torch._C._cuda_getDevice()
        handle = torch._C._cuda_getCurrentRawStream(device)
    else:
        device, handle = stream.device.index, stream.cuda_stream
    if (device, handle) not in _workspaces:
        lib = library()
        with torch.cuda.device(device):
            block = torch.empty(lib.srgan_workspace_bytes() // 4, dtype=torch.float32, device=torch.device('cuda', device))
            check(lib.srgan_set_workspace(block.data_ptr(), block.numel() * 4, handle), 'srgan_set_workspace')
        _workspaces[(device, handle)] = block
    return handle


def check(status, what):
    if status != 0:
        message = library().srgan_last_error()
        raise HipLibraryError(f'{what} failed with status {status}: {message.decode() if message else ""}', status)

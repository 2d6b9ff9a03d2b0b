"""ctypes binding of libsed_hip.so (include/sed_hip.h).  There is NO fallback: if the shared library
is missing or a call fails, a RuntimeError is raised -- the product path never computes on the CPU."""
from __future__ import annotations

import ctypes as C
import os

SED_F32, SED_BF16, SED_F32X3, SED_F32H3 = 0, 1, 2, 3
PRO_NONE, PRO_BNRELU = 0, 1
EPI_STORE, EPI_STATS, EPI_RELUBWD, EPI_POOLSTATS = 0, 1, 2, 4
DZ_POOL, DZ_BN = 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsed_hip.so")

_P, _I, _Z, _F, _D = C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_double



class GemmTnDesc(C.Structure):
    """struct sed_gemm_tn_desc (include/sed_hip.h): one problem of sed_gemm_tn_batch"""
    _fields_ = [("A", _P), ("B", _P), ("C", _P), ("colsum", _P), ("workspace", _P), ("lda", _I), ("ldb", _I), ("ldc", _I), ("M", _I),
                ("N", _I), ("K", _I), ("seq", _I), ("shift", _I), ("ksplit", _I)]


# name -> (restype, argtypes); mirrors include/sed_hip.h one to one
PROTOTYPES = {
    "sed_abi_version": (_I, []),
    "sed_last_error": (C.c_char_p, []),
    "sed_build_flags": (_I, []),
    "sed_device_cu_count": (_I, []),
    "sed_config_reload": (None, []),
    "sed_pack_conv_weight": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sed_unpack_conv_wgrad": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "sed_conv_c1_nparts": (_I, [_I, _I, _I]),
    "sed_conv3x3_c1_fwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sed_conv3x3_c1_wgrad": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "sed_conv3x3_c1_wgrad_fused": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "sed_conv_c1_gram_nparts": (_I, [_I, _I, _I]),
    "sed_conv3x3_c1_gram": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "sed_conv3x3_c1_wgrad_combine": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _P]),
    "sed_c1_mode_supported": (_I, [_I, _I, _I, _I]),
    "sed_bn_train_finalize_c1": (_I, [_P, _I, _D, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _I, _I, _P]),
    "sed_bn_train_finalize_c1_g": (_I, [_P, _I, _D, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _I, _I, _P, _P]),
    "sed_c1_bwd_tail": (_I, [_P, _I, _P, _D, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "sed_conv3x3_fwd_c1": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "sed_conv3x3_dgrad_c1": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "sed_conv_dgrad_c1_nparts": (_I, []),
    "sed_conv3x3_dgrad_c1_stats": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "sed_conv3x3_dgrad_c1_stats_g": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "sed_bn_bwd_finalize_c1": (_I, [_P, _I, _D, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "sed_conv3x3_wgrad_fused_c1": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I,
                                        _I, _I, _P]),
    "sed_pack_conv_weights_batch": (_I, [_I, _P, _I, _I, _P]),
    "sed_conv3x3_bwd_fused_c1_supported": (_I, [_I, _I, _I, _I]),
    "sed_conv3x3_bwd_fused_c1": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I,
                                      _P, _I, _I, _P]),
    "sed_conv3x3_bwd_fused_supported": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "sed_conv3x3_dgrad_dz_supported": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "sed_conv3x3_dgrad_dz": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I,
                                  _I, _I, _P]),
    "sed_conv3x3_wgrad_u": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P]),
    "sed_conv3x3_bwd_fused_supported_pool": (_I, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "sed_conv3x3_bwd_fused": (_I, [_I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I,
                                   _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P]),
    "sed_conv3x3_wgrad_fused_u": (_I, [_I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I,
                                       _I, _I, _P, _I, _I, _P]),
    "sed_conv3x3_wgrad_fused_c1_u": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I,
                                          _I, _I, _P, _I, _I, _P]),
    "sed_conv3x3_c1_wgrad_combine_u": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "sed_conv_nparts": (_I, [_I, _I, _I]),
    "sed_conv3x3_fwd": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sed_conv3x3_fwd_col": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sed_conv_wgrad_ws_floats": (_Z, [_I, _I, _I, _I, _I]),
    "sed_conv3x3_wgrad": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sed_conv3x3_wgrad_fused": (_I, [_I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I,
                                     _I, _I, _P]),
    "sed_bn_train_finalize": (_I, [_P, _I, _D, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _I, _I, _P]),
    "sed_bn_eval_coeffs": (_I, [_P, _P, _P, _P, _F, _P, _P, _I, _I, _P]),
    "sed_bn_bwd_finalize": (_I, [_P, _I, _D, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "sed_bn_relu_pool_fwd": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sed_pool_bwd_nparts": (_I, [_I, _I, _I, _I]),
    "sed_pool_relu_bwd_stats": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sed_bn_relu_pool_cnt_fwd": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "sed_dgrad_poolstats_supported": (_I, [_I, _I, _I, _I]),
    "sed_conv3x3_dgrad_poolstats": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "sed_pool_relu_bwd_stats_if": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "sed_pool_relu_bn_bwd_apply": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sed_bn_bwd_apply": (_I, [_I, _P, _P, _P, _P, _P, _P, _Z, _I, _P]),
    "sed_head_fwd": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "sed_interpolate": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "sed_bce_fwd_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P]),
    "sed_head_bwd_ws_floats": (_Z, [_I, _I, _I, _I]),
    "sed_head_bwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sed_adam_amsgrad_step": (_I, [_P, _P, _P, _P, _P, _Z, _F, _F, _F, _F, _I, _F, _P]),
    "sed_adam_amsgrad_step_dev": (_I, [_P, _P, _P, _P, _P, _Z, _P, _P, _F, _F, _F, _F, _F, _I, _P]),
    "sed_logmel_ws_bytes": (_Z, [_I, _I, _I, _I]),
    "sed_logmel_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sed_stft_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "sed_complex_to_logmel": (_I, [_P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _P]),
    "sed_complex_augment_logmel": (_I, [_P, _Z, _P, _P, _P, _P, _P, _P, C.c_ulonglong, _P, _P, _P, _P, _P, _P, _I, _I,
                                        _I, _I, _P]),
    "sed_logmel_crops": (_I, [_P, _Z, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "sed_mel_mean_fwd": (_I, [_I, _P, _P, _Z, _I, _I, _I, _P]),
    "sed_mel_mean_bwd": (_I, [_I, _P, _P, _Z, _I, _I, _I, _P]),
    "sed_gemm_nt_ws_floats": (_Z, [_I, _I, _I]),
    "sed_gemm_nt": (_I, [_I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "sed_gemm_tn_ws_floats": (_Z, [_I, _I, _I]),
    "sed_gemm_tn": (_I, [_I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "sed_gemm_tn_batch": (_I, [_I, _P, _I, _P]),
    "sed_transpose_shift": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "sed_row_sums": (_I, [_P, _I, _P, _I, _I, _P]),
    "sed_gru_pack_elems": (_Z, [_I]),
    "sed_gru_pack_weights": (_I, [_I, _P, _P, _P, _P, _I, _P]),
    "sed_gru_seq_fwd": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "sed_gru_seq_bwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "sed_metric_counts_ws_bytes": (_Z, [_I]),
    "sed_metric_counts": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _Z, _Z, _I, _P]),
    "sed_m5_conv1_len": (_I, [_I]),
    "sed_m5_conv1_nparts": (_I, [_I, _I]),
    "sed_m5_conv1_fwd": (_I, [_I, _P, _P, _P, _P, _I, _I, _P]),
    "sed_m5_conv1_wgrad": (_I, [_I, _P, _P, _P, _I, _I, _P]),
    "sed_m5_conv1_wgrad_fused": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "sed_m5_conv1_wgrad_fused_pool": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "sed_m5_zfree_supported": (_I, [_I]),
    "sed_m5_alg_supported": (_I, [_I]),
    "sed_m5_conv1_gram_floats": (_Z, []),
    "sed_m5_conv1_gram": (_I, [_P, _P, _I, _I, _P]),
    "sed_m5_conv1_bwd_stats_g1": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "sed_m5_conv1_wgrad_combine": (_I, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "sed_m5_conv1_stats": (_I, [_I, _P, _P, _P, _I, _I, _P]),
    "sed_m5_conv1_bn_relu_pool_fwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "sed_m5_fwd2_supported": (_I, [_I]),
    "sed_m5_conv1_pool_bwd_stats": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "sed_m5_conv1_wgrad_fused_pool_x": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "sed_bn_relu_maxpool4_fwd": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "sed_maxpool4_bwd_nparts": (_I, [_I, _I, _I, _I]),
    "sed_maxpool4_relu_bwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "sed_maxpool4_pooled_stats": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "sed_maxpool4_relu_bwd_if": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "sed_m5_head_fwd": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sed_m5_head_bwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sed_sum_partials": (_I, [_P, _I, _Z, _P, _P]),
    "sed_cast": (_I, [_I, _P, _I, _P, _Z, _P]),
    "sed_nchw_to_nhwc": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sed_nhwc_to_nchw": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sed_wgrad_last_slabs": (_I, []),
    "sed_wgrad_reduce": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "sed_wgrad_reduce_batch": (_I, [_P, _I, _I, _P]),
    "sed_peak_mfma_bf16": (_I, [_I, _P, _P, _P]),
    "sed_peak_stream_copy": (_I, [_P, _P, _Z, _P]),
    "sed_peak_stream_read": (_I, [_P, _Z, _P, _P]),
}

_lib = None


def lib():
    """Load (once) and return the CDLL.  Raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the MI355X kernels are not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
                f"{os.path.join(_HERE, 'csrc')}`). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)          # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        if L.sed_abi_version() != 1:
            raise RuntimeError("libsed_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().sed_last_error()
        raise RuntimeError(f"libsed_hip {what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()

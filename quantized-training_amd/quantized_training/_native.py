"""ctypes binding of libqt_hip.so -- the C ABI declared in include/qt_hip.h.

The library is the product: there is no Python or torch fallback for device tensors.
`lib()` raises if the shared object is missing (build it with `make -C quantized-training_amd`
or `python __graft_entry__.py`).
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_long, c_size_t, c_uint16, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# QT_HIP_LIB: tools/ only -- load the tuning build (make -C quantized-training_amd tuning) instead of the product library.  Never
# silent: a warning names the file, and `LIB_OVERRIDDEN` lets callers (bench.py's config) record it.
LIB_OVERRIDDEN = bool(os.environ.get("QT_HIP_LIB"))
LIB_PATH = os.environ.get("QT_HIP_LIB") or os.path.join(_HERE, "libqt_hip.so")
if LIB_OVERRIDDEN:
    import warnings
    warnings.warn(f"quantized_training loads {LIB_PATH} instead of its own libqt_hip.so (QT_HIP_LIB is set): a tuning build's ablation "
                  f"switches produce wrong results on purpose", RuntimeWarning, stacklevel=2)

QT_MAP_ENTRIES = 65536
QT_FMT_LUT, QT_FMT_IDENTITY, QT_FMT_FP_SAT, QT_FMT_INT = 0, 1, 2, 3
QT_ERR_BAD_DTYPE, QT_ERR_BAD_ARG, QT_ERR_UNALIGNED, QT_ERR_NO_DEVICE = -1, -2, -3, -4


class QtFormat(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("p0", ctypes.c_int32), ("p1", ctypes.c_int32),
                ("flo", c_float), ("fhi", c_float)]

    def key(self):
        return (self.kind, self.p0, self.p1, self.flo, self.fhi)


class QtOperandQ(ctypes.Structure):
    _fields_ = [("fmt", QtFormat), ("lut_dev", c_void_p), ("scale_f32_dev", c_void_p), ("amax_bits_dev", c_void_p)]


class QtFaninItem(ctypes.Structure):
    """qt_fanin_item of include/qt_hip.h"""
    _fields_ = [("x_dev", ctypes.c_void_p), ("fq", ctypes.c_int), ("scale_f32_dev", ctypes.c_void_p), ("amax_bits_dev", ctypes.c_void_p),
                ("out_dev", ctypes.c_void_p)]


class QtGemmProblem(ctypes.Structure):
    """qt_gemm_problem of include/qt_hip.h"""
    _fields_ = [("a", ctypes.c_void_p), ("b", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("c", ctypes.c_void_p)]


class QtAdamwTensor(ctypes.Structure):
    """qt_adamw_tensor of include/qt_hip.h"""
    _fields_ = [("param_dev", c_void_p), ("grad_dev", c_void_p), ("exp_avg_dev", c_void_p), ("exp_avg_sq_dev", c_void_p), ("step_dev", c_void_p),
                ("lr_dev", c_void_p), ("numel", c_long), ("first_chunk", c_long), ("lr", ctypes.c_double), ("beta1", ctypes.c_double),
                ("beta2", ctypes.c_double), ("eps", ctypes.c_double), ("weight_decay", ctypes.c_double), ("step", ctypes.c_double)]


class QtLinearBackward(ctypes.Structure):
    """qt_linear_backward of include/qt_hip.h"""
    _fields_ = [("gy", c_void_p), ("wq", c_void_p), ("x", c_void_p), ("gx", c_void_p), ("gw", c_void_p)]


class QtChainStage(ctypes.Structure):
    _fields_ = [("scale_f32_dev", c_void_p), ("amax_bits_dev", c_void_p), ("out_dev", c_void_p), ("src", ctypes.c_int)]


class QtRowParams(ctypes.Structure):
    """Row form of a value map (qt_build_rowparams): 512 rows {D, C | flagged bit, lo, hi}, see include/qt_hip.h."""
    _fields_ = [("row", (c_uint32 * 4) * 512), ("signed_rows", ctypes.c_int32), ("sign_mask", c_uint32),
                ("n_flagged", ctypes.c_int32), ("zero_sign", ctypes.c_int32), ("flagged", ctypes.c_uint8 * 512)]


class QtError(RuntimeError):
    pass


_P = c_void_p
_FMT = POINTER(QtFormat)
_OPQ = POINTER(QtOperandQ)

# name -> (restype, argtypes); mirrors include/qt_hip.h one to one
ABI_VERSION = 3          # include/qt_hip.h QT_ABI_VERSION

SIGNATURES = {
    "qt_abi_version": (c_int, []),
    "qt_status_string": (c_char_p, [c_int]),
    "qt_stream_capture_id": (c_int, [_P, POINTER(ctypes.c_ulonglong)]),
    "qt_build_map": (c_int, [c_char_p, _P]),
    "qt_format_for": (c_int, [c_char_p, _FMT]),
    "qt_format_apply_host": (c_uint16, [_FMT, c_uint16]),
    "qt_round_fp8_host": (c_int, [_P, _P, c_size_t, c_int, c_float, c_float]),
    "qt_round_posit_host": (c_int, [_P, _P, c_size_t, c_int, c_int]),
    "qt_round_fp8_f32": (c_int, [_P, _P, c_size_t, c_int, c_float, c_float, _P]),
    "qt_round_posit_f32": (c_int, [_P, _P, c_size_t, c_int, c_int, _P]),
    "qt_posit_quantize_host": (c_int, [_P, _P, _P, c_size_t, c_int, c_int, c_int]),
    "qt_posit_quantize_f32": (c_int, [_P, _P, _P, c_size_t, c_int, c_int, c_int, _P]),
    "qt_vmap_bf16": (c_int, [_P, _P, c_size_t, _FMT, _P, _P]),
    "qt_vmap_f32": (c_int, [_P, _P, c_size_t, _FMT, _P, _P]),
    "qt_vmap_f16": (c_int, [_P, _P, c_size_t, _FMT, _P, _P]),
    "qt_quantize_bf16": (c_int, [_P, _P, c_size_t, _FMT, _P, _P, _P, _P]),
    "qt_quantize_f32": (c_int, [_P, _P, c_size_t, _FMT, _P, _P, _P, _P]),
    "qt_dequantize_bf16": (c_int, [_P, _P, c_size_t, _P, _P, _P, _P, _P]),
    "qt_dequantize_f32": (c_int, [_P, _P, c_size_t, _P, _P, _P, _P, _P]),
    "qt_scale_update": (c_int, [_P, c_int, c_int, _P, c_float, c_int, _P]),
    "qt_scale_update_multi": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P]),
    "qt_fake_quant_bf16": (c_int, [_P, _P, c_size_t, _FMT, _P, _P, _P, _P]),
    "qt_fake_quant_f32": (c_int, [_P, _P, c_size_t, _FMT, _P, _P, _P, _P]),
    "qt_fake_quant_bf16_fp8": (c_int, [_P, _P, _P, c_size_t, _FMT, _P, _P, _P]),
    "qt_fake_quant_rows_bf16": (c_int, [_P, _P, c_long, c_long, c_long, c_long, c_long, c_long, c_long, _FMT, _P, _P, _P, _P]),
    "qt_fake_quant_mx_bf16": (c_int, [_P, _P, _P, c_size_t, c_size_t, c_int, _FMT, _P, c_float, _P, _P]),
    "qt_fake_quant_mx_f32": (c_int, [_P, _P, _P, c_size_t, c_size_t, c_int, _FMT, _P, c_float, _P, _P]),
    "qt_fake_quant_pc_bf16": (c_int, [_P, _P, c_size_t, c_size_t, c_size_t, _FMT, _P, _P, _P, _P]),
    "qt_fake_quant_pc_f32": (c_int, [_P, _P, c_size_t, c_size_t, c_size_t, _FMT, _P, _P, _P, _P]),
    "qt_linear_fq_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _OPQ, _OPQ, _P]),
    "qt_train_gemm_bf16": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_long, c_long, c_long, _P]),
    "qt_train_gemm_backward_bf16": (c_int, [_P, c_int, c_int, c_int, c_int, c_long, c_long, c_long, c_long, c_long, _P]),
    "qt_linear_fq8_plan": (c_int, [c_int, c_long, c_int, c_int, _P, _P, _P, _P, _P]),
    "qt_clip_adamw_plan": (c_long, [_P, c_int, _P, c_long]),
    "qt_clip_adamw_ws_bytes": (c_size_t, [c_int, c_long]),
    "qt_clip_adamw_bf16": (c_int, [_P, _P, c_int, c_long, c_float, _P, _P, c_size_t, c_int, _P]),
    "qt_linear_fq8_bf16": (c_int, [_P, c_int, _P, _P, _P, c_int, c_int, _P, c_int, c_int, _P]),
    "qt_fake_quant_chain_bf16": (c_int, [_P, c_long, c_long, POINTER(QtChainStage), c_int, _FMT, _P, c_int, c_float, _P, _P, c_size_t, _P]),
    "qt_fake_quant_chain_ws_bytes": (c_size_t, [c_long, c_long]),
    "qt_gelu_chain_bf16": (c_int, [_P, _P, c_long, c_long, POINTER(QtChainStage), c_int, _FMT, _P, _P]),
    "qt_gelu_backward_chain_bf16": (c_int, [_P, _P, _P, c_long, c_long, POINTER(QtChainStage), c_int, _FMT, _P, c_int, c_float, _P, _P, c_size_t, _P]),
    "qt_layernorm_train_bf16": (c_int, [_P, _P, _P, _P, _P, _P, c_long, c_long, c_float, POINTER(QtChainStage), c_int, _FMT, _P, _P, _P, _P]),
    "qt_layernorm_train_backward_groups": (c_long, [c_long]),
    "qt_layernorm_train_backward_bf16": (c_int, [_P, _P, _P, _P, _P, _P, c_long, c_long, POINTER(QtChainStage), c_int, _FMT, _P, c_int, _P, c_size_t,
                                                 _P, _P, _P, POINTER(QtFaninItem), c_int, _P]),
    "qt_softmax_fq_probs_bf16": (c_int, [_P, _P, _P, _P, c_long, c_int, c_int, c_long, c_long, c_long, c_long, c_float, _FMT, _P, _P, _P, _P]),
    "qt_softmax_backward_chain_bf16": (c_int, [_P, _P, _P, c_long, c_long, c_float, POINTER(QtChainStage), c_int, _FMT, _P, _P]),
    "qt_grad_fanin_bf16": (c_int, [_P, POINTER(QtFaninItem), c_int, _P, ctypes.c_size_t, _FMT, _P, _P]),
    "qt_embedding_backward_bf16": (c_int, [_P, _P, c_long, c_long, c_long, c_long, _P, _P, _P]),
    "qt_attention_train_supported": (c_int, [c_long, c_int, c_int, c_int]),
    "qt_attention_train_bf16": (c_int, [_P, _P, _P, c_long, c_long, c_long, _P, c_long, c_long, c_long, POINTER(QtChainStage), _P, _P, _P, c_float,
                                        c_long, c_int, c_int, c_int, c_float, _FMT, _P, _P]),
    "qt_attention_train_backward_bf16": (c_int, [_P, _P, _P, _P, c_long, c_long, c_long, _P, _P, POINTER(QtChainStage), _P, _P, _P, _P,
                                                 POINTER(QtChainStage), POINTER(ctypes.c_void_p), c_float, _P, ctypes.c_size_t, _P, c_float, c_long,
                                                 c_int, c_int, c_int, c_float, _FMT, _P, _P]),
    "qt_attention_train_backward_ws_bytes": (ctypes.c_size_t, [c_int]),
    "qt_build_rowparams": (c_int, [_P, POINTER(QtRowParams)]),
    "qt_rowparams_apply_host": (c_uint16, [POINTER(QtRowParams), c_uint16, POINTER(c_int)]),
    "qt_linear_fqt_bf16": (c_int, [_P, _P, _P, _P, c_int, _P, c_int, c_uint32, _P, _P, c_int, c_int, _P]),
    "qt_fake_quant_multi_bf16": (c_int, [_P, c_int, ctypes.c_ulonglong, _FMT, _P, _P]),
    "qt_fake_quant_multi_bf16_fp8": (c_int, [_P, c_int, ctypes.c_ulonglong, _FMT, _P]),
    "qt_value_t_rows": (c_int, [_P, _P, c_long, c_long, c_long, c_int, c_long, c_long, c_long, _FMT, _P, _P]),
    "qt_attention_rows_bf16": (c_int, [_P, _P, _P, _P, c_long, c_long, c_long, _P, c_long, c_long, c_long, _P, _P, c_int, _FMT, _P, c_long, c_int, c_int,
                                       c_int, c_int, c_float, _P]),
    "qt_colsum_bf16": (c_int, [_P, _P, c_long, c_long, _P]),
    "qt_linear_fqt_plan": (c_int, [c_int, c_long, c_int, POINTER(c_int), POINTER(c_size_t), POINTER(c_size_t)]),
    "qt_linear_fqt_ws_bf16": (c_int, [_P, _P, _P, _P, c_int, _P, c_int, c_uint32, _P, _P, c_int, c_int, _P, c_size_t, _P, c_size_t, _P]),
    "qt_mlp_fq8_bf16": (c_int, [_P, c_int, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, c_int, c_int, _P]),
    "qt_bmm_fq_bf16": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_long, c_long, c_long, c_long, c_long,
                               _OPQ, _OPQ, _P]),
    "qt_softmax_fq_bf16": (c_int, [_P, _P, _P, c_long, c_int, c_int, c_long, c_long, c_long, c_long, c_float, _FMT, _P, _P,
                                  _P, _P]),
    "qt_softmax_fq_bf16_fp8": (c_int, [_P, _P, _P, _P, c_long, c_int, c_int, c_long, c_long, c_long, c_long, c_float, _FMT, _P]),
    "qt_mask_row_live": (c_int, [_P, c_long, c_long, c_long, _P, _P]),
    "qt_mask_row_live_checked": (c_int, [_P, c_long, c_long, c_long, _P, _P, _P]),
    "qt_value_codes_t": (c_int, [_P, _P, c_long, c_long, c_long, c_int, c_long, c_long, c_long, _FMT, _P]),
    "qt_attention_fp8": (c_int, [_P, _P, _P, c_int, _P, c_long, c_long, c_long, _P, c_long, c_long, c_long, c_int, _P, _P, _P, _P, c_long, c_int,
                                 c_int, c_int, c_int, c_float, _P]),
    "qt_softmax_fq_bf16_fp8_live": (c_int, [_P, _P, _P, c_long, c_int, c_int, c_long, c_long, c_long, c_long, c_float, _FMT, _P, c_long, c_long,
                                            c_long, _P]),
    "qt_attention_fq_live_bf16": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_long, c_long, c_long, c_float, _FMT, _P, _P, _P,
                                         c_int, _P, c_long, c_long, c_long, _P, _P]),
    "qt_attention_fq_out_bf16": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_long, c_long, c_long, c_float, _FMT, _P, _P]),
    "qt_attention_fq_bf16": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_long, c_long, c_long, c_float,
                                    _FMT, _P, _P, _P, _P]),
    "qt_quantize_mx_bf16": (c_int, [_P, _P, _P, _P, _P, c_size_t, c_size_t, c_int, _FMT, _P, c_float, c_int, _P, c_int, _P]),
    "qt_quantize_mx_f32": (c_int, [_P, _P, _P, _P, _P, c_size_t, c_size_t, c_int, _FMT, _P, c_float, c_int, _P, c_int, _P]),
    "qt_causal_lm_loss_bf16": (c_int, [_P, _P, c_long, c_long, c_long, c_long, ctypes.c_longlong, _P, _P, _P]),
    "qt_rmsnorm_bf16": (c_int, [_P, _P, _P, c_long, c_long, c_float, _P]),
    "qt_rmsnorm_fq8_bf16": (c_int, [_P, _P, _P, _P, c_long, c_long, c_float, _FMT, _P]),
    "qt_rmsnorm_consumers_bf16": (c_int, [_P, _P, _P, _P, _P, c_long, c_long, c_float, c_int, _P, _P, _P]),
    "qt_add_rmsnorm_bf16": (c_int, [_P, _P, _P, _P, _P, _P, c_long, c_long, c_float, _FMT, _P]),
    "qt_layernorm_bf16": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_long, c_long, c_float, _FMT, _P]),
    "qt_layernorm_consumers_bf16": (c_int, [_P, _P, _P, _P, _P, _P, c_long, c_long, c_float, c_int, _P, _P, _P]),
    "qt_gelu_bf16": (c_int, [_P, _P, _P, c_size_t, _FMT, _P]),
    "qt_silu_mul_bf16": (c_int, [_P, _P, _P, c_size_t, c_size_t, c_size_t, c_size_t, _P]),
    "qt_fake_quant_bf16_fp8_multi": (c_int, [_P, _P, c_int, _P, _FMT, _P]),
    "qt_fake_quant_rows_bf16_fp8": (c_int, [_P, _P, _P, c_long, c_long, c_long, c_long, c_long, c_long, c_long, _FMT, _P]),
    "qt_silu_mul_fq8_bf16": (c_int, [_P, _P, _P, _P, c_size_t, c_size_t, c_size_t, c_size_t, _FMT, _P]),
    "qt_rope_bf16": (c_int, [_P, _P, _P, _P, _P, _P, c_long, c_long, c_long, c_long, c_long, c_long, c_long, _P]),
    "qt_rope_fq_bf16": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_long, c_long, c_long, c_long, c_long, c_long, c_long, _FMT,
                               _FMT, _P]),
    "qt_rmsnorm_map_bf16": (c_int, [_P, _P, _P, _P, _P, c_long, c_long, c_float, _FMT, _P, c_int, _P]),
    "qt_silu_mul_map_bf16": (c_int, [_P, _P, _P, c_size_t, c_size_t, c_size_t, c_size_t, _FMT, _P, _P]),
    "qt_rope_map_bf16": (c_int, [_P, _P, _P, _P, _P, _P, c_long, c_long, c_long, c_long, c_long, c_long, c_long, _FMT, _P, c_int, c_int, _P]),
    "qt_rope_map_value": (c_int, [_P, _P, _P, _P, _P, _P, c_long, c_long, c_long, c_long, c_long, c_long, c_long, _FMT, _P, c_int, c_int, _P, _P,
                                  c_long, c_long, c_long, _P]),
    "qt_rope_map_value_weight": (c_int, [_P, _P, _P, _P, _P, _P, c_long, c_long, c_long, c_long, c_long, c_long, c_long, _FMT, _P, c_int, c_int, _P,
                                         _P, c_long, c_long, c_long, _P, _P, c_size_t, _P]),
    "qt_add_rmsnorm_sumfq_bf16": (c_int, [_P, _P, _P, _P, _P, _P, c_long, c_long, c_float, _FMT, _FMT, _P]),
    "qt_rope_fq_inner_value": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_long, c_long, c_long, c_long, c_long, c_long, c_long, _FMT,
                                      _FMT, _FMT, _FMT, _P, _P, c_long, c_long, c_long, _FMT, _P]),
    "qt_rope_fq_value": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_long, c_long, c_long, c_long, c_long, c_long, c_long, _FMT,
                                _FMT, _P, _P, c_long, c_long, c_long, _FMT, _P]),
    "qt_fp8_gemm": (c_int, [_P, c_int, _P, c_int, c_int, _P, _P, c_long, c_int, c_int, c_int, c_long, c_long, c_long, _P,
                           c_size_t, c_int, _P]),
    "qt_fp8_gemm_library_version": (c_int, []),
    "qt_fp8_gemm_tune": (c_int, [_P, c_int, _P, c_int, c_int, _P, _P, c_long, c_int, c_int, c_int, c_long, c_long, c_long, _P,
                           c_size_t, POINTER(c_int), _P, c_int, _P]),
    "qt_mx_pack": (c_int, [_P, _P, c_int, _P, _P, c_long, c_long, c_long, c_long, c_long, c_long, c_long, c_long, c_long, c_int,
                          c_int, _P, _P]),
    "qt_mx_gemm": (c_int, [_P, _P, c_int, _P, _P, c_int, _P, c_int, _P, c_long, c_int, c_int, c_int, c_long, c_long, _P]),
    "qt_q8_gemm": (c_int, [_P, _P, _P, c_int, _P, _P, c_int, c_int, c_long, c_int, c_int, c_int, c_long, c_long, _P]),
    "qt_bench_fake_quant_bf16": (c_int, [_P, _P, c_size_t, _FMT, _P, _P, _P, c_int, c_size_t, c_int, _P,
                                        POINTER(c_float)]),
    "qt_bench_fake_quant_bf16_fp8": (c_int, [_P, _P, _P, c_size_t, _FMT, _P, _P, c_int, c_size_t, c_int, _P,
                                            POINTER(c_float)]),
}

_lib = None
_PENDING = {"device": None, "multi": None}


def note_device(index):
    """Called by the launch helpers (fake_quantize._stream_ptr) with the device ordinal of the tensors about to be handed to a kernel:
    every native call from now until the next note_device runs with that device current (the C ABI takes raw pointers and a stream
    and launches on the calling thread's current device; hipBLASLt handles are per device as well).  A launch sequence that takes
    its stream once and then issues several kernels (attention: mask scan, value codes, core) therefore stays on the tensors' device.
    Free when the process sees one GPU."""
    _PENDING["device"] = index


class _GuardedLib:
    """The ctypes library with every entry point wrapped in a device guard (see note_device)."""

    def __init__(self, cdll):
        self._cdll = cdll
        for name in SIGNATURES:
            setattr(self, name, self._guard(getattr(cdll, name)))

    @staticmethod
    def _guard(fn):
        def call(*args):
            idx = _PENDING["device"]
            if idx is None:
                return fn(*args)
            import torch
            if _PENDING["multi"] is None:
                _PENDING["multi"] = torch.cuda.device_count() > 1
            if not _PENDING["multi"] or idx == torch.cuda.current_device():
                return fn(*args)
            with torch.cuda.device(idx):
                return fn(*args)
        call.__name__ = getattr(fn, "__name__", "native")
        return call


def lib():
    """Loads libqt_hip.so once.  Raises QtError when it is missing -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise QtError(
                f"{LIB_PATH} not found: the HIP extension is not built. "
                "Run `make -C quantized-training_amd` (or `python __graft_entry__.py`).")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.qt_abi_version() != ABI_VERSION:
            raise QtError(f"{LIB_PATH}: ABI version {L.qt_abi_version()}, this package binds {ABI_VERSION} (include/qt_hip.h QT_ABI_VERSION): "
                          "rebuild with `make -C quantized-training_amd`")
        _lib = _GuardedLib(L)
    return _lib


def check(code, what=""):
    if code != 0:
        msg = lib().qt_status_string(code).decode()
        if code == QT_ERR_BAD_DTYPE:
            raise ValueError(f"Unsupported dtype: {what}")
        raise QtError(f"libqt_hip {what}: {msg} (code {code})")


def build_map_u16(dtype):
    """uint16[65536] numpy array of bf16 bit patterns (host)."""
    import numpy as np
    out = np.empty(QT_MAP_ENTRIES, dtype=np.uint16)
    name = None if dtype is None else str(dtype).encode()
    code = lib().qt_build_map(name, out.ctypes.data)
    if code == QT_ERR_BAD_DTYPE:
        raise ValueError(f"Unsupported dtype: {dtype}")
    check(code, "qt_build_map")
    return out


def build_rowparams(map_u16):
    """QtRowParams of a uint16[65536] value map (host)."""
    import numpy as np
    m = np.ascontiguousarray(map_u16, dtype=np.uint16)
    rp = QtRowParams()
    check(lib().qt_build_rowparams(m.ctypes.data, ctypes.byref(rp)), "qt_build_rowparams")
    return rp


def format_for(dtype):
    f = QtFormat()
    name = None if dtype is None else str(dtype).encode()
    code = lib().qt_format_for(name, ctypes.byref(f))
    if code == QT_ERR_BAD_DTYPE:
        raise ValueError(f"Unsupported dtype: {dtype}")
    check(code, "qt_format_for")
    return f

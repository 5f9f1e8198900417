"""Fused fake-quant GEMM entry points used by the QAT Linear (and, later, the attention matmuls).

``fused_linear_or_none`` returns ``None`` whenever the fused HIP GEMM does not apply, and the
caller then runs the unfused sequence (HIP elementwise fake-quant + library GEMM); both are
device paths -- there is no CPU fallback for device tensors.
"""
import ctypes
import os

import torch

from . import _native
from .fake_quantize import STATS, FusedAmaxObsFakeQuantize, _stream_ptr
from .quantizer.quantizer import QScheme

_IDENTITY = _native.QtFormat(_native.QT_FMT_IDENTITY, 0, 0, 0.0, 0.0)


def fused_gemm_enabled():
    return os.environ.get("QT_FUSED_GEMM", "0") == "1"


def _operand(fq, device):
    """qt_operand_q for a weight fake-quantizer (per-tensor only)."""
    op = _native.QtOperandQ()
    op.fmt = fq._qt_format
    op.lut_dev = fq.qmap.data_ptr() if fq._qt_format.kind == _native.QT_FMT_LUT else None
    op.scale_f32_dev = fq.scale.data_ptr()
    op.amax_bits_dev = fq.amax_history.data_ptr() if fq._observe else None
    return op


def fused_linear_or_none(layer, x):
    """y = x @ fq(W)^T + b with W fake-quantized while its tiles are staged (qt_linear_fq_bf16).
    Applies to bf16 device tensors under no_grad with a per-tensor weight fake-quantizer."""
    fq = layer.weight_fake_quant
    W = layer.weight
    if not (fused_gemm_enabled() and isinstance(fq, FusedAmaxObsFakeQuantize)):
        return None
    if torch.is_grad_enabled() and (W.requires_grad or x.requires_grad):
        return None
    if not (x.device.type == "cuda" and x.dtype == torch.bfloat16 and W.dtype == torch.bfloat16):
        return None
    if fq.is_per_channel or fq.qscheme in (QScheme.MICROSCALING, QScheme.GROUP_WISE_AFFINE):
        return None
    if fq.outlier_threshold is not None or fq.record_histogram or not fq._quantize:
        return None
    if fq._qt_format.kind == _native.QT_FMT_LUT:
        return None       # table formats: elementwise pass (LDS-staged table) + library GEMM is faster
    K = W.shape[1]
    if K % 8 != 0 or not W.is_contiguous() or (layer.bias is not None and layer.bias.dtype != torch.bfloat16):
        return None
    L = _native.lib()
    fq._move_to(x.device)
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    M, N = x2.shape[0], W.shape[0]
    y = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    st = _stream_ptr(x)
    if fq._observe:
        if fq.amax_history.numel() == 0:
            fq.amax_history.resize_((fq.amax_history_len,)).fill_(0.0)
            fq.scale.resize_(()).fill_(1.0)
        _native.check(L.qt_scale_update(fq.amax_history.data_ptr(), int(fq.amax_history.shape[0]), 1,
                                        fq.scale.data_ptr(), float(fq.quant_max),
                                        int(bool(fq.force_scale_power_of_two)), st), "qt_scale_update")
    qw = _operand(fq, x.device)
    qx = _native.QtOperandQ()
    qx.fmt = _IDENTITY
    code = L.qt_linear_fq_bf16(x2.data_ptr(), W.data_ptr(),
                               layer.bias.data_ptr() if layer.bias is not None else None,
                               y.data_ptr(), M, N, K, ctypes.byref(qx), ctypes.byref(qw), st)
    _native.check(code, "qt_linear_fq_bf16")
    STATS.add(W.numel())
    return y.reshape(*x.shape[:-1], N)

"""The ``quantized_ops`` operator library (upstream src/quantized_training/decomposed.py:16-262).

Schemas are the reference's, verbatim, so converted PT2E graphs that call
``torch.ops.quantized_ops.quantize.default`` / ``dequantize.default`` keep working.  Device
tensors dispatch to the HIP kernels of libqt_hip.so (no fallback: a missing library raises);
CPU tensors use the same formulas in torch ops (host-side plumbing).  The block-scaled GEMMs
(linear_mx / matmul_mx, upstream :304-363) run on gfx950's scaled MFMA when the element format and
the scales are ones the instruction takes natively (see mx_gemm.py); conv / pooling pass-throughs
are outside this engine's hot path.
"""
import ctypes
from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from torch.library import Library

from . import _native
from .fake_quantize import _cpu_vmap, _stream_ptr, hip_vmap

__all__ = ["vmap", "quantize", "dequantize", "expand", "calculate_mx_qparam", "quantize_mx", "linear_mx", "matmul_mx",
           "conv2d_mx", "quantized_decomposed_lib"]

quantized_decomposed_lib = Library("quantized_ops", "DEF")
_lib = quantized_decomposed_lib

_lib.define("vmap(Tensor self, Tensor other) -> Tensor")
_lib.define(
    "quantize(Tensor input, Tensor scale, Tensor? zero_point=None, SymInt[]? axes=None, "
    "int? block_size=None, Tensor? qmap=None, Tensor? output_code=None) -> Tensor")
_lib.define(
    "dequantize(Tensor input, Tensor scale, Tensor? zero_point=None, SymInt[]? axes=None, "
    "int? block_size=None, Tensor? input_qmap=None, Tensor? output_qmap=None) -> Tensor")
_lib.define("linear(Tensor input, Tensor weight, Tensor? bias=None) -> Tensor")
_lib.define("matmul(Tensor self, Tensor other) -> Tensor")


def expand(input, shape, block_size):
    """Broadcast a block-wise parameter to ``shape`` (upstream decomposed.py:127-140)."""
    while input.ndim < len(shape):
        input = input.unsqueeze(0)
    for dim in range(len(shape)):
        if input.shape[dim] != shape[dim]:
            input = torch.repeat_interleave(input, block_size, dim)
    if list(input.shape) != list(shape):
        input = input[tuple(slice(0, n) for n in shape)]
    return input


# ---- vmap --------------------------------------------------------------------------------------
def _vmap_cpu(input, qmap):
    return _cpu_vmap(input, qmap).contiguous()


def _vmap_hip(input, qmap):
    if qmap.dtype != torch.bfloat16 or qmap.numel() != 65536:
        # index tables (NormalFloat codes are int64, fake_quantize.py:90-93): same gather in torch device ops
        return _cpu_vmap(input, qmap).contiguous()
    return hip_vmap(input, qmap.contiguous(), None)


_lib.impl("vmap", _vmap_cpu, "CPU")
_lib.impl("vmap", _vmap_hip, "CUDA")
_lib.impl("vmap", lambda input, qmap: torch.empty_like(input, memory_format=torch.contiguous_format), "Meta")


def vmap(input: torch.Tensor, qmap: torch.Tensor, chunk_size=65536) -> torch.Tensor:
    """``out = qmap[index(input)]`` (upstream decomposed.py:146-163); ``chunk_size`` is accepted and unused."""
    return torch.ops.quantized_ops.vmap(input, qmap)


# ---- quantize / dequantize ------------------------------------------------------------------------
def _single(t):
    return t is not None and t.numel() == 1


def _scalar_like(t, ref):
    return t.to(device=ref.device, dtype=ref.dtype).reshape(1).contiguous()


def _lut_format():
    return _native.QtFormat(_native.QT_FMT_LUT, 0, 0, 0.0, 0.0)


def _quantize_impl(input, scale, zero_point=None, axes=None, block_size=None, qmap=None, output_code=None):
    assert qmap is not None, "qmap must be provided for quantization"
    # the fused kernels compute in the tensor's own dtype, which is what torch's type promotion does
    # only when scale / zero_point already have that dtype
    fast = (input.device.type == "cuda" and block_size is None and _single(scale)
            and scale.dtype == input.dtype
            and (zero_point is None or (_single(zero_point) and zero_point.dtype == input.dtype))
            and input.dtype in (torch.bfloat16, torch.float32))
    if fast:
        L = _native.lib()
        x = input.contiguous()
        y = torch.empty_like(x)
        if x.numel():
            s = _scalar_like(scale, x)
            z = _scalar_like(zero_point, x) if zero_point is not None else None
            fn = L.qt_quantize_bf16 if x.dtype == torch.bfloat16 else L.qt_quantize_f32
            fmt = _lut_format()
            _native.check(fn(x.data_ptr(), y.data_ptr(), x.numel(), ctypes.byref(fmt), qmap.data_ptr(),
                             s.data_ptr(), z.data_ptr() if z is not None else None, _stream_ptr(x)), "qt_quantize")
        return y
    pow2 = getattr(scale, "_qt_pow2", False)
    if block_size is not None:
        scale = expand(scale, input.shape, block_size)
        if zero_point is not None:
            zero_point = expand(zero_point, input.shape, block_size)
    input = input / scale if zero_point is None else input / scale + zero_point
    out = torch.ops.quantized_ops.vmap(input, qmap)
    if pow2 and zero_point is None and out.device.type == "cuda":
        from .mx_gemm import remember_format
        remember_format(out, qmap)                     # lets linear_mx / matmul_mx take the element codes directly
    return out


def _dequantize_impl(input, scale, zero_point=None, axes=None, block_size=None, input_qmap=None, output_qmap=None):
    # the fused kernels compute in the tensor's own dtype, which is what torch's type promotion does
    # only when scale / zero_point already have that dtype
    fast = (input.device.type == "cuda" and block_size is None and _single(scale)
            and scale.dtype == input.dtype
            and (zero_point is None or (_single(zero_point) and zero_point.dtype == input.dtype))
            and input.dtype in (torch.bfloat16, torch.float32))
    if fast:
        L = _native.lib()
        x = input.contiguous()
        y = torch.empty_like(x)
        if x.numel():
            s = _scalar_like(scale, x)
            z = _scalar_like(zero_point, x) if zero_point is not None else None
            fn = L.qt_dequantize_bf16 if x.dtype == torch.bfloat16 else L.qt_dequantize_f32
            _native.check(fn(x.data_ptr(), y.data_ptr(), x.numel(), s.data_ptr(),
                             z.data_ptr() if z is not None else None,
                             input_qmap.data_ptr() if input_qmap is not None else None,
                             output_qmap.data_ptr() if output_qmap is not None else None, _stream_ptr(x)),
                          "qt_dequantize")
        return y
    if input_qmap is not None:
        input = torch.ops.quantized_ops.vmap(input, input_qmap)
    if block_size is not None:
        scale = expand(scale, input.shape, block_size)
        if zero_point is not None:
            zero_point = expand(zero_point, input.shape, block_size)
    out = input * scale if zero_point is None else (input - zero_point) * scale
    if output_qmap is not None:
        out = torch.ops.quantized_ops.vmap(out, output_qmap)
    return out


for _key in ("CPU", "CUDA"):
    _lib.impl("quantize", _quantize_impl, _key)
    _lib.impl("dequantize", _dequantize_impl, _key)
_lib.impl("quantize", lambda input, *a, **k: torch.empty_like(input, memory_format=torch.contiguous_format), "Meta")
_lib.impl("dequantize", lambda input, *a, **k: torch.empty_like(input), "Meta")


def quantize(input, scale, zero_point=None, axes=None, block_size=None, qmap=None, output_code=None):
    """``vmap(input / scale [+ zero_point], qmap)`` (upstream decomposed.py:173-210)."""
    return torch.ops.quantized_ops.quantize(input, scale, zero_point, axes, block_size, qmap, output_code)


def dequantize(input, scale, zero_point=None, axes=None, block_size=None, input_qmap=None, output_qmap=None):
    """``(vmap?(input) [- zero_point]) * scale`` then optional output map (upstream decomposed.py:220-262)."""
    return torch.ops.quantized_ops.dequantize(input, scale, zero_point, axes, block_size, input_qmap, output_qmap)


# ---- block-scaled (microscaling) parameters (upstream decomposed.py:365-448) ----------------------------
_lib.define("calculate_mx_qparam(Tensor self, SymInt[] axes, int block_size, float quant_max, "
            "bool force_scale_power_of_two=False, Tensor? scale_qmap=None) -> Tensor")
_lib.define("quantize_mx(Tensor self, Tensor qmap, SymInt[] axes, int block_size, float quant_max, "
            "bool force_scale_power_of_two=False, Tensor? scale_qmap=None, Tensor? output_code=None) -> (Tensor, Tensor)")


def _const_like(value, ref):
    # a 0-dim tensor on ref's device: `tensor / python_scalar` multiplies by a rounded reciprocal on the
    # GPU, `tensor / tensor` is the IEEE division the reference's CPU path performs
    return torch.tensor(value, dtype=ref.dtype, device=ref.device)


def _calculate_mx_qparam_impl(input, axes, block_size, quant_max, force_scale_power_of_two=False, scale_qmap=None):
    from .mx_utils import _reshape_to_blocks, _shared_exponents
    import math
    assert block_size > 0
    axes = [axes] if type(axes) == int else list(axes)
    axes = [x + input.ndim if x < 0 else x for x in axes]
    blocks, axes, _, _ = _reshape_to_blocks(input, axes, block_size)
    block_axes = [x + 1 for x in axes]
    if force_scale_power_of_two:
        shared_exp = _shared_exponents(blocks, method="max", axes=block_axes, ebits=0)
        shared_exp = shared_exp - math.floor(math.log2(quant_max))
        for ax in reversed(axes):
            shared_exp = torch.squeeze(shared_exp, dim=ax + 1)
        scale = 2 ** shared_exp
    else:
        amax = torch.amax(torch.abs(blocks), dim=block_axes)
        scale = amax / _const_like(quant_max, amax)
        if scale_qmap is not None:
            scale = torch.ops.quantized_ops.vmap(scale, scale_qmap)
    scale = torch.where(scale > 0.0, scale, _const_like(1.0, scale))
    if force_scale_power_of_two:
        scale._qt_pow2 = True                          # an E8M0-representable scale, by construction
    return scale


def _quantize_mx_hip_or_none(input, qmap, axes, block_size, quant_max, pow2, scale_qmap):
    """One fused device pass (qt_quantize_mx_*) for blocks along the last axis: values, scales and, for a format the
    matrix instruction takes with power-of-two scales, the packed operand that linear_mx / matmul_mx consume."""
    if input.device.type != "cuda" or input.dtype not in (torch.bfloat16, torch.float32) or input.dim() < 1:
        return None
    axes = [axes] if isinstance(axes, int) else list(axes)
    if len(axes) != 1 or axes[0] % input.dim() != input.dim() - 1:
        return None
    per = 8 if input.dtype == torch.bfloat16 else 4
    cols = input.shape[-1]
    if (not isinstance(block_size, int) or block_size < per or block_size & (block_size - 1) or block_size > 64 * per
            or cols == 0 or cols % block_size or input.numel() == 0 or not quant_max > 0):
        return None
    if qmap.dtype != torch.bfloat16 or qmap.numel() != 65536 or \
            (scale_qmap is not None and (scale_qmap.dtype != torch.bfloat16 or scale_qmap.numel() != 65536)):
        return None
    from . import mx_gemm
    L = _native.lib()
    x = input.contiguous()
    rows = x.numel() // cols
    q = torch.empty_like(x)
    sf = torch.empty(x.shape[:-1] + (cols // block_size,), dtype=x.dtype, device=x.device)
    fmt_name = mx_gemm.format_of_qmap(qmap) if pow2 and block_size % 32 == 0 else None
    codes = e8 = None
    fid = -1
    if fmt_name is not None and (cols * mx_gemm._BITS[mx_gemm.FMT_ID[fmt_name]] // 8) % 16 == 0:
        fid = mx_gemm.FMT_ID[fmt_name]
        codes = torch.empty((rows, cols * mx_gemm._BITS[fid] // 8), dtype=torch.uint8, device=x.device)
        e8 = torch.empty((rows, cols // 32), dtype=torch.uint8, device=x.device)
    from .fake_quantize import _launch_format
    fmt = _launch_format(_lut_format(), qmap)                # the row form behind the map, when this is a map from _device_map
    fn = L.qt_quantize_mx_bf16 if x.dtype == torch.bfloat16 else L.qt_quantize_mx_f32
    _native.check(fn(x.data_ptr(), q.data_ptr(), sf.data_ptr(), codes.data_ptr() if codes is not None else None,
                     e8.data_ptr() if e8 is not None else None, rows, cols, block_size, ctypes.byref(fmt), qmap.data_ptr(),
                     float(quant_max), int(bool(pow2)), scale_qmap.data_ptr() if scale_qmap is not None else None, fid,
                     _stream_ptr(x)), "qt_quantize_mx")
    if pow2:
        sf._qt_pow2 = True
        if fmt_name is not None:
            q._qt_mx_fmt = fmt_name
        if codes is not None:
            q._qt_mx_packed_act = (fid, block_size, codes, e8)
    return sf, q


def _quantize_mx_impl(input, qmap, axes, block_size, quant_max, force_scale_power_of_two=False, scale_qmap=None,
                      output_code=None):
    fused = _quantize_mx_hip_or_none(input, qmap, axes, block_size, quant_max, force_scale_power_of_two, scale_qmap)
    if fused is not None:
        return fused
    scale = torch.ops.quantized_ops.calculate_mx_qparam(input, axes, block_size, quant_max, force_scale_power_of_two,
                                                        scale_qmap)
    q = torch.ops.quantized_ops.quantize(input, scale, None, axes, block_size, qmap)
    return scale, q


_lib.impl("calculate_mx_qparam", _calculate_mx_qparam_impl, "CompositeExplicitAutograd")
_lib.impl("quantize_mx", _quantize_mx_impl, "CompositeExplicitAutograd")


def calculate_mx_qparam(input, axes, block_size, quant_max, force_scale_power_of_two=False, scale_qmap=None):
    axes = [axes] if isinstance(axes, int) else list(axes)
    return torch.ops.quantized_ops.calculate_mx_qparam(input, axes, block_size, quant_max, force_scale_power_of_two, scale_qmap)


def quantize_mx(input, qmap, axes, block_size, quant_max, force_scale_power_of_two=False, scale_qmap=None, output_code=None):
    axes = [axes] if isinstance(axes, int) else list(axes)
    return torch.ops.quantized_ops.quantize_mx(input, qmap, axes, block_size, quant_max, force_scale_power_of_two,
                                               scale_qmap, output_code)


# ---- GEMM pass-throughs (operands arrive already fake-quantized; upstream decomposed.py:77-94) -----
_lib.impl("linear", lambda input, weight, bias=None: F.linear(input, weight, bias), "CompositeExplicitAutograd")
_lib.impl("matmul", lambda self, other: torch.matmul(self, other), "CompositeExplicitAutograd")


# ---- block-scaled GEMMs (upstream decomposed.py:265-363) ----------------------------------------------------
_MX_KW = ("*, Tensor? input_scale=None, Tensor? weight_scale=None, int? block_size=None, "
          "Tensor? input_code=None, Tensor? weight_code=None) -> Tensor")
_lib.define("conv2d_mx(Tensor input, Tensor weight, Tensor? bias=None, SymInt[2] stride=1, SymInt[2] padding=0, "
            "SymInt[2] dilation=1, SymInt groups=1, " + _MX_KW)
_lib.define("linear_mx(Tensor input, Tensor weight, Tensor? bias=None, " + _MX_KW)
_lib.define("matmul_mx(Tensor self, Tensor other, " + _MX_KW)


def _mx_operand(x, scale, code, block_size):
    """Codebook lookup, then the block scales broadcast over their blocks (what every *_mx op does to an operand)."""
    if code is not None:
        x = code[x.to(torch.long)].to(x.dtype)
    if scale is not None:
        x = x * expand(scale, x.shape, block_size)
    return x


def _conv2d_mx_impl(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, *, input_scale=None,
                    weight_scale=None, block_size=None, input_code=None, weight_code=None):
    return F.conv2d(_mx_operand(input, input_scale, input_code, block_size),
                    _mx_operand(weight, weight_scale, weight_code, block_size), bias, stride, padding, dilation, groups)


def _linear_mx_impl(input, weight, bias=None, *, input_scale=None, weight_scale=None, block_size=None,
                    input_code=None, weight_code=None):
    if input.device.type == "cuda":
        from .mx_gemm import mx_linear_or_none
        out = mx_linear_or_none(input, weight, bias, input_scale, weight_scale, block_size, input_code, weight_code)
        if out is not None:
            return out
    return F.linear(_mx_operand(input, input_scale, input_code, block_size),
                    _mx_operand(weight, weight_scale, weight_code, block_size), bias)


def _matmul_mx_impl(self, other, *, input_scale=None, weight_scale=None, block_size=None, input_code=None,
                    weight_code=None):
    if self.device.type == "cuda":
        from .mx_gemm import mx_matmul_or_none
        out = mx_matmul_or_none(self, other, input_scale, weight_scale, block_size, input_code, weight_code)
        if out is not None:
            return out
    return torch.matmul(_mx_operand(self, input_scale, input_code, block_size),
                        _mx_operand(other, weight_scale, weight_code, block_size))


_lib.impl("conv2d_mx", _conv2d_mx_impl, "CompositeExplicitAutograd")
_lib.impl("linear_mx", _linear_mx_impl, "CompositeExplicitAutograd")
_lib.impl("matmul_mx", _matmul_mx_impl, "CompositeExplicitAutograd")


# ---- outlier side path of a block-scaled linear layer (upstream decomposed.py:450-566) ---------------------------
# filter_outlier splits an activation into its inliers (|x| <= threshold, outliers zeroed) and the outliers as a CSR
# matrix of the [rows, C] view, padded to int(numel * max_pct) entries; spmm_csr multiplies that sparse matrix by the
# (dequantized) weight.  Upstream walks both with Python loops over .item(); here they are index arithmetic on the
# tensor's own device, entries in the same row-major order and, on CPU, accumulated in the same order.
_lib.define("filter_outlier(Tensor input, float threshold, float max_pct=0.05) -> (Tensor, Tensor, Tensor, Tensor)")
_lib.define("spmm_csr(Tensor data, Tensor indices, Tensor indptr, Tensor B, Tensor? B_scale=None, "
            "Tensor? B_code=None, int? block_size=None, bool weight_transposed=False) -> Tensor")


def _filter_outlier_impl(input, threshold, max_pct=0.05):
    is_outlier = torch.abs(input) > threshold
    inlier = torch.where(is_outlier, 0, input)
    outliers = torch.where(is_outlier, input, 0).reshape(-1, input.shape[-1])
    max_nnz = int(input.numel() * max_pct)
    nz = outliers != 0
    rows, cols = nz.nonzero(as_tuple=True)                      # row-major, as the reference's nested loops visit them
    counts = nz.sum(dim=1)
    indptr = torch.zeros(outliers.shape[0] + 1, dtype=torch.int32, device=input.device)
    indptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    nnz = int(rows.numel())
    if nnz > max_nnz:
        import logging
        logging.getLogger(__name__).warning(f"Number of non-zero elements {nnz} exceeds max_nnz {max_nnz}, truncating.")
    keep = min(nnz, max_nnz)
    data = torch.zeros((max_nnz,), dtype=input.dtype, device=input.device)
    indices = torch.zeros((max_nnz,), dtype=torch.int32, device=input.device)
    data[:keep] = outliers[rows[:keep], cols[:keep]]
    indices[:keep] = cols[:keep].to(torch.int32)
    return inlier, data, indices, indptr


def _spmm_csr_impl(data, indices, indptr, B, B_scale=None, B_code=None, block_size=None, weight_transposed=False):
    M = indptr.numel() - 1
    K = B.shape[1] if weight_transposed else B.shape[0]
    if B_code is not None:
        B = B_code[B.to(torch.long)]
    if B_scale is not None:
        B = B * expand(B_scale, B.shape, block_size)
    if weight_transposed:
        B = B.T
    Y = torch.zeros((M, K), dtype=data.dtype, device=data.device)
    nnz = int(indptr[-1])
    if nnz > data.numel():                                        # the truncated case: upstream's loop runs off the arrays
        raise IndexError(f"index {data.numel()} is out of bounds for dimension 0 with size {data.numel()}")
    if nnz == 0:
        return Y
    counts = (indptr[1:] - indptr[:-1]).to(torch.long)
    row_of = torch.repeat_interleave(torch.arange(M, device=data.device), counts)
    cols = indices[:nnz].to(torch.long)
    Y.index_add_(0, row_of, (data[:nnz].unsqueeze(0) * B[:, cols]).T.contiguous())
    return Y


_lib.impl("filter_outlier", _filter_outlier_impl, "CompositeExplicitAutograd")
_lib.impl("spmm_csr", _spmm_csr_impl, "CompositeExplicitAutograd")


def linear_mx(input, weight, bias=None, **kw):
    return torch.ops.quantized_ops.linear_mx(input, weight, bias, **kw)


def matmul_mx(self, other, **kw):
    return torch.ops.quantized_ops.matmul_mx(self, other, **kw)


def conv2d_mx(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, **kw):
    return torch.ops.quantized_ops.conv2d_mx(input, weight, bias, stride, padding, dilation, groups, **kw)

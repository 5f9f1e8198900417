"""Native execution of converted per-tensor PT2E graphs (upstream quantize_pt2e.py:323-446).

`convert_pt2e` produces, exactly like upstream,

    xq = quantized_ops.quantize(x, s_x, qmap)        integer / FP8 VALUES held in the model dtype
    y  = aten.linear(xq, W_codes, bias_codes)        (or aten.matmul of two quantized operands)
    out = quantized_ops.dequantize(y, s_x * s_w, out_qmap)

and upstream runs that GEMM in bf16 / fp32 arithmetic on the value tensors.  `fuse_native_gemms` rewrites each such triple
of a DEVICE model into one `quantized_ops.linear_q` / `matmul_q` node that runs the product on the integer / FP8 matrix
cores:

  * int8 x int8: both operands narrowed to int8 codes (weights once, at fusion time), `qt_q8_gemm` -- v_mfma_i32_16x16x64_i8,
    exact int32 accumulation -- with bias, the rounding of the GEMM output to the model dtype, the dequantize multiply and its
    rounding in the epilogue (the rounding points of the three-node sequence are kept);
  * fp8_e4m3 / fp8_e5m2 (per-tensor scaled): operands narrowed to OCP FP8 codes, the in-tree scaled-MFMA GEMM at unit block
    scales (`qt_mx_gemm`; the library GEMM `qt_fp8_gemm` only where that kernel does not take the shape), then the dequantize kernel.

The graph `convert_pt2e` returns is untouched on the CPU (it is pinned node for node to upstream's); fusion is a separate,
idempotent pass that convert_pt2e applies to device models (`native=None` -> automatic, `QT_PT2E_NATIVE=0` turns it off).
Accumulation is exact (int32) where upstream's fp32 / bf16 `aten.linear` rounds, so results agree within the accumulation
bound, not bit for bit.  STATS counts the native launches (tests assert the route was taken).
"""
import ctypes
import os

import torch
from torch.fx import GraphModule

from . import _native
from .fake_quantize import _stream_ptr

__all__ = ["fuse_native_gemms", "STATS"]

STATS = {"linear_int8": 0, "matmul_int8": 0, "linear_fp8": 0, "matmul_fp8": 0}
_IDENTITY_DTYPES = (None, "None", "bfloat16", "float32", "float16")
_FP8 = {"fp8_e4m3": torch.float8_e4m3fn, "fp8_e5m2": torch.float8_e5m2}

_lib = torch.library.Library("quantized_ops", "FRAGMENT")
# out_map: the dequantize node's input map (identity for the cases that are fused); kept as an operand so that the CPU / fallback
# formulation below stays the three-node sequence verbatim
_lib.define("linear_q(Tensor input, Tensor scale, Tensor qmap, Tensor weight_codes, Tensor? bias, Tensor out_scale, "
            "Tensor? out_map, str kind) -> Tensor")
_lib.define("matmul_q(Tensor self, Tensor self_scale, Tensor self_qmap, Tensor other, Tensor other_scale, Tensor other_qmap, "
            "bool other_is_kn, Tensor out_scale, Tensor? out_map, str kind) -> Tensor")


def _codes(x, scale, qmap, kind):
    """quantize(x, scale, qmap) narrowed to the format's code type (exact: the values ARE int8 / FP8 values)."""
    v = torch.ops.quantized_ops.quantize(x, scale, None, None, None, qmap)
    return v.to(torch.int8) if kind == "int8" else v.to(_FP8[kind])


def _scale_arg(out_scale, n, dtype):
    s = out_scale.to(dtype).reshape(-1).contiguous()
    if s.numel() not in (1, n):
        raise ValueError(f"dequantize scale of {s.numel()} elements behind a GEMM with {n} output columns")
    return s, int(s.numel() == n and n > 1)


def _q8(a, b, bias, out_scale, out_map, out_shape, batch, M, N, K, a_bs, b_bs, dtype):
    y = torch.empty(out_shape, dtype=dtype, device=a.device)
    fold = int(out_map is not None and dtype == torch.float32)
    s, per_col = _scale_arg(out_scale, N, dtype)
    bias_t = bias.to(dtype).contiguous() if bias is not None else None
    _native.check(_native.lib().qt_q8_gemm(a.data_ptr(), b.data_ptr(), y.data_ptr(), int(dtype == torch.float32),
                                           bias_t.data_ptr() if bias_t is not None else None, s.data_ptr(), per_col, fold, batch, M, N, K,
                                           a_bs, b_bs, _stream_ptr(a)), "qt_q8_gemm")
    return y


_UNIT_E8M0 = {}          # device -> uint8 tensor of 127s (scale 2^0): the block scales of a per-tensor FP8 operand
_UNIT_E8M0_OLD = []      # outgrown buffers stay allocated: a hipGraph captured earlier still reads their addresses
_MX_FMT = {torch.float8_e4m3fn: 0, torch.float8_e5m2: 1}


def _fp8_codes_gemm(a, w, bias):
    """C = a . w^T (+ bias) on FP8 CODES through the in-tree scaled-MFMA GEMM (qt_mx_gemm, v_mfma_scale_f32_16x16x128_f8f6f4) at unit
    block scales: per-tensor FP8 operands are a block-scaled operand whose every scale is 2^0.  None when the kernel does not take
    the problem (the caller then keeps the library GEMM)."""
    M, K = a.shape
    N = w.shape[0]
    if K % 32 or a.dtype not in _MX_FMT or w.dtype not in _MX_FMT or not a.is_contiguous() or not w.is_contiguous() or a.data_ptr() % 16 or w.data_ptr() % 16:
        return None
    need = max(M, N) * (K // 32)
    ones = _UNIT_E8M0.get(a.device)
    if ones is None or ones.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            return None
        if ones is not None:
            _UNIT_E8M0_OLD.append(ones)
        ones = torch.full((need,), 127, dtype=torch.uint8, device=a.device)
        _UNIT_E8M0[a.device] = ones
    y = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    rc = _native.lib().qt_mx_gemm(a.data_ptr(), ones.data_ptr(), _MX_FMT[a.dtype], w.data_ptr(), ones.data_ptr(), _MX_FMT[w.dtype], y.data_ptr(), 0,
                                  bias.data_ptr() if bias is not None else None, 1, M, N, K, 0, 0, _stream_ptr(a))
    if rc != 0:
        return None
    return y


def _linear_q(input, scale, qmap, weight_codes, bias, out_scale, out_map, kind):
    if input.device.type != "cuda" or input.dtype not in (torch.bfloat16, torch.float32):
        w = weight_codes.to(input.dtype)
        y = torch.nn.functional.linear(torch.ops.quantized_ops.quantize(input, scale, None, None, None, qmap), w, bias)
        return torch.ops.quantized_ops.dequantize(y, out_scale, None, None, None, out_map)
    K, N = input.shape[-1], weight_codes.shape[0]
    a = _codes(input, scale, qmap, kind).reshape(-1, K)
    M = a.shape[0]
    if kind == "int8" and K % 16 == 0 and a.data_ptr() % 16 == 0 and weight_codes.data_ptr() % 16 == 0:
        STATS["linear_int8"] += 1
        return _q8(a, weight_codes, bias, out_scale, out_map, (*input.shape[:-1], N), 1, M, N, K, 0, 0, input.dtype)
    if kind in _FP8 and input.dtype == torch.bfloat16:
        b16 = bias.to(torch.bfloat16).contiguous() if bias is not None else None
        y = _fp8_codes_gemm(a.contiguous(), weight_codes, b16)
        if y is not None:
            STATS["linear_fp8_native"] = STATS.get("linear_fp8_native", 0) + 1
        else:
            from .fused import lt_fp8_gemm
            y = lt_fp8_gemm(a, weight_codes, b16)
        if y is not None:
            STATS["linear_fp8"] += 1
            y = y.reshape(*input.shape[:-1], N)
            return torch.ops.quantized_ops.dequantize(y, out_scale, None, None, None, out_map)
    y = torch.nn.functional.linear(a.to(input.dtype).reshape(input.shape), weight_codes.to(input.dtype), bias)
    return torch.ops.quantized_ops.dequantize(y, out_scale, None, None, None, out_map)


def _matmul_q(self, self_scale, self_qmap, other, other_scale, other_qmap, other_is_kn, out_scale, out_map, kind):
    """self [..., M, K]; other: [..., N, K] (other_is_kn False: the graph multiplied by its transpose) or [..., K, N]."""
    def fallback():
        a = torch.ops.quantized_ops.quantize(self, self_scale, None, None, None, self_qmap)
        b = torch.ops.quantized_ops.quantize(other, other_scale, None, None, None, other_qmap)
        y = torch.matmul(a, b if other_is_kn else b.transpose(-1, -2))
        return torch.ops.quantized_ops.dequantize(y, out_scale, None, None, None, out_map)
    if self.device.type != "cuda" or self.dtype not in (torch.bfloat16, torch.float32) or kind != "int8":
        return fallback()
    if self.dim() < 2 or other.dim() != self.dim() or self.shape[:-2] != other.shape[:-2]:
        return fallback()
    M, K = self.shape[-2], self.shape[-1]
    N = other.shape[-1] if other_is_kn else other.shape[-2]
    if K % 16 != 0 or (other.shape[-2] if other_is_kn else other.shape[-1]) != K:
        return fallback()
    batch = 1
    for d in self.shape[:-2]:
        batch *= d
    a = _codes(self, self_scale, self_qmap, kind).contiguous()
    b = _codes(other, other_scale, other_qmap, kind)
    b = (b.transpose(-1, -2) if other_is_kn else b).contiguous()            # [..., N, K]
    if a.data_ptr() % 16 or b.data_ptr() % 16 or (batch > 1 and (M * K) % 16) or (batch > 1 and (N * K) % 16) or batch > 65535:
        return fallback()
    STATS["matmul_int8"] += 1
    return _q8(a, b, None, out_scale, out_map, (*self.shape[:-2], M, N), batch, M, N, K, M, N, self.dtype)


_lib.impl("linear_q", _linear_q, "CompositeExplicitAutograd")
_lib.impl("matmul_q", _matmul_q, "CompositeExplicitAutograd")


def _is(node, target):
    return node is not None and getattr(node, "op", None) == "call_function" and node.target == target


def _per_tensor_quantize(node):
    """`quantize(x, scale, None, None, None, qmap)` with a per-tensor scale -> (x, scale node, qmap node, dtype) or None."""
    Q = torch.ops.quantized_ops.quantize.default
    if not _is(node, Q) or len(node.args) < 6 or node.kwargs:
        return None
    x, s, zp, axes, bs, qmap = node.args[:6]
    if zp is not None or axes is not None or bs is not None or qmap is None or len(node.args) > 6 and node.args[6] is not None:
        return None
    return x, s, qmap, str(node.meta.get("dtype"))


def _sole_dequantize(model, node, dtype):
    """The `dequantize(gemm, scale, None, None, None, input_qmap)` node behind a GEMM (quantize_pt2e.py:409-413: the GEMM
    output is first rounded to the accelerator's output format through `input_qmap`, then multiplied by s_x * s_w).  Fusable
    when that map is the identity for the model dtype (output_dtype None, or bfloat16 in a bf16 model).
    Returns (dequantize node, scale node) or None."""
    DQ = torch.ops.quantized_ops.dequantize.default
    users = list(node.users.keys())
    if len(users) != 1 or not _is(users[0], DQ):
        return None
    dq = users[0]
    a = list(dq.args) + [None] * (7 - len(dq.args))
    if a[0] is not node or a[2] is not None or a[3] is not None or a[4] is not None or a[6] is not None or dq.kwargs:
        return None
    if a[5] is not None:
        try:
            tag = getattr(model.get_buffer(a[5].target), "_qt_dtype", "?")
        except AttributeError:
            return None
        if not (tag in (None, "None", "float32") or (tag == "bfloat16" and dtype == torch.bfloat16)):
            return None
    return dq, a[1], a[5]


def fuse_native_gemms(model: GraphModule) -> int:
    """Rewrites quantize -> aten.linear / aten.matmul -> dequantize triples whose operands are int8 (or FP8) codes into
    linear_q / matmul_q nodes.  Returns the number of fused GEMMs; leaves everything else as it is."""
    graph = model.graph
    fused = 0
    for node in list(graph.nodes):
        if _is(node, torch.ops.aten.linear.default):
            q = _per_tensor_quantize(node.args[0])
            w = node.args[1]
            if q is None or w.op != "get_attr":
                continue
            kind = q[3].split(",")[0]
            if kind not in ("int8", "fp8_e4m3", "fp8_e5m2") or str(w.meta.get("dtype", "")).split(",")[0] != kind:
                continue
            param = model.get_parameter(w.target) if w.target in dict(model.named_parameters()) else model.get_buffer(w.target)
            tail = _sole_dequantize(model, node, param.dtype)
            if tail is None:
                continue
            dq, s_out, m_out = tail
            codes = param.detach().to(torch.int8) if kind == "int8" else param.detach().to(_FP8[kind])
            if not torch.equal(codes.to(param.dtype), param.detach()):
                continue                                     # not exactly representable: keep the value-tensor GEMM
            name = str(w.target).replace(".", "_") + "_codes"
            i = 0
            while hasattr(model, name):
                i += 1
                name = f"{str(w.target).replace('.', '_')}_codes_{i}"
            model.register_buffer(name, codes.contiguous(), persistent=False)
            bias = node.args[2] if len(node.args) > 2 else None
            with graph.inserting_before(dq):
                c_node = graph.create_node("get_attr", name)
                new = graph.call_function(torch.ops.quantized_ops.linear_q.default,
                                          (q[0], q[1], q[2], c_node, bias, s_out, m_out, kind))
            new.meta = dict(dq.meta)
            dq.replace_all_uses_with(new)
            graph.erase_node(dq)
            graph.erase_node(node)
            fused += 1
        elif _is(node, torch.ops.aten.matmul.default):
            qa, qb = _per_tensor_quantize(node.args[0]), _per_tensor_quantize(node.args[1])
            if qa is None or qb is None:
                continue
            kind = qa[3].split(",")[0]
            if kind != "int8" or qb[3].split(",")[0] != kind:
                continue
            first = next(iter(model.parameters()), None)
            if first is None:
                first = next(iter(model.buffers()), None)
            if first is None:
                continue                                     # a graph without parameters or buffers: nothing says what dtype it runs in
            tail = _sole_dequantize(model, node, first.dtype)
            if tail is None:
                continue
            dq, s_out, m_out = tail
            other, kn = qb[0], True
            if _is(other, torch.ops.aten.transpose.int) and sorted(int(a) for a in other.args[1:3]) == [-2, -1]:
                other, kn = other.args[0], False             # quantize(x^T) = quantize(x)^T: take the [N, K] operand as stored
            with graph.inserting_before(dq):
                new = graph.call_function(torch.ops.quantized_ops.matmul_q.default,
                                          (qa[0], qa[1], qa[2], other, qb[1], qb[2], kn, s_out, m_out, kind))
            new.meta = dict(dq.meta)
            dq.replace_all_uses_with(new)
            graph.erase_node(dq)
            graph.erase_node(node)
            fused += 1
    if fused:
        graph.eliminate_dead_code(is_impure_node=lambda n: n.op in {"placeholder", "output"})
        graph.lint()
        model.recompile()
    return fused


def enabled_for(model) -> bool:
    if os.environ.get("QT_PT2E_NATIVE", "1") == "0":
        return False
    try:
        return next(iter(model.parameters())).device.type == "cuda"
    except StopIteration:
        return False

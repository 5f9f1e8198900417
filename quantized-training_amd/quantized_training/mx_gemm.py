"""linear_mx / matmul_mx on gfx950's block-scaled matrix instruction (upstream decomposed.py:304-363).

The reference states these ops as "multiply each operand by its expanded block scales, then F.linear /
torch.matmul".  When the element values are one of the formats v_mfma_scale_f32_16x16x128_f8f6f4 takes
(fp8_e4m3, fp8_e5m2, fp6_e2m3, fp6_e3m2, fp4_e2m1) and the block scales are powers of two over blocks of a
multiple of 32 elements, the same product is computed from the element codes and E8M0 scale bytes directly
(qt_mx_pack + qt_mx_gemm in libqt_hip.so): no dequantized copy of either operand is ever written.

The ops only see tensors, so what an operand holds is recovered from
  * the value map it was quantized with (torch.ops.quantized_ops.quantize / quantize_mx remember the format
    on their result; a map is recognised by identity with this package's tables or, once per buffer, by content);
  * the packer itself, which refuses (flag) values outside the format and scales that are not 2^e; weights are
    packed once and cached on the tensor, so that check costs one host sync per weight, not per call.
Anything else (codebooks, int formats, block size 16, fp8 scales ...) returns None and the caller runs the
reference formulation.  QT_MX_GEMM=0 disables the native path.
"""
import ctypes
import os

import torch

from . import _native
from .fake_quantize import _stream_ptr, _table_for

__all__ = ["mx_linear_or_none", "mx_matmul_or_none", "remember_format", "STATS"]

FMT_ID = {"fp8_e4m3": 0, "fp8_e5m2": 1, "fp6_e2m3": 2, "fp6_e3m2": 3, "fp4_e2m1": 4}
_BITS = {0: 8, 1: 8, 2: 6, 3: 6, 4: 4}
_PAIRS = {(0, 0), (0, 1), (1, 0), (1, 1), (2, 2), (3, 3), (4, 4), (0, 4), (2, 4), (3, 4)}     # kernels in qt_mx_gemm.hip
_NARROW_FIRST = ("fp4_e2m1", "fp6_e2m3", "fp6_e3m2", "fp8_e4m3", "fp8_e5m2")


class _Stats:
    """How many GEMMs took the native path (tests assert on it; a silent fallback would hide a regression)."""
    def __init__(self):
        self.native = 0
        self.fallback = 0

    def reset(self):
        self.native = self.fallback = 0


STATS = _Stats()
_DEBUG = False            # set mx_gemm._DEBUG = True to print why a problem took the reference formulation


def _fallback(why):
    STATS.fallback += 1
    if _DEBUG:
        print("mx_gemm: reference formulation:", why, flush=True)
    return None
def format_of_qmap(qmap):
    """Name of the native element format a value map rounds to, or None.  The answer is kept on the tensor object
    (module buffers are long-lived objects; a cache keyed by address would go stale when memory is re-used)."""
    name = getattr(qmap, "_qt_dtype", None)
    if name is None or getattr(qmap, "_qt_dtype_version", qmap._version) != qmap._version:
        name = ""
        if qmap.numel() == 65536 and qmap.dtype == torch.bfloat16:
            bits = qmap.view(torch.int16)
            for cand in FMT_ID:
                if torch.equal(bits, _table_for(cand, qmap.device).view(torch.int16)):       # host sync, once per buffer
                    name = cand
                    break
        qmap._qt_dtype = name
        qmap._qt_dtype_version = qmap._version
    return name if name in FMT_ID else None


def remember_format(values, qmap, scale=None, pow2=None):
    """Called by quantize / quantize_mx on their results: note which format the values are in."""
    fmt = format_of_qmap(qmap) if qmap is not None else None
    if fmt is not None:
        values._qt_mx_fmt = fmt
    if scale is not None and pow2:
        scale._qt_pow2 = True
    return values


def _enabled():
    return os.environ.get("QT_MX_GEMM", "1") != "0"


def _pack(values, scale, fmt_id, block_size, batch, rows, K, x_strides, s_strides, check):
    """-> (codes [batch, rows, K*bits/8] u8, e8m0 [batch, rows, K/32] u8) or None when the packer refuses."""
    L = _native.lib()
    dev = values.device
    codes = torch.empty((batch, rows, K * _BITS[fmt_id] // 8), dtype=torch.uint8, device=dev)
    e8 = torch.empty((batch, rows, K // 32), dtype=torch.uint8, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev) if check else None
    _native.check(L.qt_mx_pack(values.data_ptr(), scale.data_ptr(), int(values.dtype == torch.float32), codes.data_ptr(),
                               e8.data_ptr(), batch, rows, K, *x_strides, *s_strides, block_size, fmt_id,
                               bad.data_ptr() if check else None, _stream_ptr(values)), "qt_mx_pack")
    if check and int(bad.item()) != 0:
        return None
    return codes, e8


def _weight_operand(weight, scale, block_size):
    """Constant operand [N, K] (+ scale [N, K / bs]): packed once, cached on the tensor object."""
    key = (weight._version, scale.data_ptr(), scale._version, block_size)
    hit = getattr(weight, "_qt_mx_packed", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    N, K = weight.shape
    w, s = weight.contiguous(), scale.contiguous()
    known = getattr(weight, "_qt_mx_fmt", None)
    result = None
    for name in ((known,) if known in FMT_ID else _NARROW_FIRST):
        fid = FMT_ID[name]
        if (K * _BITS[fid] // 8) % 16:
            continue
        packed = _pack(w, s, fid, block_size, 1, N, K, (0, K, 1), (0, s.shape[-1], 1), check=True)
        if packed is not None:
            result = (fid, packed[0], packed[1])
            break
    weight._qt_mx_packed = (key, result)
    return result


def _common_ok(x, scale, code, block_size):
    return (code is None and scale is not None and isinstance(block_size, int) and block_size >= 32 and block_size % 32 == 0
            and x.dtype in (torch.bfloat16, torch.float32) and scale.dtype == x.dtype and x.device.type == "cuda")


def mx_linear_or_none(input, weight, bias, input_scale, weight_scale, block_size, input_code, weight_code):
    if not _enabled() or not _common_ok(input, input_scale, input_code, block_size) \
            or not _common_ok(weight, weight_scale, weight_code, block_size) or weight.dtype != input.dtype:
        return None
    fa = getattr(input, "_qt_mx_fmt", None)
    if fa not in FMT_ID or not getattr(input_scale, "_qt_pow2", False) or weight.dim() != 2 or input.dim() < 2:
        return _fallback(f"input format {fa}, power-of-two scale {getattr(input_scale, '_qt_pow2', False)}")
    K = input.shape[-1]
    N = weight.shape[0]
    if weight.shape[1] != K or K % 32 or K % block_size or input.numel() == 0:
        return _fallback(f"K = {K}, block {block_size}")
    wop = _weight_operand(weight, weight_scale, block_size)
    fid = FMT_ID[fa]
    if wop is None or (fid, wop[0]) not in _PAIRS or (K * _BITS[fid] // 8) % 16:
        return _fallback(f"weight operand {None if wop is None else wop[0]} with input format {fid}, K = {K}")
    x = input.contiguous()
    M = x.numel() // K
    pre = getattr(input, "_qt_mx_packed_act", None)          # quantize_mx already emitted the packed operand
    if pre is not None and pre[0] == fid and pre[1] == block_size and pre[2].shape == (M, K * _BITS[fid] // 8):
        a_codes, a_e8 = pre[2], pre[3]
    else:
        s = input_scale.contiguous()
        a_codes, a_e8 = _pack(x, s, fid, block_size, 1, M, K, (0, K, 1), (0, s.shape[-1], 1), check=False)
    out = torch.empty(input.shape[:-1] + (N,), dtype=input.dtype, device=input.device)
    b = bias.to(input.dtype).contiguous() if bias is not None else None
    L = _native.lib()
    _native.check(L.qt_mx_gemm(a_codes.data_ptr(), a_e8.data_ptr(), fid, wop[1].data_ptr(), wop[2].data_ptr(), wop[0],
                               out.data_ptr(), int(out.dtype == torch.float32), b.data_ptr() if b is not None else None,
                               1, M, N, K, 0, 0, _stream_ptr(x)), "qt_mx_gemm")
    STATS.native += 1
    return out


def mx_matmul_or_none(a, b, a_scale, b_scale, block_size, a_code, b_code):
    """a [..., M, K] with blocks along K (last axis); b [..., K, N] with blocks along K (axis -2)."""
    if not _enabled() or not _common_ok(a, a_scale, a_code, block_size) or not _common_ok(b, b_scale, b_code, block_size) \
            or a.dtype != b.dtype:
        return None
    fa, fb = getattr(a, "_qt_mx_fmt", None), getattr(b, "_qt_mx_fmt", None)
    ok = (fa in FMT_ID and fb in FMT_ID and getattr(a_scale, "_qt_pow2", False) and getattr(b_scale, "_qt_pow2", False)
          and a.dim() >= 2 and b.dim() >= 2 and a.shape[:-2] == b.shape[:-2] and a.shape[-1] == b.shape[-2])
    if ok:
        M, K = a.shape[-2:]
        N = b.shape[-1]
        fia, fib = FMT_ID[fa], FMT_ID[fb]
        ok = ((fia, fib) in _PAIRS and K % 32 == 0 and K % block_size == 0 and (K * _BITS[fia] // 8) % 16 == 0
              and (K * _BITS[fib] // 8) % 16 == 0 and a.numel() > 0 and b.numel() > 0
              and a_scale.shape == a.shape[:-1] + (K // block_size,) and b_scale.shape == b.shape[:-2] + (K // block_size, N))
    if not ok:
        return _fallback(f"matmul operands {fa} x {fb}, shapes {tuple(a.shape)} {tuple(b.shape)}")
    batch = 1
    for d in a.shape[:-2]:
        batch *= d
    if batch > 65535:
        return _fallback("batch > 65535")
    a3, sa3 = a.contiguous().view(batch, M, K), a_scale.contiguous().view(batch, M, K // block_size)
    b3, sb3 = b.contiguous().view(batch, K, N), b_scale.contiguous().view(batch, K // block_size, N)
    pre = getattr(a, "_qt_mx_packed_act", None)
    if pre is not None and pre[0] == fia and pre[1] == block_size and pre[2].shape == (batch * M, K * _BITS[fia] // 8):
        pa = (pre[2], pre[3])
    else:
        pa = _pack(a3, sa3, fia, block_size, batch, M, K, (M * K, K, 1), (sa3.stride(0), sa3.stride(1), 1), check=False)
    # second operand: logical rows are b's columns (stride 1), k walks b's rows (stride N)
    pb = _pack(b3, sb3, fib, block_size, batch, N, K, (K * N, 1, N), (sb3.stride(0), 1, sb3.stride(1)), check=False)
    out = torch.empty(a.shape[:-1] + (N,), dtype=a.dtype, device=a.device)
    L = _native.lib()
    _native.check(L.qt_mx_gemm(pa[0].data_ptr(), pa[1].data_ptr(), fia, pb[0].data_ptr(), pb[1].data_ptr(), fib,
                               out.data_ptr(), int(out.dtype == torch.float32), None, batch, M, N, K, M, N,
                               _stream_ptr(a)), "qt_mx_gemm")
    STATS.native += 1
    return out

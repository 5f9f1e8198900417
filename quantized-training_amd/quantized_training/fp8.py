"""``quantize_to_fp8_e4m3`` / ``quantize_to_fp8_e5m2`` on tensors (upstream src/quantized_training/fp8.py:10-67):
round-to-nearest-even to 3 / 2 mantissa bits with subnormals, saturate to +-448 / +-57344, flush
|x| <= min_subnormal/2 to +0, non-finite -> NaN.  Evaluated by the C ABI (qt_round_fp8_*)."""
import torch

from . import _native
from .fake_quantize import _stream_ptr

__all__ = ["quantize_to_fp8_e4m3", "quantize_to_fp8_e5m2"]


def _round_fp8(input, mbits, fp8_max, fp8_min):
    L = _native.lib()
    x = input.detach().to(torch.float32).contiguous()
    y = torch.empty_like(x)
    if x.numel():
        if x.device.type == "cuda":
            _native.check(L.qt_round_fp8_f32(x.data_ptr(), y.data_ptr(), x.numel(), mbits, fp8_max, fp8_min,
                                             _stream_ptr(x)), "qt_round_fp8_f32")
        else:
            _native.check(L.qt_round_fp8_host(x.data_ptr(), y.data_ptr(), x.numel(), mbits, fp8_max, fp8_min),
                          "qt_round_fp8_host")
    return y.to(input.dtype)


def quantize_to_fp8_e4m3(input: torch.Tensor, mbits: int = 3, fp8_max: float = 448, fp8_min: float = 2 ** -6):
    return _round_fp8(input, mbits, float(fp8_max), float(fp8_min))


def quantize_to_fp8_e5m2(input: torch.Tensor, mbits: int = 2, fp8_max: float = 57344, fp8_min: float = 2 ** -14):
    return _round_fp8(input, mbits, float(fp8_max), float(fp8_min))

"""``quantize_to_posit`` on tensors (upstream src/quantized_training/posit.py:6-67): posit<nbits,es>
round-to-nearest-even with regime-dominated saturation to maxpos / minpos and flush below
2^floor(-(nbits-1)*2^es + 2^(es-1)).  Evaluated by the C ABI (qt_round_posit_*)."""
import torch

from . import _native
from .fake_quantize import _stream_ptr

__all__ = ["quantize_to_posit"]


def quantize_to_posit(input: torch.Tensor, nbits: int = 8, es: int = 1, round_to_even: bool = True,
                      return_pbits: bool = False):
    """Returns the rounded tensor, or (rounded, pbits int32) with `return_pbits` (posit.py:60-65; see include/qt_hip.h
    qt_posit_quantize_* for how the pattern is defined where upstream's int32 arithmetic is not)."""
    L = _native.lib()
    x = input.detach().to(torch.float32).contiguous()
    y = torch.empty_like(x)
    pbits = torch.empty(x.shape, dtype=torch.int32, device=x.device) if return_pbits else None
    plain = round_to_even and not return_pbits
    if x.numel():
        if x.device.type == "cuda":
            st = _stream_ptr(x)
            if plain:
                _native.check(L.qt_round_posit_f32(x.data_ptr(), y.data_ptr(), x.numel(), int(nbits), int(es), st), "qt_round_posit_f32")
            else:
                _native.check(L.qt_posit_quantize_f32(x.data_ptr(), y.data_ptr(), pbits.data_ptr() if return_pbits else None, x.numel(),
                                                      int(nbits), int(es), int(bool(round_to_even)), st), "qt_posit_quantize_f32")
        elif plain:
            _native.check(L.qt_round_posit_host(x.data_ptr(), y.data_ptr(), x.numel(), int(nbits), int(es)), "qt_round_posit_host")
        else:
            _native.check(L.qt_posit_quantize_host(x.data_ptr(), y.data_ptr(), pbits.data_ptr() if return_pbits else None, x.numel(),
                                                   int(nbits), int(es), int(bool(round_to_even))), "qt_posit_quantize_host")
    y = y.to(input.dtype)
    return (y, pbits) if return_pbits else y

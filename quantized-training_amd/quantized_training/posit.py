"""``quantize_to_posit`` on tensors (upstream src/quantized_training/posit.py:6-67): posit<nbits,es>
round-to-nearest-even with regime-dominated saturation to maxpos / minpos and flush below
2^floor(-(nbits-1)*2^es + 2^(es-1)).  Evaluated by the C ABI (qt_round_posit_*)."""
import torch

from . import _native
from .fake_quantize import _stream_ptr

__all__ = ["quantize_to_posit"]


def quantize_to_posit(input: torch.Tensor, nbits: int = 8, es: int = 1, round_to_even: bool = True,
                      return_pbits: bool = False) -> torch.Tensor:
    if not round_to_even or return_pbits:
        raise NotImplementedError("only round_to_even=True without posit bit patterns is supported")
    L = _native.lib()
    x = input.detach().to(torch.float32).contiguous()
    y = torch.empty_like(x)
    if x.numel():
        if x.device.type == "cuda":
            _native.check(L.qt_round_posit_f32(x.data_ptr(), y.data_ptr(), x.numel(), int(nbits), int(es),
                                               _stream_ptr(x)), "qt_round_posit_f32")
        else:
            _native.check(L.qt_round_posit_host(x.data_ptr(), y.data_ptr(), x.numel(), int(nbits), int(es)),
                          "qt_round_posit_host")
    return y.to(input.dtype)

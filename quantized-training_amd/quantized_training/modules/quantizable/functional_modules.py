"""Hookable wrappers for functional ops so that `quantize()` can attach fake-quantizers to their
inputs (upstream src/quantized_training/modules/quantizable/functional_modules.py:8-26)."""
import os
from typing import Union

import torch
from torch import Tensor

__all__ = ["AddFunctional", "MulFunctional", "MatmulFunctional"]


class AddFunctional(torch.nn.Module):
    """``x + y`` -- residual connections hang here."""

    def forward(self, x: Tensor, y: Union[Tensor, float]) -> Tensor:
        return torch.add(x, y)


class MulFunctional(torch.nn.Module):
    """``x * y`` -- attention-score scaling hangs here."""

    def forward(self, x: Tensor, y: Union[Tensor, float]) -> Tensor:
        return torch.mul(x, y)


class MatmulFunctional(torch.nn.Module):
    """``x @ y`` -- QK^T and attention-probabilities @ V hang here."""

    def forward(self, x: Tensor, y: Tensor) -> Tensor:
        # Q . K^T with both operands already fake-quantized to exact FP8 values by the kernel that produced them
        # (model_fusions.rope_fq): the same products on the FP8 matrix cores.  The input hooks have run by now.
        x8, k8 = getattr(x, "_qt_fp8", None), getattr(y, "_qt_fp8_of_transpose", None)
        if x8 is not None and k8 is not None and os.environ.get("QT_FP8_ATTENTION", "1") != "0" and x.dim() == 4 and x8.shape == x.shape and k8.shape[:2] == x.shape[:2] \
                and k8.shape[-1] == x.shape[-1] and not (torch.is_grad_enabled() and (x.requires_grad or y.requires_grad)):
            from ...fused import lt_fp8_gemm
            B, H, S, D = x.shape
            out = lt_fp8_gemm(x8.reshape(B * H, S, D), k8.reshape(B * H, k8.shape[2], D))
            if out is not None:
                return out.view(B, H, S, k8.shape[2])
        return torch.matmul(x, y)

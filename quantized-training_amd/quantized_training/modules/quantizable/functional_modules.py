"""Hookable wrappers for functional ops so that `quantize()` can attach fake-quantizers to their
inputs (upstream src/quantized_training/modules/quantizable/functional_modules.py:8-26)."""
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
        return torch.matmul(x, y)

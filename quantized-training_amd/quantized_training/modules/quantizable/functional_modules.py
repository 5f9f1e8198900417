"""Hookable wrappers for functional ops so that `quantize()` can attach fake-quantizers to their
inputs (upstream src/quantized_training/modules/quantizable/functional_modules.py:8-26)."""
import os

import torch

__all__ = ["AddFunctional", "MulFunctional", "MatmulFunctional"]


class _BinaryOp(torch.nn.Module):
    """A module whose forward is one two-operand torch function: the place where input hooks (fake-quantizers of
    either operand) and backward hooks are registered."""

    fn = None

    def forward(self, lhs, rhs):
        return type(self).fn(lhs, rhs)


class _LazyAdd(torch.autograd.Function):
    """lhs + rhs whose VALUES the next kernel writes (train_fusions.add_layernorm_or_none: the LayerNorm launch forms the sum and stores it
    into this result's memory): the autograd node and the module call -- with the backward hooks hanging on it -- exist as usual."""

    @staticmethod
    def forward(ctx, lhs, rhs):
        return torch.empty_like(lhs)

    @staticmethod
    def backward(ctx, g):
        return g, g


class AddFunctional(_BinaryOp):
    """residual connections hang here"""

    fn = staticmethod(torch.add)

    def forward(self, lhs, rhs):
        if self.__dict__.pop("_qt_lazy_add", False) and lhs.shape == rhs.shape and lhs.dtype == rhs.dtype:
            return _LazyAdd.apply(lhs, rhs)
        return torch.add(lhs, rhs)


class MulFunctional(_BinaryOp):
    """attention-score scaling hangs here"""

    fn = staticmethod(torch.mul)


class MatmulFunctional(_BinaryOp):
    """Q.K^T and probabilities @ V hang here"""

    fn = staticmethod(torch.matmul)

    def forward(self, lhs, rhs):
        # Q . K^T with both operands already fake-quantized to exact FP8 values by the kernel that produced them
        # (model_fusions.rope_fq): the same products on the FP8 matrix cores.  The input hooks have run by now.
        q8, k8 = getattr(lhs, "_qt_fp8", None), getattr(rhs, "_qt_fp8_of_transpose", None)
        fresh = getattr(lhs, "_qt_ver", None) == lhs._version and getattr(rhs, "_qt_ver", None) == rhs._version
        if (q8 is not None and k8 is not None and fresh and os.environ.get("QT_FP8_ATTENTION", "1") != "0" and lhs.dim() == 4
                and q8.shape == lhs.shape and k8.shape[:2] == lhs.shape[:2] and k8.shape[-1] == lhs.shape[-1]
                and not (torch.is_grad_enabled() and (lhs.requires_grad or rhs.requires_grad))):
            from ...fused import lt_fp8_gemm
            B, H, S, D = lhs.shape
            out = lt_fp8_gemm(q8.reshape(B * H, S, D), k8.reshape(B * H, k8.shape[2], D))
            if out is not None:
                return out.view(B, H, S, k8.shape[2])
        return torch.matmul(lhs, rhs)

"""Linear layer whose weight is fake-quantized on every forward
(upstream src/quantized_training/modules/qat/linear.py:15-81)."""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.parametrize import (
    is_parametrized,
    transfer_parametrizations_and_params,
    type_before_parametrizations,
)

__all__ = ["Linear"]


class _LinearColsumBias(torch.autograd.Function):
    """F.linear whose backward computes the bias gradient -- grad_output.sum(0), on the gradient the backward-pre hook already
    fake-quantized (quantize.py:116-179) -- with qt_colsum_bf16 (fp32 sums in a fixed order, one rounding) instead of torch's generic
    reduction: 12.6 -> ~5 us per Linear inside the replayed training step.  Input and weight gradients are the two GEMMs autograd's own
    linear backward runs (grad_output . W and grad_output^T . x)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        import ctypes
        from ... import _native
        x, w = ctx.saved_tensors
        gy2 = gy.reshape(-1, gy.shape[-1])
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gy2.mm(w).view(x.shape)
        if ctx.needs_input_grad[1]:
            gw = gy2.t().mm(x.reshape(-1, x.shape[-1]))
        if ctx.needs_input_grad[2]:
            from ... import train_fusions
            gb = train_fusions.take_colsum(gy)          # the launch that fake-quantized this gradient summed its columns on the way
            if gb is not None:
                return gx, gw, gb
            g = gy2 if gy2.is_contiguous() else gy2.contiguous()
            if g.dtype != torch.bfloat16 or g.data_ptr() % 16 or g.shape[1] % 8 or g.shape[0] == 0:
                gb = gy2.sum(0)                         # a view at an odd storage offset, another dtype: what F.linear's backward does
            else:
                gb = torch.empty(g.shape[1], dtype=g.dtype, device=g.device)
                _native.note_device(g.device.index)
                rc = _native.lib().qt_colsum_bf16(g.data_ptr(), gb.data_ptr(), g.shape[0], g.shape[1],
                                                  ctypes.c_void_p(torch.cuda.current_stream(g.device).cuda_stream))
                if rc in (_native.QT_ERR_BAD_ARG, _native.QT_ERR_UNALIGNED):
                    gb = gy2.sum(0)
                else:
                    _native.check(rc, "qt_colsum_bf16")
        return gx, gw, gb


class Linear(nn.Linear):
    """``F.linear(x, weight_fake_quant(W), b)``; shares ``weight`` / ``bias`` Parameters with the
    float module it was made from."""

    _FLOAT_MODULE = nn.Linear

    def __init__(self, in_features, out_features, bias=True, qconfig=None, device=None, dtype=None) -> None:
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__(in_features, out_features, bias, **factory_kwargs)
        assert qconfig, "qconfig must be provided for QAT module"
        self.qconfig = qconfig
        # the fake-quantizer's buffers are real even when the layer itself is built on `meta`
        fq_device = None if device is not None and str(device) == "meta" else device
        self.weight_fake_quant = qconfig.weight(factory_kwargs={"device": fq_device, "dtype": dtype})

    def forward(self, input):
        from ...fused import cached_weight, fused_linear_or_none
        out = fused_linear_or_none(self, input)
        if out is not None:
            return out
        # (opt-in) in eval with a frozen / stateless weight fake-quantizer the quantized weight is kept, see fused.py
        wq = cached_weight(self, "dense", lambda: self.weight_fake_quant(self.weight))
        b = self.bias
        if (b is not None and b.requires_grad and torch.is_grad_enabled() and input.is_cuda and input.dtype == torch.bfloat16
                and wq.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and self.out_features % 8 == 0):
            return _LinearColsumBias.apply(input, wq, b)          # training on the device: the bias gradient through qt_colsum_bf16
        return F.linear(input, wq, b)

    @classmethod
    def from_float(cls, mod):
        """The QAT twin of a float ``nn.Linear`` that carries a ``qconfig``; weight and bias stay the SAME Parameter
        objects (or parametrizations) as the float module's."""
        kind = type_before_parametrizations(mod)
        assert kind == cls._FLOAT_MODULE, f" qat.{cls.__name__}.from_float only works for {cls._FLOAT_MODULE.__name__}"
        assert getattr(mod, "qconfig", None), "Input float module must have a valid qconfig"
        twin = cls(mod.in_features, mod.out_features, bias=mod.bias is not None, qconfig=mod.qconfig,
                   device="meta")                      # parameters are adopted from `mod` right below
        for name in ("weight", "bias"):
            if is_parametrized(mod, name):
                transfer_parametrizations_and_params(mod, twin, name)
            else:
                setattr(twin, name, getattr(mod, name))
        return twin

    def to_float(self):
        """A plain ``nn.Linear`` holding detached copies of the current parameters."""
        plain = torch.nn.Linear(self.in_features, self.out_features, bias=self.bias is not None, device="meta")
        plain.weight = torch.nn.Parameter(self.weight.detach())
        plain.bias = None if self.bias is None else torch.nn.Parameter(self.bias.detach())
        return plain.train(self.training)

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
        return F.linear(input, cached_weight(self, "dense", lambda: self.weight_fake_quant(self.weight)), self.bias)

    @classmethod
    def from_float(cls, mod):
        assert type_before_parametrizations(mod) == cls._FLOAT_MODULE, (
            f" qat.{cls.__name__}.from_float only works for {cls._FLOAT_MODULE.__name__}")
        assert hasattr(mod, "qconfig"), "Input float module must have qconfig defined"
        assert mod.qconfig, "Input float module must have a valid qconfig"
        # build on the meta device: the parameters are replaced by the float module's right below
        qat = cls(mod.in_features, mod.out_features, bias=mod.bias is not None, qconfig=mod.qconfig,
                  device="meta")
        if is_parametrized(mod, "weight"):
            transfer_parametrizations_and_params(mod, qat, "weight")
        else:
            qat.weight = mod.weight
        if is_parametrized(mod, "bias"):
            transfer_parametrizations_and_params(mod, qat, "bias")
        else:
            qat.bias = mod.bias
        return qat

    def to_float(self):
        linear = torch.nn.Linear(self.in_features, self.out_features, self.bias is not None)
        linear.weight = torch.nn.Parameter(self.weight.detach())
        if self.bias is not None:
            linear.bias = torch.nn.Parameter(self.bias.detach())
        linear.train(self.training)
        return linear

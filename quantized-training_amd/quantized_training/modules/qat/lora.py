"""LoRA linear layer whose merged weight is fake-quantized on every forward
(upstream src/quantized_training/modules/qat/lora.py:12-97).

Upstream subclasses ``peft.tuners.lora.Linear``.  ``peft`` is an optional dependency here, so the twin wraps any
layer that has peft's LoRA-linear shape instead -- the frozen projection either on the layer itself (older peft,
where the LoRA layer IS an ``nn.Linear``) or under ``base_layer`` (current peft), plus ``lora_A`` / ``lora_B``
(ModuleDicts of ``nn.Linear``), ``scaling`` (dict), ``active_adapters``, ``merged``, ``disable_adapters`` and
``fan_in_fan_out`` -- and it is registered for ``peft.tuners.lora.Linear`` in the QAT mapping only when ``peft``
imports (quantization_mappings.py).  What the forward computes is upstream's, call for call:

    W' = fq_w( W + sum_adapters T( fq_w(B) @ fq_w(A) ) * scaling )          y = x @ T(W')^T + b

with ONE weight fake-quantizer used three times per adapter and forward (A, then B, then the merged weight -- the
order matters to a delayed-scaling observer), the base weight detached (``weight.data.clone()``, upstream :46) so
that only A and B receive gradients (straight-through), and LoRA dropout not applied (upstream :41-54 never
calls it).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.parametrize import type_before_parametrizations

__all__ = ["LoraLinear", "is_lora_linear"]


def _transpose(w, fan_in_fan_out):
    return w.T if fan_in_fan_out else w


def is_lora_linear(mod) -> bool:
    """True for a peft-style LoRA linear layer (see the module docstring for the attributes looked at)."""
    return all(hasattr(mod, a) for a in ("lora_A", "lora_B", "scaling")) and \
        (hasattr(mod, "base_layer") or isinstance(mod, nn.Linear))


class LoraLinear(nn.Module):
    _FLOAT_MODULE = None                        # peft.tuners.lora.Linear when peft is installed (set by the mapping)

    def __init__(self, float_layer, qconfig=None):
        super().__init__()
        assert qconfig, "quantizer must be provided for QAT module"
        assert is_lora_linear(float_layer), "LoraLinear wraps a peft-style LoRA linear layer"
        self.qconfig = qconfig
        base = getattr(float_layer, "base_layer", float_layer)
        # parameters are shared with the float layer, exactly as upstream's from_float re-points them (:84-91)
        self.weight = base.weight
        self.bias = base.bias
        self.in_features, self.out_features = float_layer.in_features, float_layer.out_features
        self.lora_A, self.lora_B = float_layer.lora_A, float_layer.lora_B
        self.scaling = float_layer.scaling
        self.r = getattr(float_layer, "r", None)
        self.lora_alpha = getattr(float_layer, "lora_alpha", None)
        self.lora_dropout = getattr(float_layer, "lora_dropout", None)
        self.fan_in_fan_out = bool(getattr(float_layer, "fan_in_fan_out", False))
        self.merged_adapters = list(getattr(float_layer, "merged_adapters", []))
        self._disable_adapters = bool(getattr(float_layer, "disable_adapters", False))
        adapters = getattr(float_layer, "active_adapters", None)
        if adapters is None:
            adapters = getattr(float_layer, "active_adapter", None)
        if adapters is None:
            adapters = list(self.lora_A.keys())
        self.active_adapters = [adapters] if isinstance(adapters, str) else list(adapters)
        self.weight_fake_quant = qconfig.weight()

    @property
    def merged(self) -> bool:
        return bool(self.merged_adapters)

    @property
    def disable_adapters(self) -> bool:
        return self._disable_adapters

    def enable_adapters(self, enabled: bool) -> None:
        self._disable_adapters = not enabled

    def _linear(self, x):
        return F.linear(x, _transpose(self.weight, self.fan_in_fan_out), self.bias)

    def merged_weight(self):
        """``fq_w(W + sum T(fq_w(B) @ fq_w(A)) * scaling)`` in the layer's own storage orientation."""
        w = self.weight.data.clone()
        for name in self.active_adapters:
            if name in self.lora_A.keys():
                a = self.weight_fake_quant(self.lora_A[name].weight)
                b = self.weight_fake_quant(self.lora_B[name].weight)
                w = w + _transpose(b @ a, self.fan_in_fan_out) * self.scaling[name]
        return self.weight_fake_quant(w)

    def forward(self, x):
        previous_dtype = x.dtype
        if self.disable_adapters or self.merged:
            # upstream un-merges here when adapters are disabled; a merged float layer is un-merged in from_float
            # already, so both branches reduce to the plain projection (:36-41)
            result = self._linear(x)
        else:
            result = F.linear(x, _transpose(self.merged_weight(), self.fan_in_fan_out), self.bias)
        return result.to(previous_dtype)

    @classmethod
    def from_float(cls, mod):
        if cls._FLOAT_MODULE is not None:
            assert type_before_parametrizations(mod) == cls._FLOAT_MODULE, (
                f" qat.{cls.__name__}.from_float only works for {cls._FLOAT_MODULE.__name__}")
        assert hasattr(mod, "qconfig"), "Input float module must have qconfig defined"
        assert mod.qconfig, "Input float module must have a valid qconfig"
        if getattr(mod, "merged", False):
            mod.unmerge()
        return cls(mod, qconfig=mod.qconfig)

    @classmethod
    def to_float(cls):
        raise NotImplementedError

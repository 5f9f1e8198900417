"""Which float modules are swapped for which quantizable / QAT twins, and which module classes
each ``--quantize_forward`` / ``--quantize_backprop`` op group covers
(upstream src/quantized_training/quantization_mappings.py:16-72).

HF model families are looked up lazily and only if `transformers` provides them; families whose
twins this engine does not build (DistilBERT, GPT-2, Whisper, conv QAT) are absent.  The LoRA QAT layer is
registered for ``peft.tuners.lora.Linear`` when ``peft`` imports (upstream :20).
"""
import importlib
from typing import Any, Callable, Dict

import torch.nn as nn

from .modules import qat as nnqat
from .modules import quantizable

__all__ = [
    "DEFAULT_QAT_MODULE_MAPPINGS", "TRANSFORMER_MODULE_MAPPINGS", "QCONFIG_PROPAGATE_MODULE_CLASS_LIST",
]

DEFAULT_QAT_MODULE_MAPPINGS: Dict[Callable, Any] = {
    nn.Linear: nnqat.Linear,
}


try:                                                     # optional dependency (upstream imports it unconditionally)
    from peft.tuners import lora as _peft_lora
    nnqat.LoraLinear._FLOAT_MODULE = _peft_lora.Linear
    DEFAULT_QAT_MODULE_MAPPINGS[_peft_lora.Linear] = nnqat.LoraLinear
except Exception:  # noqa: BLE001
    _peft_lora = None


def _hf(module, name):
    try:
        return getattr(importlib.import_module("transformers.models." + module), name)
    except Exception:  # noqa: BLE001  (family absent from this transformers build)
        return None


_HF_TWINS = [
    ("bert.modeling_bert", "BertSelfAttention", quantizable.BertSelfAttention),
    ("bert.modeling_bert", "BertSelfOutput", quantizable.BertSelfOutput),
    ("bert.modeling_bert", "BertOutput", quantizable.BertOutput),
    ("roberta.modeling_roberta", "RobertaSelfAttention", quantizable.BertSelfAttention),
    ("roberta.modeling_roberta", "RobertaSelfOutput", quantizable.BertSelfOutput),
    ("roberta.modeling_roberta", "RobertaOutput", quantizable.BertOutput),
    ("mobilebert.modeling_mobilebert", "MobileBertSelfAttention", quantizable.MobileBertSelfAttention),
    ("mobilebert.modeling_mobilebert", "MobileBertSelfOutput", quantizable.MobileBertSelfOutput),
    ("mobilebert.modeling_mobilebert", "FFNOutput", quantizable.FFNOutput),
    ("mobilebert.modeling_mobilebert", "MobileBertOutput", quantizable.MobileBertOutput),
    # upstream leaves LLaMA commented out (quantization_mappings.py:33); the in-place twin works with current HF
    ("llama.modeling_llama", "LlamaAttention", quantizable.LlamaAttention),
]

TRANSFORMER_MODULE_MAPPINGS: Dict[Callable, Any] = {}
for _mod, _name, _twin in _HF_TWINS:
    _cls = _hf(_mod, _name)
    if _cls is not None:
        TRANSFORMER_MODULE_MAPPINGS[_cls] = _twin


def _present(*classes):
    return [c for c in classes if c is not None]


def _activation_classes():
    try:
        from transformers.activations import GELUActivation
    except Exception:  # noqa: BLE001
        GELUActivation = None
    return _present(nn.ReLU, nn.GELU, nn.Softmax, GELUActivation)


def _conv1d():
    try:
        from transformers.pytorch_utils import Conv1D
        return Conv1D
    except Exception:  # noqa: BLE001
        return None


QCONFIG_PROPAGATE_MODULE_CLASS_LIST = {
    "activation": _activation_classes(),
    "gemm": _present(nn.Conv1d, nn.Conv2d, nn.Conv3d, nn.Linear, _conv1d(), quantizable.MatmulFunctional),
    "layernorm": _present(nn.LayerNorm, _hf("llama.modeling_llama", "LlamaRMSNorm"),
                          _hf("mobilebert.modeling_mobilebert", "NoNorm")),
    "residual": [quantizable.AddFunctional],
    "scaling": [quantizable.MulFunctional],
}

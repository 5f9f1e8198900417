"""Eager-mode model rewriting: swap modules for quantizable / QAT twins and attach per-tensor
fake-quantizers through forward-pre / backward hooks.

Same public functions and behaviour as upstream src/quantized_training/quantize.py:45-283
(``propagate_config``, ``quantize``, ``prepare``, ``convert``, ``swap_module``): hook
dictionaries are named ``activation_pre_process`` / ``error_pre_process`` /
``error_post_process`` and hold one fake-quantizer per tensor-argument index, created lazily on
the first call, so state-dict keys are identical.
"""
import copy
import logging

import torch
import torch.nn as nn
import torch.ao.nn.intrinsic as nni
from torch.nn.utils.parametrize import type_before_parametrizations

from .qconfig import get_qconfig
from .quantization_mappings import (
    DEFAULT_QAT_MODULE_MAPPINGS,
    QCONFIG_PROPAGATE_MODULE_CLASS_LIST,
    TRANSFORMER_MODULE_MAPPINGS,
)

__all__ = ["propagate_config", "quantize", "prepare", "convert", "swap_module", "replace_softmax",
           "get_quantized_model"]

logger = logging.getLogger(__name__)

# layers whose grad-inputs feed a residual branch in BERT / MobileBERT (upstream quantize.py:36-43)
RESIDUAL_LAYERS_BWD = (
    "attention.self.query",
    "attention.self.key",
    "attention.self.value",
    "intermediate.dense",
    "bottleneck.input.dense",
    "bottleneck.attention.dense",
)

_HOOK_KINDS = ("activation_pre_process", "error_pre_process", "error_post_process")


def propagate_config(module, name, qconfig):
    """setattr(module, name, qconfig) on the whole tree (upstream quantize.py:45-50)."""
    for m in module.modules():
        setattr(m, name, qconfig)


def _is_hf_model(model):
    try:
        from transformers import PretrainedConfig
    except Exception:  # noqa: BLE001
        return False
    return hasattr(model, "config") and isinstance(model.config, PretrainedConfig)


def quantize(model, args, inplace=True):
    """Prepare ``model`` for fake-quantized inference / training (upstream quantize.py:52-101)."""
    if not inplace:
        model = copy.deepcopy(model)

    wants_twins = (args.activation is not None or args.error is not None
                   or getattr(args, "posit_exp", False) or getattr(args, "posit_exp_shifted", False)
                   or getattr(args, "posit_reciprocal", False))
    if wants_twins and _is_hf_model(model):
        propagate_config(model, "config", model.config)
        convert(model, inplace=True, custom_module_class_mapping=TRANSFORMER_MODULE_MAPPINGS)

    if hasattr(model, "hf_device_map"):
        from accelerate import dispatch_model
        dispatch_model(model, device_map=model.hf_device_map)

    if getattr(args, "posit_exp", False) or getattr(args, "posit_exp_shifted", False) \
            or getattr(args, "posit_reciprocal", False):
        replace_softmax(model, args.posit_exp, args.posit_exp_shifted, args.posit_reciprocal)

    if getattr(args, "bf16", False):
        model.bfloat16()
    if args.activation is None:
        args.quantize_forward = None
    if args.error is None:
        args.quantize_backprop = None

    qconfig = get_qconfig(args.activation, args.weight, args.error,
                          getattr(args, "record_histogram", False),
                          getattr(args, "force_scale_power_of_two", False))
    propagate_config(model, "qconfig", qconfig)
    convert(model, mapping=DEFAULT_QAT_MODULE_MAPPINGS, inplace=True)
    prepare(model, True, args.quantize_forward, args.quantize_backprop, getattr(args, "op_fusion", None))
    if _is_hf_model(model):
        from .model_fusions import apply_bert_fusions, apply_llama_fusions
        apply_llama_fusions(model)          # inference-only one-launch RMSNorm / rotary / SiLU*up (no-op off device)
        apply_bert_fusions(model)           # same for BERT-style blocks: add + LayerNorm, GELU, q / k / v sibling group
    return model


def _parse_ops(op_str):
    ops = {op.lower() for op in op_str.split(",")} if op_str is not None else set()
    valid = set(QCONFIG_PROPAGATE_MODULE_CLASS_LIST)
    bad = ops - valid
    assert not bad, f"Invalid operation(s) {', '.join(bad)}. Options are {', '.join(valid)}."
    return tuple(cls for op in ops for cls in QCONFIG_PROPAGATE_MODULE_CLASS_LIST[op])


def _unique_devices(mod):
    return {p.device for p in mod.parameters()} | {b.device for b in mod.buffers()}


class _TensorArgQuantizer:
    """The hook body: fake-quantize every Tensor positional argument with its own lazily created
    fake-quantizer, keyed by argument index (upstream quantize.py:128-140)."""

    def __init__(self, owner_dict, ctr, qualified_name, forward_side=False):
        self.fqs = owner_dict
        self.ctr = ctr
        self.name = qualified_name
        self.forward_side = forward_side

    def __call__(self, module, tensors):
        out = []
        for i, t in enumerate(tensors):
            if not isinstance(t, torch.Tensor):
                out.append(t)
                continue
            key = str(i)
            if key not in self.fqs:
                fq = self.ctr(device=t.device)
                fq.name = f"{self.name}.{key}"
                self.fqs[key] = fq
                if self.forward_side:
                    from .fused import mark_fp8_producer
                    mark_fp8_producer(module, fq)
            out.append(self.fqs[key](t))
        return tuple(out)


def _register_module_hook(module, hook_name, name):
    assert hook_name in _HOOK_KINDS
    holder = nn.ModuleDict()
    module.add_module(hook_name, holder)
    ctr = module.qconfig.activation if hook_name == "activation_pre_process" else module.qconfig.error
    body = _TensorArgQuantizer(holder, ctr, name, forward_side=(hook_name == "activation_pre_process"))
    if hook_name == "activation_pre_process":
        module.register_forward_pre_hook(body)
    elif hook_name == "error_pre_process":
        module.register_full_backward_pre_hook(body)
    else:
        module.register_full_backward_hook(lambda mod, grad_in, grad_out: body(mod, grad_in))


def _add_observer_(module, fwd_classes, bwd_classes, bwd_residual, op_fusion, prefix):
    def attach(m, name):
        if getattr(m, "qconfig", None) is None:
            return
        if op_fusion is not None and any(tag in name for tag in op_fusion):
            return
        if isinstance(m, fwd_classes):
            _register_module_hook(m, "activation_pre_process", name)
        if isinstance(m, bwd_classes):
            _register_module_hook(m, "error_pre_process", name)
        if bwd_residual and (any(tag in name for tag in RESIDUAL_LAYERS_BWD)
                             or isinstance(m, _parse_ops("residual"))):
            _register_module_hook(m, "error_post_process", name)

    for child_name, child in list(module.named_children()):
        child_prefix = f"{prefix}.{child_name}" if prefix else child_name
        if isinstance(child, nni._FusedModule):
            attach(child, child_prefix)
        else:
            _add_observer_(child, fwd_classes, bwd_classes, bwd_residual, op_fusion, child_prefix)
    attach(module, prefix)


def prepare(model, inplace=False, fwd_quantized_ops=None, bwd_quantized_ops=None, op_fusion=None):
    """Attach the hooks (upstream quantize.py:181-193)."""
    if not inplace:
        model = copy.deepcopy(model)
    _add_observer_(model, _parse_ops(fwd_quantized_ops), _parse_ops(bwd_quantized_ops),
                   bool(bwd_quantized_ops) and "residual" in bwd_quantized_ops, op_fusion, prefix="")
    return model


def convert(module, mapping=None, inplace=False, custom_module_class_mapping=None):
    """Swap sub-modules according to ``mapping`` (``from_float``) and
    ``custom_module_class_mapping`` (``from_observed``) (upstream quantize.py:195-235)."""
    if not inplace:
        module = copy.deepcopy(module)
    mapping = DEFAULT_QAT_MODULE_MAPPINGS if mapping is None else mapping
    custom = custom_module_class_mapping or {}
    _convert(module, mapping, custom)
    return module


def _convert(module, mapping, custom):
    swapped = {}
    for name, child in module.named_children():
        if not isinstance(child, nni._FusedModule) and type_before_parametrizations(child) not in custom:
            _convert(child, mapping, custom)
        swapped[name] = swap_module(child, mapping, custom)
    for name, new in swapped.items():
        module._modules[name] = new


def swap_module(mod, mapping, custom_module_class_mapping):
    """Return the twin of ``mod`` (or ``mod`` itself), keeping its hooks and device
    (upstream quantize.py:237-283)."""
    kind = type_before_parametrizations(mod)
    if kind in custom_module_class_mapping:
        new = custom_module_class_mapping[kind].from_observed(mod)
    elif getattr(mod, "qconfig", None) is not None and kind in mapping:
        new = mapping[kind].from_float(mod)
    else:
        return mod
    if new is mod:                       # converted in place: hooks and device are already right
        return mod
    for fn in mod._forward_pre_hooks.values():
        new.register_forward_pre_hook(fn)
    for fn in mod._forward_hooks.values():
        new.register_forward_hook(fn)
    for fn in mod._backward_pre_hooks.values():
        new.register_full_backward_pre_hook(fn)
    for fn in mod._backward_hooks.values():
        new.register_full_backward_hook(fn)
    devices = _unique_devices(mod)
    assert len(devices) <= 1, f"swap_module only works with cpu or single-device CUDA modules, but got devices {devices}"
    if devices:
        new.to(next(iter(devices)))
    return new


def replace_softmax(module, posit_exp, posit_exp_shifted, posit_reciprocal, dtype=None, device=None):
    """Upstream swaps nn.Softmax for LUT-based posit16 exp / reciprocal approximations
    (modules/softmax.py) whose gold tables are missing from the public checkout
    (.MISSING_LARGE_BLOBS:3-5); they cannot be reproduced."""
    raise NotImplementedError("posit softmax approximations need gold tables that upstream does not ship")


def get_quantized_model(model, qconfig, op_fusion=None, device=None):
    """Upstream's legacy explicit-call API (quantize.py:305-339) depends on vendored BERT copies that
    do not import against current `transformers`; use `quantize(model, args)`."""
    raise NotImplementedError("use quantize(model, args)")

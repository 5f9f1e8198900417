"""Pattern annotators for the PT2E (torch.export) quantization flow.

Same roles as upstream src/quantized_training/quantizer/xnnpack_quantizer_utils.py:45-282: a
``QuantizationConfig`` names the spec of a GEMM's input activation / output / weight / bias, and each
annotator walks the exported ATen graph and records, on every matching node, which of its input edges (and
whether its output) receive a fake-quantizer.  torch.ao's ``prepare_pt2e`` then inserts the modules.
Covered patterns: linear, conv1d/2d, matmul, residual add (the ones ``get_default_quantizer`` enables) and the rest of
upstream's static list -- activation, softmax, layer_norm (:371-503), reached through ``set_global`` /
``set_module_name`` / ``set_object_type`` -- plus the registered-but-unlisted ``add`` / ``mul`` annotators (:285-368).
"""
from dataclasses import dataclass, replace
from typing import Callable, Dict, List, Optional

import torch
from torch.ao.quantization.quantizer import QuantizationAnnotation
from torch.fx import Node

from .quantizer import DerivedQuantizationSpec, QuantizationSpec

__all__ = ["QuantizationConfig", "OP_TO_ANNOTATOR", "STATIC_OPS"]


@dataclass(eq=True, frozen=True)
class QuantizationConfig:
    input_activation: Optional[QuantizationSpec]
    output_activation: Optional[QuantizationSpec]
    weight: Optional[QuantizationSpec]
    bias: Optional[QuantizationSpec]
    is_qat: bool = False      # kept for signature compatibility


_KEY = "quantization_annotation"


def _annotated(node: Node) -> bool:
    a = node.meta.get(_KEY)
    return a is not None and a._annotated


def _mark(node: Node):
    if _KEY not in node.meta:
        node.meta[_KEY] = QuantizationAnnotation()
    node.meta[_KEY]._annotated = True


def _set(node: Node, input_map: Dict[Node, object], output_spec):
    node.meta[_KEY] = QuantizationAnnotation(input_qspec_map=input_map, output_qspec=output_spec, _annotated=True)


def _bias_spec(cfg: QuantizationConfig, act: Node, weight: Node, user: Node):
    spec = cfg.bias
    if isinstance(spec, DerivedQuantizationSpec):       # scale derives from this GEMM's activation and weight edges
        spec = replace(spec, derived_from=[(act, user), (weight, user)])
    return spec


def _gemm_with_weight(targets):
    def annotate(gm, cfg: QuantizationConfig, filter_fn: Optional[Callable[[Node], bool]] = None):
        done = []
        for node in gm.graph.nodes:
            if node.op != "call_function" or node.target not in targets:
                continue
            if (filter_fn and not filter_fn(node)) or _annotated(node):
                continue
            act, weight = node.args[0], node.args[1]
            bias = node.args[2] if len(node.args) > 2 else None
            inputs = {act: cfg.input_activation, weight: cfg.weight}
            group = [node, weight]
            if isinstance(bias, Node):
                inputs[bias] = _bias_spec(cfg, act, weight, node)
                group.append(bias)
            _set(node, inputs, cfg.output_activation)
            for n in group[1:]:
                _mark(n)
            done.append(group)
        return done
    return annotate


def _annotate_matmul(gm, cfg: QuantizationConfig, filter_fn=None):
    done = []
    for node in gm.graph.nodes:
        if node.op != "call_function" or node.target != torch.ops.aten.matmul.default:
            continue
        if (filter_fn and not filter_fn(node)) or _annotated(node):
            continue
        inputs = {}
        if isinstance(node.args[0], Node):
            inputs[node.args[0]] = cfg.input_activation
        if isinstance(node.args[1], Node):          # the `weight` slot of the config describes the second operand
            inputs[node.args[1]] = cfg.weight
        _set(node, inputs, cfg.output_activation)
        done.append([node])
    return done


def _annotate_residual(gm, cfg: QuantizationConfig, filter_fn=None):
    """add(a, b) of two same-shape activations: the operand defined EARLIER in the graph (the skip branch)
    is the one that gets quantized (upstream :232-282)."""
    order = {n: i for i, n in enumerate(gm.graph.nodes)}
    done = []
    for node in gm.graph.nodes:
        if node.op != "call_function" or node.target not in (torch.ops.aten.add.Tensor, torch.ops.aten.add_.Tensor):
            continue
        if _annotated(node) or (filter_fn and not filter_fn(node)):
            continue
        a, b = node.args[0], node.args[1]
        if not isinstance(a, Node) or not isinstance(b, Node) or a.op == "get_attr" or b.op == "get_attr":
            continue
        if a.meta["val"].shape != b.meta["val"].shape:
            continue
        first = a if order[a] < order[b] else b
        _set(node, {first: cfg.input_activation}, cfg.output_activation)
        done.append([node])
    return done


def _unary_or_binary(targets, n_inputs):
    """Every Node among the first ``n_inputs`` arguments gets the input-activation spec (upstream's add / mul /
    softmax / activation annotators differ only in the targets they match)."""
    def annotate(gm, cfg: QuantizationConfig, filter_fn=None):
        done = []
        for node in gm.graph.nodes:
            if node.op != "call_function" or node.target not in targets:
                continue
            if _annotated(node) or (filter_fn and not filter_fn(node)):
                continue
            inputs = {a: cfg.input_activation for a in node.args[:n_inputs] if isinstance(a, Node)}
            _set(node, inputs, cfg.output_activation)
            done.append([node])
        return done
    return annotate


def _annotate_layer_norm(gm, cfg: QuantizationConfig, filter_fn=None):
    """aten.layer_norm(x, shape, weight, bias, ...): activation, weight and bias specs on the three tensor inputs
    (upstream :408-453)."""
    done = []
    for node in gm.graph.nodes:
        if node.op != "call_function" or node.target != torch.ops.aten.layer_norm.default:
            continue
        if _annotated(node) or (filter_fn and not filter_fn(node)):
            continue
        act, weight = node.args[0], node.args[2]
        assert isinstance(act, Node) and isinstance(weight, Node)
        inputs = {act: cfg.input_activation, weight: cfg.weight}
        bias = node.args[3] if len(node.args) > 3 else None
        if bias:
            assert isinstance(bias, Node)
            inputs[bias] = cfg.bias
        _set(node, inputs, cfg.output_activation)
        done.append([node])
    return done


_A = torch.ops.aten
_ACTIVATIONS = (_A.relu.default, _A.sigmoid.default, _A.tanh.default, _A.hardswish.default, _A.hardtanh.default,
                _A.silu.default, _A.gelu.default, _A.relu_.default, _A.sigmoid_.default, _A.tanh_.default,
                _A.hardswish_.default, _A.hardtanh_.default, _A.silu_.default, _A.gelu_.default)

OP_TO_ANNOTATOR = {
    "linear": _gemm_with_weight((torch.ops.aten.linear.default,)),
    "conv": _gemm_with_weight((torch.ops.aten.conv1d.default, torch.ops.aten.conv2d.default)),
    "matmul": _annotate_matmul,
    "residual": _annotate_residual,
    "add": _unary_or_binary((_A.add.Tensor, _A.add_.Tensor), 2),
    "mul": _unary_or_binary((_A.mul.Tensor, _A.mul_.Tensor), 2),
    "softmax": _unary_or_binary((_A.softmax.int,), 1),
    "layer_norm": _annotate_layer_norm,
    "activation": _unary_or_binary(_ACTIVATIONS, 1),
}
# the patterns the quantizer applies, in upstream's order (fusions before singular ops; xnnpack_quantizer.py:160-168)
STATIC_OPS = ["linear", "conv", "matmul", "residual", "activation", "softmax", "layer_norm"]

"""PT2E (torch.export) graph-mode quantization: the route the reference's current LLaMA driver uses
(examples/language_modeling/wikitext.py:68-136).

Same entry points as upstream src/quantized_training/quantize_pt2e.py:
  get_default_quantizer (:155-236)   spec strings -> an XNNPACKQuantizer for linear / conv / matmul / residual add
  export_model / prepare_pt2e (:239-273)  torch.export the model and let torch.ao insert one
                                     FusedAmaxObsFakeQuantize `call_module` per annotated edge
  convert_pt2e (:975-1002)           replace each fake-quant module by real
                                     quantize -> GEMM -> dequantize(s_x * s_w) nodes (:323-446)
The fake-quant modules and the quantize / dequantize operators are the same HIP-backed ones as in eager mode.
Not covered here: microscaling / group-wise conversion to *_mx operators and the dequantize-sinking
clean-up pass (accelerator code generation concerns).
"""
import copy
import logging
from dataclasses import asdict, replace
from typing import Any, Dict, List, Optional, Tuple

import torch
from torch import Tensor
from torch.fx import GraphModule, Node

from .fake_quantize import FusedAmaxObsFakeQuantize, _DerivedObserverOrFakeQuantize, _table_for
from .quantizer.quantizer import DerivedQuantizationSpec, QScheme, QuantizationSpec
from .quantizer.xnnpack_quantizer import XNNPACKQuantizer
from .quantizer.xnnpack_quantizer_utils import QuantizationConfig

__all__ = ["get_default_quantizer", "get_microscaling_quantizer", "get_per_channel_act_quantizer",
           "derive_bias_qparams_fn", "export_model", "prepare_pt2e", "convert_pt2e"]

logger = logging.getLogger(__name__)

_GEMM_TARGETS = (torch.ops.aten.conv2d.default, torch.ops.aten.linear.default, torch.ops.aten.matmul.default)


# ---- observer construction (replaces torch.ao's, which only knows its own QuantizationSpec) --------------
def _make_obs_or_fq(spec, made, is_qat):
    if spec is None:
        return None
    if isinstance(spec, DerivedQuantizationSpec):
        return _DerivedObserverOrFakeQuantize(spec.dtype, [made[k] for k in spec.derived_from], spec.derive_qparams_fn)
    assert isinstance(spec, QuantizationSpec)
    kwargs = copy.deepcopy(asdict(spec))
    ctr = spec.observer_or_fake_quant_ctr
    kwargs.pop("observer_or_fake_quant_ctr")
    return ctr(**kwargs)


def _get_obs_or_fq_map(edge_or_node_to_group_id, edge_or_node_to_qspec, is_qat):
    """One fake-quantizer per sharing group (upstream :69-87)."""
    made, by_group = {}, {}
    for edge_or_node, spec in edge_or_node_to_qspec.items():
        gid = edge_or_node_to_group_id[edge_or_node]
        if gid not in by_group:
            by_group[gid] = _make_obs_or_fq(spec, made, is_qat)
        made[edge_or_node] = by_group[gid]
    return made


# ---- quantizer factories -------------------------------------------------------------------------------------
def derive_bias_qparams_fn(obs_or_fqs) -> Tensor:
    """bias scale = activation scale * weight scale (upstream :145-152)."""
    assert len(obs_or_fqs) == 2, f"Expecting two obs/fqs, one for activation and one for weight, got: {len(obs_or_fqs)}"
    return obs_or_fqs[0].calculate_qparams() * obs_or_fqs[1].calculate_qparams().flatten()


def _with_axis(spec, axis):
    return None if spec is None else replace(spec, ch_axis=axis)


def get_microscaling_quantizer(activation, weight):
    """Blocks run along the reduction dimension of every GEMM (upstream :97-118)."""
    conv = QuantizationConfig(_with_axis(activation, 1), None, _with_axis(weight, 1), None)
    lin = QuantizationConfig(_with_axis(activation, -1), None, _with_axis(weight, -1), None)
    mm = QuantizationConfig(_with_axis(activation, -1), None, _with_axis(activation, -2), None)
    return (XNNPACKQuantizer()
            .set_object_type(torch.ops.aten.conv2d.default, conv)
            .set_object_type(torch.ops.aten.linear.default, lin)
            .set_object_type(torch.ops.aten.matmul.default, mm))


def get_per_channel_act_quantizer(input_activation, output_activation, weight, bias):
    """Per-channel activations: per-tensor for conv, outer dimension for linear / matmul (upstream :121-142)."""
    conv = QuantizationConfig(replace(input_activation, qscheme=QScheme.PER_TENSOR_SYMMETRIC), output_activation, weight, bias)
    lin = QuantizationConfig(replace(input_activation, ch_axis=-2), output_activation, weight, bias)
    mm = QuantizationConfig(replace(input_activation, ch_axis=-2), output_activation, replace(input_activation, ch_axis=-1), None)
    return (XNNPACKQuantizer()
            .set_object_type(torch.ops.aten.conv2d.default, conv)
            .set_object_type(torch.ops.aten.linear.default, lin)
            .set_object_type(torch.ops.aten.matmul.default, mm))


def get_default_quantizer(input_activation: Optional[str], output_activation: Optional[str] = None,
                          weight: Optional[str] = None, bias: Optional[str] = None, record_histogram: bool = False,
                          force_scale_power_of_two: bool = False, **kwargs: Any) -> XNNPACKQuantizer:
    """Spec strings -> quantizer for linear, conv2d, matmul and residual adds (upstream :155-236)."""
    ctr = FusedAmaxObsFakeQuantize.with_args(record_histogram=record_histogram,
                                             force_scale_power_of_two=force_scale_power_of_two)

    def parse(s):
        if s is None:
            return None
        spec = QuantizationSpec.from_str(s)
        return replace(spec, observer_or_fake_quant_ctr=ctr)

    input_activation, output_activation, weight = parse(input_activation), parse(output_activation), parse(weight)
    schemes = [s.qscheme for s in (input_activation, weight) if s is not None and s.qscheme is not None]
    if schemes and QScheme.MICROSCALING not in schemes:
        assert bias is not None, "Bias quantization is required when quantizing activations and weights."
    if bias is not None:       # the bias dtype stands for the accumulator type; its scale is derived per GEMM
        bias = DerivedQuantizationSpec(derived_from=None, derive_qparams_fn=derive_bias_qparams_fn, dtype=bias)
    if QScheme.MICROSCALING in schemes:
        assert len(set(schemes)) == 1, f"Quantization scheme {schemes[0]} does not work with {schemes[1]}"
        return get_microscaling_quantizer(input_activation, weight)
    if weight is not None and weight.qscheme == QScheme.PER_CHANNEL_SYMMETRIC:
        assert weight.ch_axis == 0, "Per-channel weight quantization only supports quantizing output channel dimension (dim=0)."
    if input_activation is not None and input_activation.qscheme == QScheme.PER_CHANNEL_SYMMETRIC:
        return get_per_channel_act_quantizer(input_activation, output_activation, weight, bias)
    gemm = QuantizationConfig(input_activation, output_activation, weight, bias)
    mm = QuantizationConfig(input_activation, output_activation, input_activation, None)
    return (XNNPACKQuantizer()
            .set_object_type(torch.ops.aten.conv2d.default, gemm)
            .set_object_type(torch.ops.aten.linear.default, gemm)
            .set_object_type(torch.ops.aten.matmul.default, mm)
            .set_object_type(torch.ops.aten.add.Tensor, gemm)
            .set_object_type(torch.ops.aten.add_.Tensor, gemm))


# ---- export + prepare --------------------------------------------------------------------------------------------
def export_model(model: torch.nn.Module, args: Tuple[Any, ...], kwargs: Optional[Dict[str, Any]] = None, *,
                 dynamic_shapes: Optional[Dict[str, Any]] = None):
    """Pre-autograd ATen graph of `model` (upstream :239-259)."""
    if hasattr(torch.export, "export_for_training"):
        return torch.export.export_for_training(model, args, kwargs, dynamic_shapes=dynamic_shapes).module()
    return torch.export.export(model, args, kwargs, dynamic_shapes=dynamic_shapes).module()


def prepare_pt2e(model, quantizer, args=None, kwargs=None, dynamic_shapes=None):
    """Export (unless already a GraphModule) and insert the fake-quant modules the quantizer asks for
    (upstream :262-273)."""
    from torch.ao.quantization.pt2e import prepare as _prepare
    from torch.ao.quantization.quantize_pt2e import prepare_pt2e as _torch_prepare_pt2e
    _prepare._get_obs_or_fq_map = _get_obs_or_fq_map          # torch.ao only constructs its own spec type
    if not isinstance(model, GraphModule):
        model = export_model(model, args, kwargs, dynamic_shapes=dynamic_shapes)
    return _torch_prepare_pt2e(model, quantizer)


# ---- convert -----------------------------------------------------------------------------------------------------
def _fresh_attr(module, prefix):
    prefix = prefix.replace(".", "_")
    name, i = prefix, 0
    while hasattr(module, name):
        i += 1
        name = f"{prefix}_{i}"
    return name


def _buffer_node(model, graph, prefix, value):
    name = _fresh_attr(model, prefix)
    model.register_buffer(name, value.clone().detach() if isinstance(value, Tensor) else torch.tensor(value))
    return graph.create_node("get_attr", name)


def _tag(node):
    node.meta.setdefault("source_fn_stack", []).append((node.name, node.target))


def _lower_fake_quant(model: GraphModule, node: Node, fq, output_dtype):
    """quantize -> user -> dequantize around one per-tensor / per-channel fake-quant node (upstream :323-446)."""
    graph = model.graph
    dtype = next(iter(model.parameters())).dtype
    device = next(iter(model.parameters())).device
    scale = fq.calculate_qparams().to(dtype)
    users = list(node.users.keys())
    src = node.args[0]
    qnode = None
    if src.op == "get_attr":
        # weights are stored as quantized codes; the fake-quant node disappears
        param = model.get_parameter(src.target)
        param.data = torch.ops.quantized_ops.quantize(param.data, scale.to(param.device), None, None, None,
                                                      fq.qmap.to(param.device))
        node.replace_all_uses_with(src)
        src.meta["dtype"] = fq.dtype
        if scale.ndim == 4:
            scale = scale.view(-1, 1, 1)
        elif scale.ndim == 2:
            scale = scale.view(-1)
    else:
        with graph.inserting_before(node):
            s_node = _buffer_node(model, graph, users[0].name + "_scale", scale)
            m_node = _buffer_node(model, graph, "qmap", fq.qmap)
            qnode = graph.call_function(torch.ops.quantized_ops.quantize.default, (src, s_node, None, None, None, m_node))
        _tag(qnode)
        qnode.meta["dtype"] = fq.dtype
        node.replace_all_uses_with(qnode)
    graph.erase_node(node)

    if isinstance(fq, _DerivedObserverOrFakeQuantize) or fq.qscheme is None:
        return                                        # bias, or a scale-free format: nothing to undo
    order = {n: i for i, n in enumerate(graph.nodes)}
    for user in users:
        if user.target in _GEMM_TARGETS:
            user.meta["dtype"] = output_dtype
            first = min(user.users.keys(), key=lambda n: order.get(n, float("inf")))
            if first.op == "call_function" and first.target == torch.ops.quantized_ops.dequantize.default:
                # second operand of the same GEMM: fold its scale into the existing dequantize
                buf_name = first.args[1].target
                model.register_buffer(buf_name, scale * model.get_buffer(buf_name))
                continue
            out_map = _table_for(output_dtype, device)
            with graph.inserting_before(first):
                s_node = _buffer_node(model, graph, user.name + "_scale", scale)
                m_node = _buffer_node(model, graph, "qmap", out_map)
                dq = graph.call_function(torch.ops.quantized_ops.dequantize.default,
                                         (user, s_node, None, None, None, m_node), {})
            _tag(dq)
            for consumer in list(user.users.keys()):
                if consumer is not dq:
                    consumer.replace_input_with(user, dq)
        else:
            with graph.inserting_before(qnode.next):
                s_node = _buffer_node(model, graph, user.name + "_scale", scale)
                dq = graph.call_function(torch.ops.quantized_ops.dequantize.default, (qnode, s_node), {})
            _tag(dq)
            user.replace_input_with(qnode, dq)


def convert_pt2e(model: GraphModule, output_dtype: str = None, eliminate_no_effect: bool = True):
    """Lower every FusedAmaxObsFakeQuantize `call_module` of a prepared (and calibrated) graph to
    quantized_ops.quantize / dequantize nodes (upstream :975-1002)."""
    modules = dict(model.named_modules(remove_duplicate=False))
    for node in list(model.graph.nodes):
        if node.op != "call_module":
            continue
        mod = modules.get(str(node.target))
        if not isinstance(mod, torch.ao.quantization.FakeQuantizeBase):
            continue
        if mod.qscheme in (QScheme.MICROSCALING, QScheme.GROUP_WISE_AFFINE):
            raise NotImplementedError("convert_pt2e lowers per-tensor / per-channel fake-quantizers; block-scaled "
                                      "formats stay fake-quantized (prepare_pt2e output) in this engine")
        _lower_fake_quant(model, node, mod, output_dtype)
    model.graph.lint()
    model.graph.eliminate_dead_code(is_impure_node=lambda n: n.op in {"placeholder", "output"})
    model.recompile()
    model.delete_all_unused_submodules()
    return model

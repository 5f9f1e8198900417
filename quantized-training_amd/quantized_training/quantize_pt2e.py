"""PT2E (torch.export) graph-mode quantization: the route the reference's current LLaMA driver uses
(examples/language_modeling/wikitext.py:68-136).

Same entry points as upstream src/quantized_training/quantize_pt2e.py:
  get_default_quantizer (:155-236)   spec strings -> an XNNPACKQuantizer for linear / conv / matmul / residual add
  export_model / prepare_pt2e (:239-273)  torch.export the model and let torch.ao insert one
                                     FusedAmaxObsFakeQuantize `call_module` per annotated edge
  convert_pt2e (:975-1002)           replace each fake-quant module by real
                                     quantize -> GEMM -> dequantize(s_x * s_w) nodes (:323-446)
The fake-quant modules and the quantize / dequantize operators are the same HIP-backed ones as in eager mode.
Block-scaled fake-quantizers lower to quantize_mx / calculate_mx_qparam + quantize nodes whose consumers become
linear_mx / matmul_mx / conv2d_mx (:456-700); group-wise affine weights lower to stored codes + dequantize (:754-826).
Not covered here: the dequantize-sinking
clean-up pass, which depends on the accelerator code generator.
"""
import copy
import logging
import operator
import re
from collections import OrderedDict
from dataclasses import asdict, replace
from typing import Any, Dict, List, Optional, Tuple

import torch
from torch import Tensor
from torch.fx import GraphModule, Node

from .fake_quantize import FusedAmaxObsFakeQuantize, _DerivedObserverOrFakeQuantize, _table_for, get_quantization_map
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


def prepare_pt2e(model, quantizer, args=None, kwargs=None, dynamic_shapes=None, fuse=None):
    """Export (unless already a GraphModule) and insert the fake-quant modules the quantizer asks for
    (upstream :262-273).

    The graph returned for a model on the CPU is upstream's node for node.  For a DEVICE model (`fuse=None`: any parameter on a GPU;
    `fuse=True/False` forces it) the chains between the fake-quantizer nodes are then rewritten to the fused HIP kernels
    (pt2e_fusion.fuse_prepared_graph: same values, fewer launches); `convert_pt2e` restores the plain graph first.  A model
    prepared on the CPU and moved afterwards takes the same route through `pt2e_fusion.fuse_prepared_graph(model)`."""
    from torch.ao.quantization.pt2e import prepare as _prepare
    from torch.ao.quantization.quantize_pt2e import prepare_pt2e as _torch_prepare_pt2e
    _prepare._get_obs_or_fq_map = _get_obs_or_fq_map          # torch.ao only constructs its own spec type
    if not isinstance(model, GraphModule):
        # upstream's driver hands CPU example ids to a model that already lives on a GPU (wikitext.py:81-96); this torch's exporter
        # refuses mixed devices, so the examples follow the model
        dev = next((p.device for p in model.parameters()), None)
        if dev is not None and dev.type == "cuda":
            from torch.utils._pytree import tree_map
            follow = lambda t: t.to(dev) if isinstance(t, torch.Tensor) and t.device != dev else t      # noqa: E731
            args, kwargs = tree_map(follow, args), tree_map(follow, kwargs)
        model = export_model(model, args, kwargs, dynamic_shapes=dynamic_shapes)
        constants = {k: v for k, v in (kwargs or {}).items() if not isinstance(v, torch.Tensor)}
    else:
        constants = {}
    model = _torch_prepare_pt2e(model, quantizer)
    if constants:
        # the exported forward takes every example keyword as a required argument; upstream's calibration loop calls
        # `model(input_ids, labels=target_ids)` (wikitext.py:128), so the non-tensor example keywords (`use_cache=False`) become defaults
        def _fill_constants(module, call_args, call_kwargs, _c=constants):
            missing = {k: v for k, v in _c.items() if k not in call_kwargs}
            return (call_args, {**call_kwargs, **missing}) if missing else None
        model.register_forward_pre_hook(_fill_constants, with_kwargs=True)
    if fuse or (fuse is None and any(p.device.type == "cuda" for p in model.parameters())):
        from . import pt2e_fusion
        pt2e_fusion.fuse_prepared_graph(model)
    return model


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
    buf = value.clone().detach() if isinstance(value, Tensor) else torch.tensor(value)
    for tag in ("_qt_dtype", "_qt_pow2", "_qt_mx_fmt"):        # hints for the block-scaled GEMMs (mx_gemm.py)
        if hasattr(value, tag):
            setattr(buf, tag, getattr(value, tag))
    model.register_buffer(name, buf)
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


def _mx_op_mapping():
    return {torch.ops.aten.conv2d.default: torch.ops.quantized_ops.conv2d_mx.default,
            torch.ops.aten.linear.default: torch.ops.quantized_ops.linear_mx.default,
            torch.ops.aten.matmul.default: torch.ops.quantized_ops.matmul_mx.default}


def _param_or_buffer(model, target):
    try:
        return model.get_parameter(target)
    except AttributeError:
        return model.get_buffer(target)


def _lower_mx_fake_quant(model: GraphModule, node: Node, fq):
    """One microscaling fake-quant node -> (block scales, element codes) + a block-scaled GEMM consumer
    (upstream :456-700).  Weights are quantized once and stored with their scales; activations get a fused
    quantize_mx when the blocks run along the last axis (channel axis for a conv), else calculate_mx_qparam +
    quantize.  Each consumer (linear / matmul / conv2d) becomes its *_mx twin with the scale / code kwargs."""
    graph = model.graph
    device = next(iter(fq.buffers())).device
    src = node.args[0]
    src_dtype = fq.dtype
    if isinstance(fq.ch_axis, int):
        fq.ch_axis = (fq.ch_axis,)
    fuse = False
    if len(fq.ch_axis) == 1:
        axis = fq.ch_axis[0]
        if any(u.target == torch.ops.aten.conv2d.default for u in node.users):
            fuse = axis in (1, -3)
        else:
            fuse = axis == -1 or ("val" in src.meta and axis == src.meta["val"].ndim - 1)

    to_quantize = src
    csr = None
    if fq.outlier_threshold is not None:                 # outliers leave through a CSR side path (upstream :489-510)
        assert src.op != "get_attr", "Outlier suppression is not supported for weight quantization."
        with graph.inserting_before(node):
            flt = graph.call_function(torch.ops.quantized_ops.filter_outlier.default, (src, fq.outlier_threshold), {})
            to_quantize = graph.call_function(operator.getitem, (flt, 0), {})
            csr = tuple(graph.call_function(operator.getitem, (flt, i), {}) for i in (1, 2, 3))

    quant_map = get_quantization_map(fq.dtype, device)
    dequant_code = quant_code = None
    if isinstance(quant_map, tuple):                     # NormalFloat: indices in the graph, values as a codebook
        digits = re.findall(r"\d+", fq.dtype)
        fq.dtype = f"int{digits[0]}"
        indices, values = quant_map
        fq.qmap = indices
        with graph.inserting_before(node):
            dequant_code = _buffer_node(model, graph, "code", values)
            if src.op != "get_attr":
                quant_code = _buffer_node(model, graph, "code", (values[:-1] + values[1:]) / 2)
        if len(digits) > 1:
            dequant_code.meta["dtype"] = f"int{digits[1]}"

    if src.op == "get_attr":
        param = _param_or_buffer(model, src.target)
        scale = torch.ops.quantized_ops.calculate_mx_qparam(param.data, list(fq.ch_axis), fq.block_size, fq.quant_max,
                                                            fq.force_scale_power_of_two, fq.scale_qmap)
        weight = torch.ops.quantized_ops.quantize(param.data, scale, None, list(fq.ch_axis), fq.block_size, fq.qmap)
        with graph.inserting_before(node):
            q_node = _buffer_node(model, graph, src.name + "_" + src_dtype, weight)
            s_node = _buffer_node(model, graph, src.name + "_scale", scale)
    elif fuse:
        with graph.inserting_before(node):
            m_node = _buffer_node(model, graph, "qmap", fq.qmap)
            sm_node = _buffer_node(model, graph, "qmap", fq.scale_qmap) if fq.scale_qmap is not None else None
            mx = graph.call_function(torch.ops.quantized_ops.quantize_mx.default,
                                     (to_quantize, m_node, fq.ch_axis, fq.block_size, fq.quant_max, fq.force_scale_power_of_two,
                                      sm_node, quant_code))
            s_node = graph.call_function(operator.getitem, (mx, 0))
            q_node = graph.call_function(operator.getitem, (mx, 1))
        mx.meta["dtype"] = ("fp8_e8m0" if fq.force_scale_power_of_two else fq.scale_dtype, fq.dtype)
        _tag(mx)
    else:
        with graph.inserting_before(node):
            args = [src, fq.ch_axis, fq.block_size, fq.quant_max, fq.force_scale_power_of_two]
            if fq.scale_qmap is not None:
                args.append(_buffer_node(model, graph, "qmap", fq.scale_qmap))
            s_node = graph.call_function(torch.ops.quantized_ops.calculate_mx_qparam.default, tuple(args), {})
            if fq.scale_dtype is not None:
                s_node.meta["dtype"] = fq.scale_dtype
            m_node = _buffer_node(model, graph, "qmap", fq.qmap)
            q_node = graph.call_function(torch.ops.quantized_ops.quantize.default,
                                         (src, s_node, None, fq.ch_axis, fq.block_size, m_node, quant_code))
        _tag(q_node)
    q_node.meta["dtype"] = fq.dtype
    if fq.force_scale_power_of_two:
        s_node.meta["dtype"] = "fp8_e8m0"
    elif fq.scale_dtype is not None:
        s_node.meta["dtype"] = fq.scale_dtype

    users = list(node.users.keys())
    node.replace_all_uses_with(q_node)
    graph.erase_node(node)
    if len(src.users) == 0:
        graph.erase_node(src)

    mapping = _mx_op_mapping()
    for user in users:
        code_arg, scale_arg = dequant_code, s_node
        if user.target == torch.Tensor.to:               # device-alignment hop: move the side inputs with it
            with graph.inserting_before(user):
                if code_arg is not None:
                    code_arg = graph.call_function(torch.Tensor.to, (dequant_code, user.args[1]))
                scale_arg = graph.call_function(torch.Tensor.to, (s_node, user.args[1]))
            user = next(iter(user.users))
        kwargs = OrderedDict(user.kwargs)
        kwargs.setdefault("block_size", fq.block_size)
        if src.op == "get_attr" or (len(user.args) > 1 and q_node is user.args[1]):
            kwargs.setdefault("weight_code", code_arg)
            kwargs.setdefault("weight_scale", scale_arg)
        else:
            kwargs.setdefault("input_code", code_arg)
            kwargs.setdefault("input_scale", scale_arg)
        order = ["input_scale", "weight_scale", "block_size", "input_code", "weight_code"]
        kwargs = OrderedDict((k, kwargs[k]) for k in order if k in kwargs)
        if user.target in mapping:
            with graph.inserting_before(user):
                mx_op = graph.call_function(mapping[user.target], user.args, kwargs)
            user.replace_all_uses_with(mx_op)
            graph.erase_node(user)
            mx_op.meta = user.meta
            stack = mx_op.meta.setdefault("source_fn_stack", [])
            stack.append((mx_op.name, stack[-1][1] if stack else mx_op.target))
        elif user.target in mapping.values():
            mx_op = user
            user.kwargs = kwargs
        elif user.target == torch.ops.quantized_ops.spmm_csr.default:        # the weight of an outlier side path (:705-716)
            assert src.op == "get_attr", f"Expect input node to be a get_attr, but found {src.op}"
            user.args = user.args[:-1] + (q_node,)
            user.kwargs = {"B_scale": kwargs.get("weight_scale"), "B_code": kwargs.get("weight_code"),
                           "block_size": fq.block_size}
            continue
        else:
            raise RuntimeError(f"Unsupported user node {user.target} for quantization, expected one of "
                               f"{list(mapping.keys())}")
        if csr is not None and src.op != "get_attr":     # y = linear_mx(inliers) + outliers_csr @ W^T (:721-750)
            assert mx_op.target in (torch.ops.aten.linear.default, torch.ops.quantized_ops.linear_mx.default), \
                f"Only torch.nn.Linear is supported for outlier suppresion, got {user.target}"
            with graph.inserting_after(mx_op):
                spmm = graph.call_function(torch.ops.quantized_ops.spmm_csr.default, csr + (mx_op.args[1],),
                                           {"B_scale": kwargs.get("weight_scale"), "B_code": kwargs.get("weight_code"),
                                            "block_size": fq.block_size})
            with graph.inserting_after(spmm):
                add = graph.call_function(torch.ops.aten.add.Tensor, (spmm, mx_op), {})
            mx_op.replace_all_uses_with(add)
            add.replace_input_with(add, mx_op)


def _lower_group_wise_affine(model: GraphModule, node: Node, fq):
    """Group-wise affine weight fake-quant -> stored codes, scales, zero points + one dequantize node
    (upstream :754-826; activations are not supported there either)."""
    graph = model.graph
    if isinstance(fq.ch_axis, int):
        fq.ch_axis = (fq.ch_axis,)
    src = node.args[0]
    if src.op != "get_attr":
        raise NotImplementedError
    param = _param_or_buffer(model, src.target)
    fq(param.data)
    scale, zero_point = fq.calculate_qparams()
    scale, zero_point = scale.to(param.data.dtype), zero_point.to(param.data.dtype)
    weight = torch.ops.quantized_ops.quantize(param.data, scale, zero_point, list(fq.ch_axis), fq.block_size, fq.qmap)
    with graph.inserting_before(node):
        q_node = _buffer_node(model, graph, src.name + "_" + fq.dtype, weight)
        s_node = _buffer_node(model, graph, src.name + "_scale", scale)
        z_node = _buffer_node(model, graph, src.name + "_zero_point", zero_point)
    q_node.meta["dtype"] = fq.dtype
    if fq.scale_dtype is not None:
        s_node.meta["dtype"] = z_node.meta["dtype"] = fq.scale_dtype
    with graph.inserting_before(node):
        dq = graph.call_function(torch.ops.quantized_ops.dequantize.default, (q_node, s_node, z_node, fq.ch_axis, fq.block_size))
    _tag(dq)
    node.replace_all_uses_with(dq)
    graph.erase_node(node)
    if len(src.users) == 0:
        graph.erase_node(src)


def _eliminate_dequantize_with_no_effect(model: GraphModule):
    """Drop dequantize nodes whose stored scale is all ones and that do not re-quantize their output (upstream :829-853)."""
    for node in list(model.graph.nodes):
        if node.target != torch.ops.quantized_ops.dequantize.default:
            continue
        s_node = node.args[1]
        if s_node.op != "get_attr" or torch.any(model.get_buffer(s_node.target) != 1):
            continue
        out_map = node.args[6] if len(node.args) > 6 else node.kwargs.get("output_qmap")
        if out_map is not None:
            continue
        node.replace_all_uses_with(node.args[0])
        model.graph.erase_node(node)
        logger.info(f"Eliminate dequantize node {node} with no effect")
    model.graph.lint()
    model.graph.eliminate_dead_code()
    model.recompile()
    return model


def convert_pt2e(model: GraphModule, output_dtype: str = None, eliminate_no_effect: bool = True, native=None):
    """Lower every FusedAmaxObsFakeQuantize `call_module` of a prepared (and calibrated) graph to
    quantized_ops nodes (upstream :975-1002): quantize / dequantize for per-tensor and per-channel specs,
    quantize_mx + *_mx GEMMs for microscaling, stored codes + dequantize for group-wise affine weights."""
    from . import pt2e_fusion
    pt2e_fusion.unfuse_prepared_graph(model)                  # the lowering below works on the plain prepared graph
    modules = dict(model.named_modules(remove_duplicate=False))
    for node in list(model.graph.nodes):
        if node.op != "call_module":
            continue
        mod = modules.get(str(node.target))
        if not isinstance(mod, torch.ao.quantization.FakeQuantizeBase):
            continue
        if mod.qscheme == QScheme.MICROSCALING:
            _lower_mx_fake_quant(model, node, mod)
        elif mod.qscheme == QScheme.GROUP_WISE_AFFINE:
            _lower_group_wise_affine(model, node, mod)
        else:
            _lower_fake_quant(model, node, mod, output_dtype)
    if eliminate_no_effect:
        _eliminate_dequantize_with_no_effect(model)
    model.graph.lint()
    model.graph.eliminate_dead_code(is_impure_node=lambda n: n.op in {"placeholder", "output"})
    model.recompile()
    model.delete_all_unused_submodules()
    # Device models: the quantize -> GEMM -> dequantize triples of per-tensor int8 / FP8 specs run on the integer / FP8 matrix
    # cores (pt2e_native.py).  The graph above is upstream's node for node; this pass only runs where it can execute natively.
    from . import pt2e_native
    if native or (native is None and pt2e_native.enabled_for(model)):
        pt2e_native.fuse_native_gemms(model)
    return model

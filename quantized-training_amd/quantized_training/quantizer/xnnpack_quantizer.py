"""Quantizer object of the PT2E flow: maps (module name / module type / operator / global) selectors to
``QuantizationConfig``s and annotates an exported graph accordingly.  Same builder methods and
precedence as upstream src/quantized_training/quantizer/xnnpack_quantizer.py:163-279
(name+operator+index, then module name, module type, operator, global)."""
import re
from collections import defaultdict
from typing import Callable, Dict, List, Optional, Tuple, Union

import torch
from torch.ao.quantization.quantizer import Quantizer
from torch.fx import Node

from .xnnpack_quantizer_utils import OP_TO_ANNOTATOR, STATIC_OPS, QuantizationConfig

__all__ = ["XNNPACKQuantizer", "get_node_name_to_scope"]


def _strip(path: str) -> str:
    return path[len("L['self']."):] if path.startswith("L['self'].") else path


def get_node_name_to_scope(model) -> Dict[str, Tuple[str, type, int]]:
    """node name -> (innermost module path, module type, ordinal of this operator inside that module)
    (upstream pt2e_utils.py:352-370)."""
    scope = {}
    counters = defaultdict(lambda: defaultdict(int))
    for n in model.graph.nodes:
        stack = n.meta.get("nn_module_stack")
        if stack is None:
            scope[n.name] = ("", type(None), 0)
            continue
        cur = None
        for path, tp in stack.values():
            idx = counters[path][n.target]
            counters[path][n.target] += 1
            cur = (path, tp, idx)
        scope[n.name] = cur
    return scope


def _name_filter(module_name: str):
    def f(n: Node) -> bool:
        names = [_strip(p) for p, _ in n.meta.get("nn_module_stack", {}).values()]
        return module_name in names or any(re.search(module_name, nm) for nm in names)
    return f


def _type_filter(tp: Callable):
    want = tp.__module__ + "." + tp.__qualname__

    def f(n: Node) -> bool:
        seen = []
        for _, t in n.meta.get("nn_module_stack", {}).values():
            seen.append(t.__module__ + "." + t.__qualname__ if isinstance(t, type) else t)
        return want in seen
    return f


class XNNPACKQuantizer(Quantizer):
    STATIC_OPS = STATIC_OPS

    def __init__(self):
        super().__init__()
        self.global_config: Optional[QuantizationConfig] = None
        self.object_type_config: Dict[Union[Callable, str], Optional[QuantizationConfig]] = {}
        self.module_type_config: Dict[Callable, Optional[QuantizationConfig]] = {}
        self.module_name_config: Dict[str, Optional[QuantizationConfig]] = {}
        self.module_name_object_type_order_config: Dict[Tuple[str, Callable, int], Optional[QuantizationConfig]] = {}

    def set_global(self, quantization_config):
        self.global_config = quantization_config
        return self

    def set_object_type(self, object_type, quantization_config):
        self.object_type_config[object_type] = quantization_config
        return self

    def set_module_type(self, module_type, quantization_config):
        self.module_type_config[module_type] = quantization_config
        return self

    def set_module_name(self, module_name, quantization_config):
        self.module_name_config[module_name] = quantization_config
        return self

    def set_module_name_object_type_order(self, module_name, object_type, index, quantization_config):
        self.module_name_object_type_order_config[(module_name, object_type, index)] = quantization_config
        return self

    def transform_for_annotation(self, model):
        """Python scalars in add / mul / div become 0-d buffers (``_tensor_constant_<i>``) read through get_attr nodes,
        so that these ops only have tensor operands when they are annotated (upstream xnnpack_quantizer.py:225-229,
        xnnpack_quantizer_utils.py:506-541)."""
        targets = (torch.ops.aten.add.Tensor, torch.ops.aten.mul.Tensor, torch.ops.aten.div.Tensor)
        devices = {t.device for t in list(model.parameters()) + list(model.buffers())}
        assert len(devices) <= 1, f"expected the model on one device, got {devices}"
        device = next(iter(devices)) if devices else None
        counter = 0
        for node in list(model.graph.nodes):
            if node.op != "call_function" or node.target not in targets or all(isinstance(a, Node) for a in node.args):
                continue
            dtypes = {a.meta["val"].dtype for a in node.all_input_nodes}
            assert len(dtypes) <= 1
            dtype = next(iter(dtypes)) if dtypes else None
            args = []
            for a in node.args:
                if isinstance(a, Node):
                    args.append(a)
                    continue
                while hasattr(model, f"_tensor_constant_{counter}"):
                    counter += 1
                name = f"_tensor_constant_{counter}"
                value = torch.tensor(float(a), dtype=dtype, device=device)
                model.register_buffer(name, value)
                with model.graph.inserting_before(node):
                    const = model.graph.create_node("get_attr", name, (), {})
                const.meta["val"] = node.meta["val"].fake_mode.from_tensor(value, static_shapes=True)
                args.append(const)
            node.args = tuple(args)
        model.recompile()
        return model

    def _apply(self, model, config, filter_fn):
        if config is None:                  # "None" = matched nodes are explicitly left unquantized
            config = QuantizationConfig(None, None, None, None)
        for op in self.STATIC_OPS:
            OP_TO_ANNOTATOR[op](model, config, filter_fn)

    def annotate(self, model):
        if self.module_name_object_type_order_config:
            scope = get_node_name_to_scope(model)
        for (name, op, index), cfg in self.module_name_object_type_order_config.items():
            def f(n, name=name, op=op, index=index):
                cur = scope[n.name]
                return (name == cur[0] or re.search(name, cur[0]) is not None) and op == n.target and index == cur[2]
            self._apply(model, cfg, f)
        for name, cfg in self.module_name_config.items():
            self._apply(model, cfg, _name_filter(name))
        for tp, cfg in self.module_type_config.items():
            self._apply(model, cfg, _type_filter(tp))
        for op, cfg in self.object_type_config.items():
            self._apply(model, cfg, lambda n, op=op: n.target == op)
        if self.global_config is not None:
            excl = [_type_filter(t) for t in self.module_type_config] + [_name_filter(m) for m in self.module_name_config]
            self._apply(model, self.global_config, lambda n: not any(f(n) for f in excl))
        return model

    def validate(self, model) -> None:
        pass

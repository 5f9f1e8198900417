"""PT2E route (get_default_quantizer / prepare_pt2e / convert_pt2e) against traces of the reference's own
flow on a toy model: identical prepared graph (which edges get a fake-quantizer, sharing, module names),
calibrated outputs and scales, converted graph and outputs.  CPU tensors."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import quantized_training as qt
from quantized_training import quantize_pt2e as qp

G = os.path.join(os.path.dirname(__file__), "golden")
META = json.load(open(os.path.join(G, "pt2e.json")))


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(16, 32)
        self.fc2 = nn.Linear(32, 16)
        self.ln = nn.LayerNorm(16)

    def forward(self, x):
        h = torch.relu(self.fc1(x))
        y = self.fc2(h)
        y = y + x
        a = torch.matmul(y, y.transpose(-1, -2))
        return self.ln(torch.matmul(torch.softmax(a, -1), y))


def _rows(gm):
    def nm(a):
        if isinstance(a, torch.fx.Node):
            return a.name
        if isinstance(a, (list, tuple)):
            return [nm(x) for x in a]
        return a if isinstance(a, (int, float, str, bool, type(None))) else str(a)
    return [[n.op, n.name, str(n.target), nm(list(n.args))] for n in gm.graph.nodes]


def _canon32(t):
    b = t.detach().float().contiguous().view(torch.int32).numpy().view(np.uint32).copy()
    b[((b & 0x7F800000) == 0x7F800000) & ((b & 0x7FFFFF) != 0)] = 0x7FC00000
    return b


def _model(arr):
    m = Toy().eval()
    r = np.random.default_rng(3)
    with torch.no_grad():
        for n, p in sorted(m.named_parameters()):
            p.copy_(torch.from_numpy((r.standard_normal(tuple(p.shape)) * 0.3).astype(np.float32)))
    for n, p in m.named_parameters():
        assert np.array_equal(_canon32(p), arr["param__" + n].reshape(-1).view(np.uint32).reshape(p.shape))
    return m


@pytest.mark.parametrize("name", sorted(META))
def test_pt2e_flow_matches_reference(name):
    arr = np.load(os.path.join(G, "pt2e.npz"))
    info = META[name]
    xs = [torch.from_numpy(arr[f"x{i}"].view(np.float32)).reshape(4, 8, 16) for i in range(4)]
    gm = qp.prepare_pt2e(_model(arr), qp.get_default_quantizer(**info["kw"]), (xs[0],))
    assert _rows(gm) == info["prepared_graph"]
    fq = [n for n, mod in gm.named_modules() if isinstance(mod, torch.ao.quantization.FakeQuantizeBase)]
    assert fq == info["fq_modules"]
    with torch.no_grad():
        for i in range(3):
            gm(xs[i])
        y1 = gm(xs[3])
    assert np.array_equal(_canon32(y1).reshape(-1), arr[f"{name}__y_prepared"].reshape(-1))
    scales = {k: [float(t) for t in v.reshape(-1)] for k, v in gm.state_dict().items() if k.endswith(".scale")}
    assert scales == info["scales"]
    gc = qp.convert_pt2e(gm, info["output_dtype"]) if info["output_dtype"] else qp.convert_pt2e(gm)
    assert [r[2] for r in _rows(gc)] == [r[2] for r in info["converted_graph"]]
    assert [r[1] for r in _rows(gc)] == [r[1] for r in info["converted_graph"]]
    with torch.no_grad():
        y2 = gc(xs[3])
    assert np.array_equal(_canon32(y2).reshape(-1), arr[f"{name}__y_converted"].reshape(-1))
    bufs = {k: [float(t) for t in v.reshape(-1)][:4] for k, v in gc.named_buffers() if "scale" in k}
    assert bufs == info["converted_buffers"]


def test_quantizer_selectors_and_exclusion():
    """set_module_name_object_type_order(name, op, index, None) leaves that node unquantized (the reference's
    LLaMA driver excludes the rotary-embedding matmul this way, wikitext.py:76-78)."""
    class Two(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Linear(8, 8)
            self.b = nn.Linear(8, 8)

        def forward(self, x):
            return self.b(torch.relu(self.a(x)))

    q = qp.get_default_quantizer("int8,qs=per_tensor_symmetric", None, "int8,qs=per_tensor_symmetric", "int24")
    q.set_module_name_object_type_order(r"^b$", torch.ops.aten.linear.default, 0, None)
    gm = qp.prepare_pt2e(Two().eval(), q, (torch.randn(2, 8),))
    lin = [n for n in gm.graph.nodes if n.target == torch.ops.aten.linear.default]
    assert lin[0].args[0].op == "call_module" and lin[1].args[0].op != "call_module"
    with pytest.raises(AssertionError):
        qp.get_default_quantizer("int8,qs=per_tensor_symmetric", None, "int8,qs=per_tensor_symmetric", None)

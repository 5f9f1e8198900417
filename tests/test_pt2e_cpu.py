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


# ---- block-scaled specs: quantize_mx / linear_mx / matmul_mx graphs (upstream quantize_pt2e.py:456-700) --------
META_MX = json.load(open(os.path.join(G, "pt2e_mx.json")))


class ToyMX(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(64, 128)
        self.fc2 = nn.Linear(128, 64)

    def forward(self, x):
        h = torch.relu(self.fc1(x))
        y = self.fc2(h) + x
        a = torch.matmul(y, y.transpose(-1, -2))
        return torch.matmul(torch.softmax(a, -1), y)


def _bits(t):
    if t.dtype == torch.bfloat16:
        b = t.detach().contiguous().view(torch.int16).numpy().view(np.uint16).copy()
        b[((b & 0x7F80) == 0x7F80) & ((b & 0x7F) != 0)] = 0x7FC0
        return b
    return _canon32(t)


def mx_setup(name, arr, device="cpu"):
    info = META_MX[name]
    dt = getattr(torch, info["dtype"])
    m = ToyMX().eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(torch.from_numpy(arr["param__" + n].view(np.float32)))
    xs = [torch.from_numpy(arr[f"x{i}"].view(np.float32)).to(dt).to(device) for i in range(2)]
    return info, m.to(dt).to(device), xs


@pytest.mark.parametrize("name", sorted(META_MX))
def test_pt2e_microscaling_flow_matches_reference(name):
    arr = np.load(os.path.join(G, "pt2e_mx.npz"))
    info, m, xs = mx_setup(name, arr)
    gm = qp.prepare_pt2e(m, qp.get_default_quantizer(**info["kw"]), (xs[0],))
    assert _rows(gm) == info["prepared_graph"]
    fq = {n: [mod.dtype, str(mod.ch_axis), str(mod.block_size)] for n, mod in gm.named_modules()
          if isinstance(mod, torch.ao.quantization.FakeQuantizeBase)}
    assert fq == info["fq_modules"]
    with torch.no_grad():
        gm(xs[0])
        y1 = gm(xs[1])
    assert np.array_equal(_bits(y1), arr[f"{name}__y_prepared"])
    gc = qp.convert_pt2e(gm)
    assert _rows(gc) == info["converted_graph"]
    kw = {n.name: {k: (v.name if isinstance(v, torch.fx.Node) else v) for k, v in n.kwargs.items()}
          for n in gc.graph.nodes if n.kwargs}
    assert kw == info["converted_kwargs"]
    nd = {n.name: (list(n.meta["dtype"]) if isinstance(n.meta["dtype"], tuple) else n.meta["dtype"])
          for n in gc.graph.nodes if n.meta.get("dtype") is not None}
    assert nd == info["node_dtype"]
    bufs = dict(gc.named_buffers())
    assert {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in bufs.items()} == info["buffers"]
    for k, v in bufs.items():
        assert np.array_equal(_bits(v), arr[f"{name}__buf__{k}"]), k
    with torch.no_grad():
        y2 = gc(xs[1])
    assert np.array_equal(_bits(y2), arr[f"{name}__y_converted"])


def test_group_wise_affine_weights_convert_to_codes_and_dequantize():
    """Weights with a group-wise affine spec become stored codes + scale + zero point and one dequantize node
    (upstream quantize_pt2e.py:754-826); the converted module computes what the prepared one did."""
    from quantized_training.quantizer.quantizer import QuantizationSpec
    from quantized_training.quantizer.xnnpack_quantizer import XNNPACKQuantizer
    from quantized_training.quantizer.xnnpack_quantizer_utils import QuantizationConfig
    from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize
    w = QuantizationSpec.from_str("uint4,qs=group_wise_affine,bs=32,ax=-1")
    w.observer_or_fake_quant_ctr = FusedAmaxObsFakeQuantize
    q = XNNPACKQuantizer().set_object_type(torch.ops.aten.linear.default, QuantizationConfig(None, None, w, None))
    torch.manual_seed(0)
    m = nn.Sequential(nn.Linear(64, 32)).eval()
    x = torch.randn(3, 64)
    gm = qp.prepare_pt2e(m, q, (x,))
    with torch.no_grad():
        y1 = gm(x)
    gc = qp.convert_pt2e(gm)
    targets = [str(n.target) for n in gc.graph.nodes]
    assert "quantized_ops.dequantize.default" in targets
    names = dict(gc.named_buffers())
    assert any(k.endswith("_uint4") for k in names) and any(k.endswith("_zero_point") for k in names)
    with torch.no_grad():
        y2 = gc(x)
    assert torch.allclose(y1, y2, atol=1e-5, rtol=1e-5)


PATTERNS = json.load(open(os.path.join(G, "pt2e_patterns.json")))


class ToyPatterns(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(16, 32)
        self.fc2 = nn.Linear(32, 16)
        self.ln = nn.LayerNorm(16)

    def forward(self, x):
        h = torch.nn.functional.gelu(self.fc1(x))
        y = self.fc2(torch.relu(h))
        y = y + x
        a = torch.matmul(y, y.transpose(-1, -2)) * 0.25
        return self.ln(torch.matmul(torch.softmax(a, -1), y))


def _pattern_quantizer(kind):
    from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize
    from quantized_training.quantizer.quantizer import QuantizationSpec
    from quantized_training.quantizer.xnnpack_quantizer import XNNPACKQuantizer
    from quantized_training.quantizer.xnnpack_quantizer_utils import QuantizationConfig

    def spec(s):
        q = QuantizationSpec.from_str(s)
        q.observer_or_fake_quant_ctr = FusedAmaxObsFakeQuantize.with_args(record_histogram=False, force_scale_power_of_two=False)
        return q
    cfg = QuantizationConfig(spec("int8,qs=per_tensor_symmetric"), None, spec("int8,qs=per_tensor_symmetric"), None)
    cfg_e = QuantizationConfig(spec("e4m3"), None, spec("e4m3"), None)
    if kind == "global":
        return XNNPACKQuantizer().set_global(cfg)
    if kind == "object_types":
        return (XNNPACKQuantizer().set_object_type(torch.ops.aten.softmax.int, cfg)
                .set_object_type(torch.ops.aten.layer_norm.default, cfg_e)
                .set_object_type(torch.ops.aten.gelu.default, cfg_e)
                .set_object_type(torch.ops.aten.relu.default, cfg))
    return XNNPACKQuantizer().set_module_name("fc1", cfg_e).set_global(cfg)


@pytest.mark.parametrize("kind", sorted(PATTERNS))
def test_remaining_static_patterns_match_reference(kind):
    """activation / softmax / layer_norm annotators (upstream xnnpack_quantizer_utils.py:371-503) through set_global,
    set_object_type and set_module_name: same prepared graph, same inserted modules and formats, same calibrated
    output and scales as the reference's quantizer on the same model."""
    arr = np.load(os.path.join(G, "pt2e_patterns.npz"))
    info = PATTERNS[kind]
    xs = [torch.from_numpy(arr[f"x{i}"].view(np.float32)).reshape(4, 8, 16) for i in range(3)]
    m = ToyPatterns().eval()
    r = np.random.default_rng(3)
    with torch.no_grad():
        for n, p in sorted(m.named_parameters()):
            p.copy_(torch.from_numpy((r.standard_normal(tuple(p.shape)) * 0.3).astype(np.float32)))
    gm = qp.prepare_pt2e(m, _pattern_quantizer(kind), (xs[0],))
    assert _rows(gm) == info["prepared_graph"]
    fq = {n: mod.dtype for n, mod in gm.named_modules() if isinstance(mod, torch.ao.quantization.FakeQuantizeBase)}
    assert fq == info["fq_modules"]
    with torch.no_grad():
        gm(xs[0])
        gm(xs[1])
        y = gm(xs[2])
    assert np.array_equal(_canon32(y).reshape(-1), arr[f"{kind}__y_prepared"].reshape(-1))
    scales = {k: [float(t) for t in v.reshape(-1)] for k, v in gm.state_dict().items() if k.endswith(".scale")}
    assert scales == info["scales"]


# ---- outlier side path (filter_outlier / spmm_csr and their lowering) against the reference ----------------------
OUTLIER = json.load(open(os.path.join(G, "outlier.json")))


def _f32(arr, key, shape):
    return torch.from_numpy(arr[key].view(np.float32).copy()).reshape(shape)


def test_filter_outlier_and_spmm_csr_operators():
    """quantized_ops::filter_outlier / spmm_csr (upstream decomposed.py:450-566): same inliers, same CSR arrays (row-major
    entries, padded to int(numel * max_pct)), and bit-identical products -- plain, with block scales, transposed."""
    arr = np.load(os.path.join(G, "outlier.npz"))
    ops = torch.ops.quantized_ops
    x = _f32(arr, "op__x", (3, 5, 32))
    inl, data, idx, ptr = ops.filter_outlier(x, OUTLIER["op"]["threshold"], OUTLIER["op"]["max_pct"])
    assert np.array_equal(_canon32(inl).reshape(-1), arr["op__inlier"].reshape(-1))
    assert np.array_equal(_canon32(data), arr["op__data"].reshape(-1))
    assert idx.dtype == torch.int32 and ptr.dtype == torch.int32
    assert np.array_equal(idx.numpy().astype(np.int64), arr["op__indices"]) and np.array_equal(ptr.numpy().astype(np.int64), arr["op__indptr"])
    assert int(ptr[-1]) == OUTLIER["op"]["nnz"]
    w, ws = _f32(arr, "op__w", (24, 32)), _f32(arr, "op__ws", (24, 1))
    assert np.array_equal(_canon32(ops.spmm_csr(data, idx, ptr, w)).reshape(-1), arr["op__y_plain"].reshape(-1))
    assert np.array_equal(_canon32(ops.spmm_csr(data, idx, ptr, w, ws, None, 32)).reshape(-1), arr["op__y_scaled"].reshape(-1))
    assert np.array_equal(_canon32(ops.spmm_csr(data, idx, ptr, w.T.contiguous(), None, None, None, True)).reshape(-1),
                          arr["op__y_transposed"].reshape(-1))
    # more outliers than the padded arrays hold: truncated with a warning; the product then fails as upstream's loop does
    _, d2, i2, p2 = ops.filter_outlier(x, 0.5, 0.01)
    assert d2.numel() == int(x.numel() * 0.01) and int(p2[-1]) > d2.numel()
    with pytest.raises(IndexError):
        ops.spmm_csr(d2, i2, p2, w)


def test_outlier_spec_lowering_matches_reference():
    """An activation spec with `outlier=`: convert_pt2e inserts filter_outlier in front of quantize_mx and adds
    spmm_csr(outliers, quantized weight) to the block-scaled linear (upstream quantize_pt2e.py:489-510, 705-750):
    identical converted graph, kwargs and outputs."""
    arr = np.load(os.path.join(G, "outlier.npz"))
    info = OUTLIER["model"]

    class Two(nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = nn.Linear(64, 128)
            self.fc2 = nn.Linear(128, 64)

        def forward(self, x):
            return self.fc2(torch.relu(self.fc1(x)))

    m = Two().eval()
    r = np.random.default_rng(3)
    with torch.no_grad():
        for n, p in sorted(m.named_parameters()):
            p.copy_(torch.from_numpy((r.standard_normal(tuple(p.shape)) * 0.1).astype(np.float32)))
    x0, x1 = _f32(arr, "x0", (16, 64)), _f32(arr, "x1", (16, 64))
    gm = qp.prepare_pt2e(m, qp.get_default_quantizer(**info["kw"]), (x0,))
    with torch.no_grad():
        gm(x0)
        y1 = gm(x1)
    assert np.array_equal(_canon32(y1).reshape(-1), arr["y_prepared"].reshape(-1))
    gc = qp.convert_pt2e(gm)
    assert _rows(gc) == info["converted_graph"]
    kwargs = {n.name: {k: (v.name if isinstance(v, torch.fx.Node) else v) for k, v in n.kwargs.items()}
              for n in gc.graph.nodes if n.kwargs}
    assert kwargs == info["converted_kwargs"]
    with torch.no_grad():
        y2 = gc(x1)
    assert np.array_equal(_canon32(y2).reshape(-1), arr["y_converted"].reshape(-1))
    assert float((y1 - y2).abs().max()) <= 1e-5 * float(y1.abs().max())      # the side path restores what the fake-quantizer kept


# ---- fusion pass for prepared graphs (pt2e_fusion.py): the rewrite changes launches, never values ------------------------------
def _tiny_llama_prepared(spec, layers=2, kv_heads=2):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=layers, num_attention_heads=2, num_key_value_heads=kv_heads,
                      vocab_size=512, max_position_embeddings=256, attn_implementation="eager")
    torch.manual_seed(0)
    model = LlamaForCausalLM(cfg).bfloat16().eval()
    ids = torch.randint(0, 512, (1, 64))
    q = qp.get_default_quantizer(spec, None, spec, "int24" if "qs=" in spec else None)
    q.set_module_name_object_type_order(r"model\.rotary_emb", torch.ops.aten.matmul.default, 0, None)      # wikitext.py:76-78
    seq = torch.export.Dim("seq_length", min=3, max=64)
    dyn = {"input_ids": {1: seq}, "labels": {1: seq}, "use_cache": None}
    with torch.no_grad():
        gm = qp.prepare_pt2e(model, q, (ids,), {"labels": ids.clone(), "use_cache": False}, dyn)
    return gm, ids


@pytest.mark.parametrize("spec", ["e4m3", "posit8_1", "int8,qs=per_tensor_symmetric"])
def test_fused_prepared_graph_is_the_prepared_graph(spec):
    """upstream's wikitext.py flow (:60-136) on a LLaMA-shaped model: every chain pt2e_fusion recognises (15 Linears, 2 q/k/v groups,
    2 MLPs, 5 RMSNorms -- 4 with the residual add in front --, 2 rotary + attention cores) is rewritten, and on CPU tensors, where each
    fused module runs the node sequence it replaced, logits, loss and the fake-quantized element count are those of the plain graph bit
    for bit -- other sequence lengths, live observers (int8 with a scale) and a later convert_pt2e included."""
    from quantized_training import fake_quantize as fqm, pt2e_fusion
    gm, ids = _tiny_llama_prepared(spec)
    plain_nodes = len(list(gm.graph.nodes))
    with torch.no_grad():
        fqm.STATS.reset()
        want = gm(ids, labels=ids.clone(), use_cache=False)
        e_plain = fqm.STATS.elements
        if "qs=" in spec:                                   # delayed scaling: the reference call above changed the observers' state
            gm, ids = _tiny_llama_prepared(spec)
        counts = pt2e_fusion.fuse_prepared_graph(gm)
        assert {k: counts[k] for k in ("linear", "sibling_groups", "mlp", "rmsnorm", "add_rmsnorm", "attention", "loss")} == \
            {"linear": 15, "sibling_groups": 2, "mlp": 2, "rmsnorm": 1, "add_rmsnorm": 4, "attention": 2, "loss": 1}
        assert counts["shape_only_nodes"] >= 40            # causal mask + rotary tables: functions of the sequence length alone
        assert len(list(gm.graph.nodes)) < plain_nodes // 2
        assert pt2e_fusion.fuse_prepared_graph(gm) == {}    # idempotent
        fqm.STATS.reset()
        got = gm(ids, labels=ids.clone(), use_cache=False)
        assert fqm.STATS.elements == e_plain
        assert torch.equal(got.logits, want.logits) and torch.equal(got.loss, want.loss)
        short = torch.randint(0, 512, (1, 40))
        a = gm(short, labels=short.clone(), use_cache=False)
        assert pt2e_fusion.unfuse_prepared_graph(gm) and len(list(gm.graph.nodes)) == plain_nodes
        assert not [n for n, _ in gm.named_children() if n.startswith("_qt_")]
        b = gm(short, labels=short.clone(), use_cache=False)
        if "qs=" not in spec:
            assert torch.equal(a.logits, b.logits)
        pt2e_fusion.fuse_prepared_graph(gm)
        gc = qp.convert_pt2e(gm)                            # restores the plain graph first, then lowers it as upstream does
        assert not any(n.op == "call_module" and str(n.target).startswith("_qt_") for n in gc.graph.nodes)
        assert any("quantized_ops" in str(n.target) for n in gc.graph.nodes)
        gc(short, labels=short.clone(), use_cache=False)


def test_fusion_leaves_unrecognised_chains_alone():
    """The toy model of the golden traces has no chain the pass knows beyond its two Linears: they become QAT-Linear nodes (same values),
    everything else stays node for node."""
    from quantized_training import pt2e_fusion
    arr = np.load(os.path.join(G, "pt2e.npz"))
    kw = META["e4m3_noqs"]["kw"]
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((4, 16)).astype(np.float32))
    g1 = qp.prepare_pt2e(_model(arr), qp.get_default_quantizer(**kw), (x,))
    g2 = qp.prepare_pt2e(_model(arr), qp.get_default_quantizer(**kw), (x,))
    counts = pt2e_fusion.fuse_prepared_graph(g2)
    assert counts["linear"] == 2 and counts["attention"] == 0 and counts["mlp"] == 0
    for _ in range(3):                                      # observers evolve identically
        assert torch.equal(g1(x), g2(x))


def test_fusion_on_a_grouped_query_model_keeps_what_it_does_not_recognise():
    """Grouped-query attention (one key / value head for two query heads) puts a repeat_kv chain between the rotary embedding and the
    matmuls: the attention pattern does not match and stays node for node, everything else is rewritten, and the graph still computes
    bit for bit what the plain one does."""
    from quantized_training import pt2e_fusion
    gm, ids = _tiny_llama_prepared("e4m3", kv_heads=1)
    with torch.no_grad():
        want = gm(ids, labels=ids.clone(), use_cache=False)
        counts = pt2e_fusion.fuse_prepared_graph(gm)
        assert counts["attention"] == 0 and counts["linear"] == 15 and counts["mlp"] == 2 and counts["add_rmsnorm"] == 4 and counts["loss"] == 1
        got = gm(ids, labels=ids.clone(), use_cache=False)
    assert torch.equal(got.logits, want.logits) and torch.equal(got.loss, want.loss)


def test_fusion_on_a_statically_shaped_export_with_the_rotary_matmul_quantized():
    """No dynamic sequence length (the shape-only sub-graph then has no size inputs and is computed once) and the rotary matmul left to
    the default annotator (its fake-quantizers are modules, so that chain is NOT shape-only and stays in the graph): still bit for bit."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from quantized_training import pt2e_fusion
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=2,
                      vocab_size=512, max_position_embeddings=256, attn_implementation="eager")
    torch.manual_seed(0)
    model = LlamaForCausalLM(cfg).bfloat16().eval()
    ids = torch.randint(0, 512, (1, 64))
    with torch.no_grad():
        gm = qp.prepare_pt2e(model, qp.get_default_quantizer("e4m3", None, "e4m3", None), (ids,), {"labels": ids.clone(), "use_cache": False})
        want = gm(ids, labels=ids.clone())                  # `use_cache` defaults to the example's value
        counts = pt2e_fusion.fuse_prepared_graph(gm)
        got = gm(ids, labels=ids.clone())
    assert counts["attention"] == 1 and counts["linear"] == 8 and counts["shape_only_nodes"] >= 20
    assert torch.equal(got.logits, want.logits) and torch.equal(got.loss, want.loss)


def test_fused_prepared_graph_keeps_the_prepared_graphs_state_dict():
    """A checkpoint of a fused prepared graph has exactly the keys of the plain prepared graph (upstream's format): the fused helper
    modules share Parameters and fake-quantizers with the graph and must not add `_qt_*` entries.  Strict loading works in both
    directions, and the loaded calibration state (amax history, scale) is what the fused modules then use."""
    from quantized_training import pt2e_fusion
    spec = "int8,qs=per_tensor_symmetric"
    plain, ids = _tiny_llama_prepared(spec)
    fused, _ = _tiny_llama_prepared(spec)
    with torch.no_grad():
        plain(ids, labels=ids.clone(), use_cache=False)          # sizes the lazily built buffers, moves the observers
        fused(ids, labels=ids.clone(), use_cache=False)
    keys_plain = list(plain.state_dict().keys())
    assert pt2e_fusion.fuse_prepared_graph(fused)["linear"] == 15
    sd_fused = fused.state_dict()
    assert list(sd_fused.keys()) == keys_plain and not [k for k in sd_fused if "_qt_" in k]
    # fused -> plain: perturb the fused graph's state, save, load strictly into the plain graph
    with torch.no_grad():
        for name, buf in fused.named_buffers():
            if name.endswith("scale") and "_qt_" not in name:
                buf.mul_(2.0)
    res = plain.load_state_dict(fused.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for (ka, va), (kb, vb) in zip(plain.state_dict().items(), fused.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    # plain -> fused: strict load of a plain checkpoint into the fused graph; the fused modules see it (shared objects)
    with torch.no_grad():
        for name, buf in plain.named_buffers():
            if name.endswith("scale"):
                buf.mul_(0.25)
    res = fused.load_state_dict(plain.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    short = torch.randint(0, 512, (1, 24))
    with torch.no_grad():
        a = fused(short, labels=short.clone(), use_cache=False)
        b = plain(short, labels=short.clone(), use_cache=False)
    assert torch.equal(a.logits, b.logits)
    # unfusing removes the hooks with the helpers
    assert pt2e_fusion.unfuse_prepared_graph(fused)
    assert list(fused.state_dict().keys()) == keys_plain and "_qt_state_hooks" not in fused.__dict__


def test_shape_memo_never_evicts_and_follows_its_constants():
    """ShapeMemo keeps what it computed (a captured hipGraph may replay reads of those tensors), stops keeping past its cap instead of
    evicting, and recomputes when a constant it reads is rewritten in place or moved."""
    from quantized_training.pt2e_fusion import ShapeMemo

    class Sub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.register_buffer("c", torch.ones(1), persistent=False)

        def forward(self, n):
            return (torch.arange(n) * self.c,)

    memo = ShapeMemo(Sub())
    first = memo(5)
    assert memo(5) is first
    for n in range(6, 6 + ShapeMemo.kMax + 8):
        memo(n)
    assert memo(5) is first and len(memo.__dict__["kept"]) == ShapeMemo.kMax       # nothing evicted; the surplus was not kept
    memo.sub.c.mul_(3.0)
    again = memo(5)
    assert again is not first and float(again[0][1]) == 3.0

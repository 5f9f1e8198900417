"""Host logic of the drop-in surface on CPU tensors: spec mini-language, CLI flags, QConfig, the
fake-quant module's state machine and state-dict, and quantize(model, args) (hook names, lazily
created fake-quantizers, outputs) -- all against golden data produced by running the reference."""
import json
import os
from dataclasses import asdict

import numpy as np
import pytest
import torch
import torch.nn as nn

import quantized_training as qt
from quantized_training.modules.quantizable import AddFunctional, MatmulFunctional, MulFunctional

G = os.path.join(os.path.dirname(__file__), "golden")
SPEC = json.load(open(os.path.join(G, "spec.json")))


def _canon16(b):
    b = b.copy()
    b[((b & 0x7F80) == 0x7F80) & ((b & 0x7F) != 0)] = 0x7FC0
    return b


def _canon32(b):
    b = b.copy()
    b[((b & 0x7F800000) == 0x7F800000) & ((b & 0x7FFFFF) != 0)] = 0x7FC00000
    return b


def _bits(t):
    if t.dtype == torch.bfloat16:
        return _canon16(t.contiguous().view(torch.int16).numpy().view(np.uint16))
    return _canon32(t.float().contiguous().view(torch.int32).numpy().view(np.uint32))


@pytest.mark.parametrize("s", sorted(SPEC["specs"]))
def test_spec_from_str(s):
    d = asdict(qt.QuantizationSpec.from_str(s))
    d.pop("observer_or_fake_quant_ctr")
    if d["qscheme"] is not None:
        d["qscheme"] = d["qscheme"].value
    for k in ("ch_axis", "block_size"):
        if isinstance(d[k], tuple):
            d[k] = list(d[k])
    assert d == SPEC["specs"][s]


@pytest.mark.parametrize("s", sorted(SPEC["errors"]))
def test_spec_errors(s):
    exp = SPEC["errors"][s]
    if exp is None:
        qt.QuantizationSpec.from_str(s)
    else:
        with pytest.raises(Exception) as ei:
            qt.QuantizationSpec.from_str(s)
        assert type(ei.value).__name__ == exp


def test_spec_accepts_parsed_object():
    spec = qt.QuantizationSpec.from_str("fp8_e5m2,qs=per_tensor_symmetric,qmax=57344,ahl=10")
    assert qt.QuantizationSpec.from_str(spec) is spec
    cfg = qt.get_qconfig("int8,qs=per_tensor_symmetric", None, spec)      # the reference raises TypeError here
    fq = cfg.error()
    assert fq.dtype == "fp8_e5m2" and fq.amax_history_len == 10 and fq.quant_max == 57344.0
    assert cfg.weight is nn.Identity


def test_quant_min_max():
    for dt, (lo, hi) in SPEC["quant_min_max"].items():
        got = qt.get_quant_min_max(dt)
        assert [float(got[0]), float(got[1])] == [lo, hi], dt
    with pytest.raises(ValueError):
        qt.get_quant_min_max("bogus")


def test_cli_flags_and_defaults():
    p = qt.add_qspec_args()
    flags = sorted(o for a in p._actions for o in a.option_strings)
    assert flags == SPEC["flags"]
    got = {k: (v if isinstance(v, (int, float, str, bool, type(None), list)) else repr(v))
           for k, v in vars(p.parse_args([])).items()}
    assert got == SPEC["arg_defaults"]
    a = p.parse_args(["--activation", "e4m3", "--op_fusion", "a,b", "--error", "fp8_e5m2,qs=per_tensor_symmetric",
                      "--bf16", "--quantize_forward", "gemm,residual"])
    assert a.op_fusion == ["a", "b"] and a.bf16 and isinstance(a.error, qt.QuantizationSpec)
    a = p.parse_args(["slurm", "--job-name", "x"])
    assert a.action == "slurm"


def test_package_exports():
    for name in ["quantize", "prepare", "convert", "propagate_config", "add_qspec_args", "get_qconfig", "QConfig",
                 "QuantizationSpec", "FusedAmaxObsFakeQuantize", "quantize_to_fp8_e4m3", "quantize_to_fp8_e5m2",
                 "quantize_to_posit", "setup_logging", "per_tensor_symmetric", "per_channel_symmetric",
                 "microscaling", "group_wise_affine"]:
        assert hasattr(qt, name), name
    for op in ["vmap", "quantize", "dequantize", "linear", "matmul"]:
        assert hasattr(torch.ops.quantized_ops, op)


def test_exported_rounding_functions_cpu():
    d = np.load(os.path.join(G, "direct_fns.npz"))
    x = torch.from_numpy(d["x"].view(np.float32))
    assert np.array_equal(_bits(qt.quantize_to_fp8_e4m3(x)), d["e4m3"])
    assert np.array_equal(_bits(qt.quantize_to_fp8_e5m2(x)), d["e5m2"])
    assert np.array_equal(_bits(qt.quantize_to_posit(x, 8, 1)), d["posit8_1"])
    xb = torch.from_numpy(d["xb"].view(np.int16)).view(torch.bfloat16)
    assert np.array_equal(_bits(qt.quantize_to_fp8_e4m3(xb)), d["e4m3_b"])
    assert np.array_equal(_bits(qt.quantize_to_posit(xb, 8, 1)), d["posit8_1_b"])


def test_ops_cpu_tensors():
    d = np.load(os.path.join(G, "vmap.npz"))
    for dt in ["int8", "e4m3", "posit8_1", "fp8_e4m3"]:
        qmap = qt.get_quantization_map(dt)
        x = torch.from_numpy(d["x32"].view(np.float32))
        assert np.array_equal(_bits(torch.ops.quantized_ops.vmap(x, qmap)), d[f"y32_{dt}"])
        xb = torch.from_numpy(d["xb"].view(np.int16)).view(torch.bfloat16)
        assert np.array_equal(_bits(qt.vmap(xb, qmap)), d[f"yb_{dt}"])
    with pytest.raises(ValueError):
        qt.get_quantization_map("bogus")
    with pytest.raises(AssertionError):
        torch.ops.quantized_ops.quantize(torch.ones(3), torch.ones(1))


FQ_META = json.load(open(os.path.join(G, "fake_quant.json")))


@pytest.mark.parametrize("case", FQ_META, ids=[c["name"] for c in FQ_META])
def test_module_traces_cpu(case):
    d = np.load(os.path.join(G, "fake_quant.npz"))
    kw = asdict(qt.QuantizationSpec.from_str(case["spec"]))
    m = qt.FusedAmaxObsFakeQuantize(**kw, force_scale_power_of_two=case["pow2"])
    assert sorted(n for n, _ in m.named_buffers()) == case["buffers"]
    bf16 = case["in"] == "bf16"
    for ci in range(case["n_calls"]):
        k = f"{case['name']}__{ci}__"
        if bf16:
            x = torch.from_numpy(d[k + "x"].view(np.int16)).view(torch.bfloat16).reshape(case["shape"])
        else:
            x = torch.from_numpy(d[k + "x"].view(np.float32)).reshape(case["shape"])
        y = m(x)
        assert np.array_equal(_bits(y).reshape(-1), d[k + "y"].reshape(-1)), ci
        assert np.array_equal(m.scale.reshape(-1).view(torch.int32).numpy().view(np.uint32), d[k + "scale"])
        if m._observe:
            assert np.array_equal(_canon32(m.amax_history.reshape(-1).view(torch.int32).numpy().view(np.uint32)), d[k + "hist"])
    sd = m.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == case["state_dict"]
    # a fresh module loads the state (0-sized buffers are resized) and continues identically
    m2 = qt.FusedAmaxObsFakeQuantize(**kw, force_scale_power_of_two=case["pow2"])
    m2.load_state_dict(sd)
    assert torch.equal(m2.scale, m.scale) and m2._observe == m._observe
    probe = torch.ones(case["shape"])
    assert torch.equal(torch.nan_to_num(m2(probe)), torch.nan_to_num(m(probe)))


def test_enable_disable_flags_and_ste():
    m = qt.FusedAmaxObsFakeQuantize("int4", qscheme=qt.per_tensor_symmetric, quant_max=7.0, amax_history_len=2)
    x = torch.tensor([0.3, -2.6, 9.0], requires_grad=True)
    y = m(x)
    y.sum().backward()
    assert torch.equal(x.grad, torch.ones(3))             # straight-through
    m.disable_observer()
    assert int(m.observer_enabled[0]) == 0 and not m._observe
    h = m.amax_history.clone()
    m(x.detach() * 100)
    assert torch.equal(m.amax_history, h)
    m.disable_fake_quant()
    z = x.detach()
    assert m(z) is z
    m.apply(torch.ao.quantization.enable_fake_quant)
    m.apply(torch.ao.quantization.enable_observer)
    assert m._quantize and m._observe
    assert "dtype=int4" in repr(m)


# ---- quantize(model, args) on the toy model of tests/golden/gen_golden.py --------------------------
class Block(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.query = nn.Linear(d, d)
        self.key = nn.Linear(d, d)
        self.value = nn.Linear(d, d)
        self.qk_matmul = MatmulFunctional()
        self.attn_scaling = MulFunctional()
        self.softmax = nn.Softmax(dim=-1)
        self.av_matmul = MatmulFunctional()
        self.dense = nn.Linear(d, d)
        self.residual = AddFunctional()
        self.LayerNorm = nn.LayerNorm(d)
        self.act = nn.GELU()

    def forward(self, x):
        q, k, v = self.query(x), self.key(x), self.value(x)
        s = self.qk_matmul(q, k.transpose(-1, -2))
        s = self.attn_scaling(s, 0.25)
        p = self.softmax(s)
        c = self.av_matmul(p, v)
        h = self.act(self.dense(c))
        return self.LayerNorm(self.residual(h, x))


class Toy(nn.Module):
    def __init__(self, d=16):
        super().__init__()
        self.layer0 = Block(d)
        self.layer1 = Block(d)
        self.head = nn.Linear(d, 4)

    def forward(self, x):
        return self.head(self.layer1(self.layer0(x)))


EAGER = json.load(open(os.path.join(G, "eager_trace.json")))


def _toy_from_golden(arrays):
    m = Toy()
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(torch.from_numpy(arrays["param__" + n].view(np.float32)).reshape(p.shape))
    return m


@pytest.mark.parametrize("run", sorted(EAGER), ids=sorted(EAGER))
def test_quantize_toy_model(run):
    arrays = np.load(os.path.join(G, "eager_trace.npz"))
    info = EAGER[run]
    m = _toy_from_golden(arrays)
    args = qt.add_qspec_args().parse_args([])
    for k, v in info["args"].items():
        setattr(args, k, v)
    qt.quantize(m, args)
    x = torch.from_numpy(arrays["x"].view(np.float32)).reshape(3, 5, 16)
    if info["args"].get("bf16"):
        x = x.bfloat16()
    if "losses" not in info:
        m.eval()
        with torch.no_grad():
            for i in range(info["n_fwd"]):
                y = m(x * (1.0 + i))
                assert np.array_equal(_bits(y).reshape(-1), arrays[f"{run}__y{i}"].reshape(-1)), (run, i)
    else:
        m.train()
        opt = torch.optim.SGD(m.parameters(), lr=0.05)
        for i in range(3):
            xi = (x * (1.0 + 0.5 * i)).requires_grad_(True)
            y = m(xi)
            loss = (y.float() ** 2).mean()
            opt.zero_grad()
            loss.backward()
            assert np.array_equal(_bits(y.detach()).reshape(-1), arrays[f"{run}__y{i}"].reshape(-1)), i
            assert np.array_equal(_bits(xi.grad).reshape(-1), arrays[f"{run}__gx{i}"].reshape(-1)), i
            assert np.array_equal(_bits(m.layer0.query.weight.grad).reshape(-1), arrays[f"{run}__gw{i}"].reshape(-1)), i
            opt.step()
            assert abs(float(loss) - info["losses"][i]) <= 1e-6 * abs(info["losses"][i])
    sd = m.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == info["state_dict"]
    assert [(n, type(mod).__name__) for n, mod in m.named_modules()] == [tuple(e) for e in info["modules"]]
    for k, v in sd.items():
        if k.endswith(".scale") or k.endswith(".amax_history"):
            exp = arrays[f"{run}__sd__{k}"]
            assert np.array_equal(_canon32(v.float().reshape(-1).view(torch.int32).numpy().view(np.uint32)), exp), k


def test_convert_shares_parameters_and_prepare_is_idempotent_on_names():
    m = Toy()
    w = m.layer0.query.weight
    args = qt.add_qspec_args().parse_args(["--activation", "int8,qs=per_tensor_symmetric", "--weight", "e4m3"])
    qt.quantize(m, args)
    assert m.layer0.query.weight is w
    assert type(m.layer0.query).__name__ == "Linear" and hasattr(m.layer0.query, "weight_fake_quant")
    assert isinstance(m.layer0.query.activation_pre_process, nn.ModuleDict)
    assert len(m.layer0.query.activation_pre_process) == 0          # created lazily on the first call
    m(torch.randn(2, 3, 16))
    assert list(m.layer0.query.activation_pre_process.keys()) == ["0"]
    assert list(m.layer0.qk_matmul.activation_pre_process.keys()) == ["0", "1"]
    assert m.layer0.query.activation_pre_process["0"].name == "layer0.query.0"
    f = m.layer0.query.to_float()
    assert isinstance(f, nn.Linear) and torch.equal(f.weight, w)
    with pytest.raises(AssertionError):
        qt.prepare(Toy(), True, "bogus_op", None)


MX_META = json.load(open(os.path.join(G, "mx.json")))


@pytest.mark.parametrize("case", MX_META, ids=[c["name"] for c in MX_META])
def test_block_scaled_module_cpu(case):
    """qs=microscaling / qs=group_wise_affine through the module on CPU tensors vs the reference's outputs."""
    d = np.load(os.path.join(G, "mx.npz"))
    n = case["name"]
    kw = asdict(qt.QuantizationSpec.from_str(case["spec"]))
    m = qt.FusedAmaxObsFakeQuantize(**kw, force_scale_power_of_two=case["pow2"])
    if case["in"] == "bf16":
        x = torch.from_numpy(d[n + "__x"].view(np.int16)).view(torch.bfloat16).reshape(case["shape"])
    else:
        x = torch.from_numpy(d[n + "__x"].view(np.float32)).reshape(case["shape"])
    y = m(x)
    assert np.array_equal(_bits(y).reshape(-1), d[n + "__y"].reshape(-1))
    assert list(m.scale.shape) == case["scale_shape"]
    assert np.array_equal(_canon32(m.scale.float().reshape(-1).view(torch.int32).numpy().view(np.uint32)), d[n + "__scale"])
    if "group_wise" in case["spec"]:
        assert np.array_equal(_canon32(m.zero_point.float().reshape(-1).view(torch.int32).numpy().view(np.uint32)), d[n + "__zp"])


@pytest.mark.parametrize("dt", ["nf4", "nf4_6", "nf3", "nf2_4"])
def test_normal_float_maps(dt):
    """NormalFloat code books: (indices, values) like the reference, flattened table equal to its values[indices]."""
    g = np.load(os.path.join(G, "maps.npz"))
    idx, vals = qt.get_quantization_map(dt)
    assert np.array_equal(_canon16(vals[idx].view(torch.int16).numpy().view(np.uint16)), g[dt])
    assert np.array_equal(idx.numpy().astype(np.uint16), g[dt + "__indices"])
    m = qt.FusedAmaxObsFakeQuantize(dt)
    assert np.array_equal(_canon16(m.qmap.view(torch.int16).numpy().view(np.uint16)), g[dt])
    i2, v2 = qt.quantize_to_nf(torch.tensor([0.3, -2.0, 0.0], dtype=torch.bfloat16), 4)
    assert v2.numel() == 16 and float(v2[i2[1]]) == -1.0 and float(v2[i2[2]]) == 0.0


# ---------------------------------------------------------------------------------------------------------------
# LoRA QAT layer (upstream modules/qat/lora.py) against three reference training steps
LORA = json.load(open(os.path.join(G, "lora.json")))
LORA_NPZ = np.load(os.path.join(G, "lora.npz"))


def _from_bits16(a, shape):
    return torch.from_numpy(a.astype(np.uint16).view(np.int16).copy()).view(torch.bfloat16).reshape(shape)


class _PeftStyleLoraLinear(nn.Module):
    """The attributes peft's LoRA linear layer exposes (current layout: frozen projection under base_layer)."""

    def __init__(self, fin, fout, r, fan_in_fan_out):
        super().__init__()
        self.base_layer = nn.Linear(fin, fout).to(torch.bfloat16)
        if fan_in_fan_out:
            self.base_layer.weight = nn.Parameter(torch.empty(fin, fout, dtype=torch.bfloat16))
        self.base_layer.weight.requires_grad_(False)
        self.base_layer.bias.requires_grad_(False)
        self.in_features, self.out_features = fin, fout
        self.lora_A = nn.ModuleDict({"default": nn.Linear(fin, r, bias=False).to(torch.bfloat16)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(r, fout, bias=False).to(torch.bfloat16)})
        self.scaling = {"default": 2.0}
        self.r, self.lora_alpha = {"default": r}, {"default": 2 * r}
        self.active_adapters = ["default"]
        self.merged_adapters = []
        self.disable_adapters = False
        self.fan_in_fan_out = fan_in_fan_out


def _close_bits(got, want_bits, shape, rel):
    """A device result against the reference's bf16 bits: within `rel` of the largest magnitude (the device's GEMMs add in another order
    than the CPU's)."""
    want = _from_bits16(want_bits, shape).float()
    return bool(((got.detach().float().cpu() - want).abs() <= rel * float(want.abs().max()) + 1e-30).all())


@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("case", LORA, ids=[c["name"] for c in LORA])
def test_lora_qat_linear_matches_reference_steps(case, device):
    """modules/qat/lora.py:34-55 upstream, against steps recorded from the reference (tests/golden/gen_golden.py): on CPU tensors bit for
    bit; on the device (the three weight fake-quantizer calls per forward run as HIP launches, the products as device GEMMs) the
    delayed-scaling scale bit for bit or one bf16 step of the merged weight's amax away, outputs and adapter gradients within 2^-6 of
    their largest magnitude (another summation order in the rank-r product and the projection)."""
    from quantized_training.modules.qat import LoraLinear
    n, fin, fout, r = case["name"], case["in"], case["out"], case["r"]
    key = lambda k: LORA_NPZ[(n + "/" + k).replace("/", "__")]
    flt = _PeftStyleLoraLinear(fin, fout, r, case["fan_in_fan_out"])
    if device != "cpu":
        _lora_on_device(case, flt, key, device)
        return
    with torch.no_grad():
        flt.base_layer.weight.copy_(_from_bits16(key("w"), flt.base_layer.weight.shape))
        flt.base_layer.bias.copy_(_from_bits16(key("b"), (fout,)))
        flt.lora_A["default"].weight.copy_(_from_bits16(key("A"), (r, fin)))
        flt.lora_B["default"].weight.copy_(_from_bits16(key("B"), (fout, r)))
    flt.qconfig = qt.get_qconfig(None, qt.QuantizationSpec.from_str(case["spec"]), None)
    layer = LoraLinear.from_float(flt)
    assert layer.weight is flt.base_layer.weight and layer.lora_A is flt.lora_A
    a, b = flt.lora_A["default"].weight, flt.lora_B["default"].weight
    opt = torch.optim.SGD([a, b], lr=case["lr"])
    for step in range(case["steps"]):
        x = _from_bits16(key(f"{step}/x"), (5, fin))
        y = layer(x)
        opt.zero_grad()
        y.float().square().mean().backward()
        assert np.array_equal(_bits(y.detach()), key(f"{step}/y")), step
        assert np.array_equal(_bits(a.grad), key(f"{step}/gA")), step
        assert np.array_equal(_bits(b.grad), key(f"{step}/gB")), step
        assert np.array_equal(layer.weight_fake_quant.scale.detach().float().reshape(-1).view(torch.int32).numpy().view(np.uint32),
                              key(f"{step}/scale")), step
        opt.step()
    assert layer.weight.grad is None
    # adapters disabled -> the plain projection
    layer.enable_adapters(False)
    x = _from_bits16(key("0/x"), (5, fin))
    w = layer.weight.T if case["fan_in_fan_out"] else layer.weight
    assert torch.equal(layer(x), torch.nn.functional.linear(x, w, layer.bias))


def _lora_on_device(case, flt, key, device):
    from quantized_training.modules.qat import LoraLinear
    from quantized_training.fake_quantize import STATS
    fin, fout, r = case["in"], case["out"], case["r"]
    with torch.no_grad():
        flt.base_layer.weight.copy_(_from_bits16(key("w"), flt.base_layer.weight.shape))
        flt.base_layer.bias.copy_(_from_bits16(key("b"), (fout,)))
        flt.lora_A["default"].weight.copy_(_from_bits16(key("A"), (r, fin)))
        flt.lora_B["default"].weight.copy_(_from_bits16(key("B"), (fout, r)))
    flt = flt.to(device)
    flt.qconfig = qt.get_qconfig(None, qt.QuantizationSpec.from_str(case["spec"]), None)
    layer = LoraLinear.from_float(flt).to(device)
    a, b = flt.lora_A["default"].weight, flt.lora_B["default"].weight
    opt = torch.optim.SGD([a, b], lr=case["lr"])
    STATS.reset()
    for step in range(case["steps"]):
        x = _from_bits16(key(f"{step}/x"), (5, fin)).to(device)
        y = layer(x)
        opt.zero_grad()
        y.float().square().mean().backward()
        assert y.is_cuda and _close_bits(y, key(f"{step}/y"), (5, fout), 2.0 ** -6), step
        assert _close_bits(a.grad, key(f"{step}/gA"), (r, fin), 2.0 ** -5), step
        assert _close_bits(b.grad, key(f"{step}/gB"), (fout, r), 2.0 ** -5), step
        got = float(layer.weight_fake_quant.scale.detach().float().reshape(-1)[0])
        want = float(torch.from_numpy(key(f"{step}/scale").view(np.float32).copy()).reshape(-1)[0])
        assert abs(got - want) <= 2.0 ** -7 * abs(want), (step, got, want)
        opt.step()
    assert layer.weight.grad is None
    # three fake-quantizer calls per forward (A, B, the merged weight), all counted by the device path's statistics
    assert STATS.calls >= 3 * case["steps"], STATS.calls

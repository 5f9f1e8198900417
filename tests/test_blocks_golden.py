"""Upstream's quantizable BERT / MobileBERT twins, its SQuAD-loop logits, the histogram / outlier options and a calibrated
checkpoint -- all recorded by RUNNING the reference (tests/golden/gen_golden_blocks.py) -- against this repo's
`quantize()` on the same drivers (tests/golden/mini_models.py).

On the CPU the product's torch formulation reproduces the fixtures bit for bit (same operations in the same order).  On
the device (marker gpu) the elementwise fake-quant steps are bit-exact kernels but the GEMMs accumulate in a different
order (FP8 / bf16 matrix cores), and a value that lands on the other side of a rounding boundary moves by one step of the
format; the check there is per ROW of every tap: relative L2 error of each row within a bound that a dropped residual, a
wrong mask broadcast or a skipped fake-quantizer on any single row exceeds by an order of magnitude.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
sys.path.insert(0, G)
import mini_models as mm  # noqa: E402

import quantized_training as qt  # noqa: E402
from quantized_training import harness  # noqa: E402

BLOCKS_META = json.load(open(os.path.join(G, "blocks.json")))
QA_META = json.load(open(os.path.join(G, "qa_logits.json")))
FQX_META = json.load(open(os.path.join(G, "fq_extra.json")))
CKPT_META = json.load(open(os.path.join(G, "checkpoint.json")))
KINDS = {"bert": ("BertBlock", "tiny_bert_config", 64, 0.25), "mobilebert": ("MobileBertBlock", "tiny_mobilebert_config", 64, 0.25),
         "bert_hd64": ("BertBlock", "bert_hd64_config", 256, 0.08)}
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


@pytest.fixture(scope="module")
def blocks():
    return np.load(os.path.join(G, "blocks.npz"))


def arr(npz, key):
    return npz[key.replace("/", "__")]


def bits(t):
    t = t.detach().cpu()
    if t.dtype == torch.bfloat16:
        return t.contiguous().view(torch.int16).numpy().astype(np.uint16)
    return t.float().contiguous().view(torch.int32).numpy().astype(np.uint32)


def values(a):
    if a.dtype == np.uint16:
        return torch.from_numpy((a.astype(np.uint32) << 16).view(np.float32).copy())
    return torch.from_numpy(a.view(np.float32).copy())


def argv_of(kw, dtype):
    argv = []
    for k, v in kw.items():
        if v is not None:
            argv += ["--" + k, v]
    if dtype == "bfloat16":
        argv.append("--bf16")
    return qt.add_qspec_args().parse_args(argv)


def block_inputs(seed, dtype, H=64, B=2, S=24):
    r = np.random.default_rng(seed)
    h = torch.from_numpy(r.standard_normal((B, S, H)).astype(np.float32)).to(dtype)
    keep = torch.ones(B, S)
    keep[1, S - 5:] = 0
    return h, mm.additive_mask(keep, dtype)


def rows_close(got, exp, bound):
    """Every row (last axis) of `got` within `bound` relative L2 error of `exp` (rows of ~zero norm: absolute)."""
    g, e = got.detach().float().cpu().reshape(-1, got.shape[-1]), exp.float().reshape(-1, got.shape[-1])
    err = (g - e).norm(dim=1)
    ref = e.norm(dim=1).clamp_min(1e-3 * float(e.norm()) / max(1, e.shape[0]) ** 0.5 + 1e-12)
    worst = float((err / ref).max())
    return worst <= bound, worst


def build_block(kind, dtype, device):
    cls, cfg, hidden, std = KINDS[kind]
    blk = mm.seeded_init_(getattr(mm, cls)(getattr(mm, cfg)()), 11, std=std).eval()
    if dtype == "bfloat16":
        blk = blk.bfloat16()
    return blk.to(device), hidden


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("key", sorted(BLOCKS_META))
def test_blocks_match_upstream_twins(blocks, key, device):
    """modules/quantizable/modeling_bert.py:33-214, modeling_mobilebert.py:38-206 through upstream quantize(): same module
    tree, hook names and state-dict keys; same tensors at every tap, same delayed-scaling state after every forward."""
    kind, sname = key.split("/")
    info = BLOCKS_META[key]
    dtype = info["dtype"]
    td = getattr(torch, dtype)
    blk, hidden = build_block(kind, dtype, device)
    qt.quantize(blk, argv_of(info["args"], dtype))
    bound = 0.0 if device == "cpu" else (0.06 if td == torch.bfloat16 else 0.02)
    with torch.no_grad():
        for i in range(info["n_fwd"]):
            h, mask = block_inputs(100 + i, td, H=hidden)
            taps = {}
            blk((h * (1.0 + 0.5 * i)).to(device), mask.to(device), taps)
            for t, v in taps.items():
                exp = arr(blocks, f"{key}/fwd{i}/{t}")
                if device == "cpu":
                    assert np.array_equal(bits(v), exp), (key, i, t)
                else:
                    ok, worst = rows_close(v, values(exp).reshape(v.shape), bound)
                    assert ok, (key, i, t, worst)
            for k, v in blk.state_dict().items():
                if not (k.endswith(".scale") or k.endswith(".amax_history")):
                    continue
                exp = arr(blocks, f"{key}/fwd{i}/sd/{k}")
                if device == "cpu":
                    assert np.array_equal(bits(v.reshape(-1)), exp), (key, i, k)
                else:                                           # amax of a tensor that differs in the last bits of a few values
                    assert torch.allclose(v.reshape(-1).float().cpu(), values(exp), rtol=0.03, atol=1e-6), (key, i, k)
    assert sorted(n for n, m in blk.named_modules() if type(m).__name__ == "FusedAmaxObsFakeQuantize") == info["fake_quantizers"]
    assert {k: list(v.shape) for k, v in blk.state_dict().items()} == info["state_dict"]
    assert sorted((n, type(m).__name__) for n, m in blk.named_modules()) == sorted(map(tuple, info["modules"]))


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["bert_hd64/e4m3_act_weight", "bert_hd64/posit8_1_act_bf16", "bert/e4m3_act_weight", "mobilebert/e4m3_act_weight"])
def test_device_fake_quant_outputs_are_single_code_steps_from_the_cpu_run(key, monkeypatch):
    """what the device run's differences from the CPU run (which reproduces upstream bit for bit, test above) ARE.
    Every fake-quantizer's output is tapped on both runs of the same block: all values lie on the format's grid, and where the device
    value differs from the CPU value it is the NEIGHBOURING grid value -- an input that arrived one bf16 step off (another summation
    order in a GEMM, LayerNorm or softmax) and fell on the other side of a rounding boundary -- on a bounded share of the elements.
    The fused routes that never materialise a fake-quantized tensor are switched off for this comparison (they have their own
    kernel-level tests against the oracle); the plain route runs every fake-quantizer as a HIP pass."""
    from oracle import qt_oracle as o
    from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize
    for k in ("QT_FP8_GEMM", "QT_FUSED_SOFTMAX", "QT_FUSED_MODEL_OPS", "QT_FUSED_PRODUCER_FQ", "QT_FP8_ATTENTION", "QT_FUSED_ATTENTION",
              "QT_FP8_ATTENTION_KERNEL", "QT_FQT_GEMM"):
        monkeypatch.setenv(k, "0")
    kind, sname = key.split("/")
    info = BLOCKS_META[key]
    dtype = info["dtype"]
    td = getattr(torch, dtype)
    runs = {}
    for device in ("cpu", "cuda"):
        blk, hidden = build_block(kind, dtype, device)
        qt.quantize(blk, argv_of(info["args"], dtype))
        taps = {}

        def hook(name):
            def fn(mod, args, out):
                taps.setdefault(name, []).append((out[0] if isinstance(out, tuple) else out).detach().float().cpu())
            return fn
        with torch.no_grad():
            h, mask = block_inputs(100, td, H=hidden)
            blk(h.to(device), mask.to(device))                       # the first forward creates the per-input fake-quantizers
            hs = [m.register_forward_hook(hook(n)) for n, m in blk.named_modules() if isinstance(m, FusedAmaxObsFakeQuantize)]
            blk(h.to(device), mask.to(device))
            for hk in hs:
                hk.remove()
        runs[device] = taps
    assert runs["cpu"].keys() == runs["cuda"].keys() and len(runs["cpu"]) >= 6
    dt = info["args"]["activation"]
    qmap = o.get_quantization_map(dt)
    grid = np.unique(o.bf16_to_f32(qmap[np.isfinite(o.bf16_to_f32(qmap))]).astype(np.float64))
    worst_share = 0.0
    for name in runs["cpu"]:
        for a, b in zip(runs["cpu"][name], runs["cuda"][name]):
            a, b = a.numpy().astype(np.float64), b.numpy().astype(np.float64)
            assert np.isin(b, grid).all(), name
            steps = np.abs(np.searchsorted(grid, a) - np.searchsorted(grid, b))
            assert steps.max() <= 1, (name, int(steps.max()))
            worst_share = max(worst_share, float((steps > 0).mean()))
    assert worst_share <= 0.04, worst_share


def _qa_logits(sname, device):
    npz = np.load(os.path.join(G, "qa_logits.npz"))
    info = QA_META[sname]
    dtype = info["dtype"]
    model = mm.qa_model(info.get("model", "bert"))
    if dtype == "bfloat16":
        model = model.bfloat16()
    model = model.to(device)
    qt.quantize(model, argv_of(info["args"], dtype))
    batches = []
    for i in range(3):
        batches.append({"input_ids": torch.from_numpy(arr(npz, f"batch{i}/input_ids")),
                        "attention_mask": torch.from_numpy(arr(npz, f"batch{i}/attention_mask"))})
    start, end = harness.collect_qa_logits(model, batches, device=torch.device(device))
    return npz, info, model, start, end


# Routes of the device path for a bf16 FP8 model, from the one that repeats upstream's operations to the default one:
#   plain   - fake-quantized VALUES through bf16 GEMMs, Hugging Face's LayerNorm / GELU / softmax kernels
#   fused   - same GEMMs, one-launch LayerNorm / GELU / softmax (summation order of the statistics differs: 1 bf16 ulp)
#   default - FP8 codes through the FP8 matrix instruction as well (its 128-term dot product is not the CPU's fp32 chain)
# A last-bit difference in a hidden value flips an 8-bit code (a 6-12 % step) in a few per cent of the cases, and a logit is a
# 64-term dot product of such values: the plain route must reproduce upstream's logits, the other two are bounded
# statistically (and by what they do to the answer: the arg max).
QA_ROUTES = {
    "plain": {"QT_FP8_GEMM": "0", "QT_FUSED_MODEL_OPS": "0", "QT_FUSED_SOFTMAX": "0", "QT_FUSED_ATTENTION": "0"},
    "fused": {"QT_FP8_GEMM": "0"},
    "default": {},
}


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("sname", sorted(QA_META))
def test_qa_loop_logits_match_upstream(sname, device, monkeypatch):
    """The SQuAD evaluation loop (run_qa_no_trainer.py:914-959) on a seeded tiny BERT with upstream's twins: start / end
    logits over three padded batches, collected by harness.collect_qa_logits."""
    routes = QA_ROUTES if (device != "cpu" and QA_META[sname]["dtype"] == "bfloat16") else {"default": {}}
    for route, env in routes.items():
        for k in ("QT_FP8_GEMM", "QT_FUSED_MODEL_OPS", "QT_FUSED_SOFTMAX", "QT_FUSED_ATTENTION"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        npz, info, model, start, end = _qa_logits(sname, device)
        dtype = info["dtype"]
        for name, got in (("start_logits", start), ("end_logits", end)):
            exp = arr(npz, f"{sname}/{name}")
            if device == "cpu":
                assert np.array_equal(bits(got.float()), exp), (sname, name)
                continue
            e = values(exp).reshape(got.shape)
            g = got.float().cpu()
            d = (g - e).abs()
            scale = float(e.abs().max())
            rms, worst = float(d.pow(2).mean().sqrt()) / scale, float(d.max()) / scale
            share = float((d > 0.01 * scale).float().mean())
            tag = (sname, route, name, rms, worst, share)
            if dtype != "bfloat16" or route == "plain":
                # fp32 models (table formats) and the plain bf16 route: the same operations as upstream's, the logits
                # agree to accumulation order
                assert share <= 0.02 and rms <= 0.01 and worst <= 0.2, tag
            else:
                corr = float(torch.corrcoef(torch.stack([g.flatten(), e.flatten()]))[0, 1])
                # the answer: the position we would pick scores, in upstream's logits, within `regret` of upstream's pick
                # (random-init logits are flat -- ties within the noise are common, so positions themselves may differ)
                regret = float((e.max(-1).values - e.gather(-1, g.argmax(-1, keepdim=True)).squeeze(-1)).max()) / scale
                lim = {"fused": (0.03, 0.15, 0.995, 0.1), "default": (0.05, 0.2, 0.99, 0.15)}[route]
                assert rms <= lim[0] and worst <= lim[1] and corr >= lim[2] and regret <= lim[3], (corr, regret) + tag
        assert {k: list(v.shape) for k, v in model.state_dict().items()} == info["state_dict"]
        if "fake_quantizers" in info:                     # where upstream's quantize() put fake-quantizers, by module name
            assert sorted(n for n, m in model.named_modules() if type(m).__name__ == "FusedAmaxObsFakeQuantize") == info["fake_quantizers"]
        if device == "cpu":
            for k, v in model.state_dict().items():
                if k.endswith(".scale") or k.endswith(".amax_history"):
                    assert np.array_equal(bits(v.reshape(-1)), arr(npz, f"{sname}/sd/{k}")), k


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("case", FQX_META, ids=[c["name"] for c in FQX_META])
def test_histogram_and_outlier_options(case, device):
    """record_histogram / outlier_threshold (fake_quantize.py:348-359, 401-402): outputs, histogram buffer, scale and
    max_outlier_pct after every call -- bit for bit on both devices (every step is elementwise or an exact count)."""
    from dataclasses import asdict
    npz = np.load(os.path.join(G, "fq_extra.npz"))
    kw = asdict(qt.QuantizationSpec.from_str(case["spec"]))
    kw.update(case["extra"])
    m = qt.FusedAmaxObsFakeQuantize(**kw, device=device)
    if case["disable"]:
        m.disable_observer()
        m.disable_fake_quant()
    td = torch.bfloat16 if case["in"] == "bf16" else torch.float32
    for ci in range(case["n_calls"]):
        k = f"{case['name']}/{ci}"
        x = values(arr(npz, k + "/x")).reshape(case["shape"]).to(td).to(device)
        with torch.no_grad():
            y = m(x)
        exp = arr(npz, k + "/y")
        got = bits(y)
        nan16 = lambda b: np.where(((b & 0x7F80) == 0x7F80) & ((b & 0x7F) != 0), 0x7FC0, b)  # noqa: E731
        if td == torch.bfloat16:
            assert np.array_equal(nan16(got), exp), (k, "y")
        else:
            assert np.array_equal(got, exp), (k, "y")
        assert np.array_equal(bits(m.histogram), arr(npz, k + "/histogram")), (k, "histogram")
        assert np.array_equal(bits(m.scale.reshape(-1)), arr(npz, k + "/scale")), (k, "scale")
        assert float(getattr(m, "max_outlier_pct", -1.0)) == pytest.approx(case["calls"][ci]["max_outlier_pct"], abs=1e-12)
    assert sorted(n for n, _ in m.named_buffers()) == case["buffers"]
    assert sorted(m.state_dict().keys()) == case["state_dict"]


@pytest.mark.parametrize("device", DEVICES)
def test_upstream_checkpoint_loads_and_reproduces(device):
    """A state_dict written by upstream after two calibration forwards (amax_history / scale sized by the first observed
    call, fake_quantize.py:406-435) loads into a freshly converted block whose buffers are still empty; with the observers
    frozen (run_qa_no_trainer.py:834-847) the next forward equals upstream's."""
    npz = np.load(os.path.join(G, "checkpoint.npz"))
    blk, hidden = build_block("bert", "float32", device)
    qt.quantize(blk, argv_of(CKPT_META["args"], "float32"))
    sd = {}
    for k, m in CKPT_META["state_dict"].items():
        a = arr(npz, "sd/" + k)
        t = torch.from_numpy(a.copy()) if m["dtype"] in ("uint8", "int64") else values(a)
        sd[k] = t.reshape(m["shape"]).to(getattr(torch, m["dtype"]))
    # (1) upstream's flow (run_qa_no_trainer.py:826-832, 983-988): one forward registers the activation fake-quantizers (on
    # other data than the checkpoint saw), then the checkpoint is loaded over whatever state that left
    h0, mask0 = block_inputs(999, torch.float32, H=hidden)
    with torch.no_grad():
        blk(h0.to(device) * 3.0, mask0.to(device))
    missing, unexpected = blk.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    # (2) a fake-quantizer that has never observed anything holds 0-sized buffers (fake_quantize.py:406-435): they take
    # the checkpoint's shapes
    name = "attention.self.query.activation_pre_process.0."
    from dataclasses import asdict
    fresh = qt.FusedAmaxObsFakeQuantize(**asdict(qt.QuantizationSpec.from_str(CKPT_META["args"]["activation"])), device=device)
    assert fresh.amax_history.numel() == 0
    fresh.load_state_dict({k[len(name):]: v for k, v in sd.items() if k.startswith(name)}, strict=True)
    assert list(fresh.amax_history.shape) == CKPT_META["state_dict"][name + "amax_history"]["shape"]
    assert np.array_equal(bits(fresh.scale.reshape(-1)), arr(npz, "sd/" + name + "scale").reshape(-1))
    harness.freeze_observers(blk)
    h, mask = block_inputs(CKPT_META["eval_seed"], torch.float32, H=hidden)
    taps = {}
    with torch.no_grad():
        blk(h.to(device), mask.to(device), taps)
    for t, v in taps.items():
        exp = arr(npz, "after/" + t)
        if device == "cpu":
            assert np.array_equal(bits(v), exp), t
        else:
            ok, worst = rows_close(v, values(exp).reshape(v.shape), 0.02)
            assert ok, (t, worst)
    for k, v in blk.state_dict().items():                       # frozen: the loaded state is still there
        if k.endswith(".scale") or k.endswith(".amax_history"):
            assert np.array_equal(bits(v.reshape(-1)), arr(npz, "sd/" + k).reshape(-1)), k


# ---- upstream's quantize() toy trace (tests/golden/eager_trace.*, pinned bit for bit on the CPU in test_surface_cpu.py) on the device
@pytest.mark.gpu
@pytest.mark.parametrize("run", sorted(json.load(open(os.path.join(G, "eager_trace.json")))))
def test_quantize_toy_model_on_device(run):
    """The reference's own quantize() traces -- evaluation forwards of five specs and three SGD steps with int8 forward and
    E5M2 backward hooks (quantize.py:52-193) -- executed by the HIP path: per-row bounds on outputs and gradients, losses
    within 1 %, identical module tree and state-dict keys, delayed-scaling state within the same bound."""
    from test_surface_cpu import EAGER, _toy_from_golden
    arrays = np.load(os.path.join(G, "eager_trace.npz"))
    info = EAGER[run]
    m = _toy_from_golden(arrays).cuda()
    args = qt.add_qspec_args().parse_args([])
    for k, v in info["args"].items():
        setattr(args, k, v)
    qt.quantize(m, args)
    x = torch.from_numpy(arrays["x"].view(np.float32)).reshape(3, 5, 16).cuda()
    bf16 = bool(info["args"].get("bf16"))
    if bf16:
        x = x.bfloat16()
    bound = 0.08 if bf16 else 0.03

    def exp_of(key, like):
        return values(arrays[key]).reshape(like.shape)

    if "losses" not in info:
        m.eval()
        with torch.no_grad():
            for i in range(info["n_fwd"]):
                y = m(x * (1.0 + i))
                ok, worst = rows_close(y, exp_of(f"{run}__y{i}", y), bound)
                assert ok, (run, i, worst)
    else:
        m.train()
        opt = torch.optim.SGD(m.parameters(), lr=0.05)
        for i in range(3):
            xi = (x * (1.0 + 0.5 * i)).requires_grad_(True)
            y = m(xi)
            loss = (y.float() ** 2).mean()
            opt.zero_grad()
            loss.backward()
            for key, got in ((f"{run}__y{i}", y.detach()), (f"{run}__gx{i}", xi.grad), (f"{run}__gw{i}", m.layer0.query.weight.grad)):
                ok, worst = rows_close(got, exp_of(key, got), 0.12)       # E5M2 gradients: 2 mantissa bits, a flipped code is 25 %
                assert ok, (run, i, key, worst)
            opt.step()
            assert abs(float(loss) - info["losses"][i]) <= 1e-2 * abs(info["losses"][i])
    sd = m.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == info["state_dict"]
    assert [(n, type(mod).__name__) for n, mod in m.named_modules()] == [tuple(e) for e in info["modules"]]
    for k, v in sd.items():
        if k.endswith(".scale") or k.endswith(".amax_history"):
            exp = values(arrays[f"{run}__sd__{k}"])
            assert torch.allclose(torch.nan_to_num(v.float().reshape(-1).cpu()), torch.nan_to_num(exp), rtol=0.05, atol=1e-6), k


# ---- quantize_to_posit(..., round_to_even=False) / (..., return_pbits=True)                     posit.py:27-35, 50-53, 60-65
@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("nbits,es", [(8, 0), (8, 1), (8, 2), (16, 1), (6, 1), (16, 2)])
def test_posit_options_match_upstream(nbits, es, device):
    """Values: bit for bit on every input, in both rounding modes.  Bit patterns: equal wherever upstream's int32 arithmetic
    defines them -- finite, non-zero, not regime-dominated, pattern length 2 + run + es + 23 <= 33 (beyond that its
    `regime << (23 + es)` overflows; regime-dominated inputs are shifted by platform-dependent counts) -- and saturated
    (maxpos / minpos / 0) where it does not."""
    d = np.load(os.path.join(G, "posit_opts.npz"))
    xb = d["x"]
    x = torch.from_numpy(xb.view(np.float32).copy()).to(device)
    y0 = qt.quantize_to_posit(x, nbits, es, round_to_even=False)
    y1, pb = qt.quantize_to_posit(x, nbits, es, return_pbits=True)
    assert pb.dtype == torch.int32 and pb.shape == x.shape

    def canon(t):
        b = bits(t).copy()
        b[((b & 0x7F800000) == 0x7F800000) & ((b & 0x7FFFFF) != 0)] = 0x7FC00000
        return b
    assert np.array_equal(canon(y0), d[f"p{nbits}_{es}__y_no_rte"])
    assert np.array_equal(canon(y1), d[f"p{nbits}_{es}__y"])
    a = xb & 0x7FFFFFFF
    scale = (a >> 23).astype(np.int64) - 127
    k = scale >> es
    run = np.where(scale >= 0, 1 + k, -k)
    dominated = np.abs(scale) > ((nbits - 2) << es)
    comparable = (a < 0x7F800000) & (a != 0) & ~dominated & (2 + run + es + 23 <= 33)
    got = pb.cpu().numpy()
    assert comparable.sum() > 2000 and np.array_equal(got[comparable], d[f"p{nbits}_{es}__pbits"][comparable])
    maxpat = (1 << (nbits - 1)) - 1
    assert (got[a == 0] == 0).all() and (np.abs(got) <= maxpat).all()
    big = (a < 0x7F800000) & dominated & (scale > 0)
    assert (np.abs(got[big]) == maxpat).all() and (np.sign(got[big]) == np.where(xb[big] >> 31, -1, 1)).all()

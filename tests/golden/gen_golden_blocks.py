"""Golden fixtures from upstream's quantizable BERT / MobileBERT twins and its evaluation flow (run by gen_golden.py; THIS
container only -- imports /root/reference through _ref_import).

  blocks.npz + blocks.json
        upstream `quantize(model, args)` on the drivers of mini_models.py: its TRANSFORMER_MODULE_MAPPINGS swap HF's
        BertSelfAttention / BertSelfOutput / BertOutput and the MobileBERT blocks for the twins of
        modules/quantizable/modeling_bert.py:33-214 and modeling_mobilebert.py:38-206; recorded per spec: module names
        and classes, hook (fake-quantizer) names, state-dict keys and shapes, the taps of every forward (attention
        context, attention output, block output) and the delayed-scaling state after every forward.
  qa_logits.npz + qa_logits.json
        start / end logits of a seeded tiny BERT QA model over padded SQuAD-style batches, collected the way
        examples/question_answering/run_qa_no_trainer.py:914-959 does (eval mode, no_grad, concatenated in batch order).
  fq_extra.npz + fq_extra.json
        FusedAmaxObsFakeQuantize with record_histogram / outlier_threshold (fake_quantize.py:348-359, 401-402): outputs,
        histogram buffer, max_outlier_pct, per call.
  attn_chain.npz + attn_chain.json
        the attention chain of upstream's BertSelfAttention twin (modeling_bert.py:118-158) tapped with forward (pre-)hooks
        registered AFTER upstream's quantize(): the fake-quantized q / k^T / probabilities / v as the matmuls receive them, the
        raw scores, the scaled scores, the softmax output and the context -- at fp32 and bf16, e4m3 and posit8_1, head_dim 64
        and 128.  Pins oracle.softmax_fq / attention_fq (tests/test_oracle_golden.py::test_attention_chain).
  checkpoint.npz + checkpoint.json
        a state_dict produced by upstream after calibration forwards (lazily sized amax_history / scale buffers,
        fake_quantize.py:406-435), the inputs of the next forward and upstream's outputs for it.
"""
import importlib
import json
import os

import numpy as np
import torch

import mini_models as mm

SPECS = {
    # name: (argv-style settings, dtype of the run, number of forwards)
    "posit8_1_act": (dict(activation="posit8_1", weight=None, quantize_forward="gemm"), torch.float32, 1),
    "e4m3_act_weight": (dict(activation="e4m3", weight="e4m3", quantize_forward="gemm"), torch.bfloat16, 1),
    "posit8_1_act_bf16": (dict(activation="posit8_1", weight=None, quantize_forward="gemm"), torch.bfloat16, 1),
    "e4m3_f32": (dict(activation="e4m3", weight="e4m3", quantize_forward="gemm"), torch.float32, 1),
    "int8_qs_all": (dict(activation="int8,qs=per_tensor_symmetric", weight="int8,qs=per_tensor_symmetric",
                         quantize_forward="gemm,residual,layernorm,activation,scaling"), torch.float32, 3),
}


def _helpers():
    import gen_golden as g
    return g


def _ref_twin_mappings(ref):
    """Fill upstream's (shim-emptied) TRANSFORMER_MODULE_MAPPINGS with its real twins for the HF classes of this image."""
    from transformers.models.bert import modeling_bert as hb
    from transformers.models.mobilebert import modeling_mobilebert as hm
    rb = importlib.import_module("quantized_training.modules.quantizable.modeling_bert")
    rm = importlib.import_module("quantized_training.modules.quantizable.modeling_mobilebert")
    mapping = ref.quantize.TRANSFORMER_MODULE_MAPPINGS
    mapping.clear()
    mapping.update({
        hb.BertSelfAttention: rb.BertSelfAttention, hb.BertSelfOutput: rb.BertSelfOutput, hb.BertOutput: rb.BertOutput,
        hm.MobileBertSelfAttention: rm.MobileBertSelfAttention, hm.MobileBertSelfOutput: rm.MobileBertSelfOutput,
        hm.FFNOutput: rm.FFNOutput, hm.MobileBertOutput: rm.MobileBertOutput,
    })
    # upstream's class list for the norm-quantized ops names MobileBERT's NoNorm (quantization_mappings.py:68-72)
    ln = ref.quantize.QCONFIG_PROPAGATE_MODULE_CLASS_LIST["layernorm"]
    if hm.NoNorm not in ln:
        ln.append(hm.NoNorm)
    from transformers.activations import GELUActivation
    act = ref.quantize.QCONFIG_PROPAGATE_MODULE_CLASS_LIST["activation"]
    if GELUActivation not in act:
        act.append(GELUActivation)


def make_args(ref, **kw):
    a = ref.training_args.add_qspec_args().parse_args([])
    for k, v in kw.items():
        setattr(a, k, v)
    return a


KINDS = {
    # name: (driver class, config, hidden size, specs)
    "bert": ("BertBlock", "tiny_bert_config", 64, None),
    "mobilebert": ("MobileBertBlock", "tiny_mobilebert_config", 64, None),
    "bert_hd64": ("BertBlock", "bert_hd64_config", 256, ("e4m3_act_weight", "posit8_1_act_bf16")),
}


def block_inputs(seed, dtype, B=2, S=24, H=64):
    r = np.random.default_rng(seed)
    h = torch.from_numpy(r.standard_normal((B, S, H)).astype(np.float32)).to(dtype)
    keep = torch.ones(B, S)
    keep[1, S - 5:] = 0                                    # one padded sequence
    return h, mm.additive_mask(keep, dtype)


def _state(m, g):
    out = {}
    for k, v in m.state_dict().items():
        if k.endswith(".scale") or k.endswith(".amax_history"):
            out[k] = g.canon_nan32(g.f32_bits(v.detach().float().reshape(-1)))
    return out


def gen_blocks(ref, out):
    g = _helpers()
    _ref_twin_mappings(ref)
    arrays, meta = {}, {}
    for kind, (build, cfg_fn, hidden, only) in KINDS.items():
        for sname, (kw, dtype, nfwd) in SPECS.items():
            if only is not None and sname not in only:
                continue
            cfg = getattr(mm, cfg_fn)()
            blk = mm.seeded_init_(getattr(mm, build)(cfg), 11, std=0.25 if hidden == 64 else 0.08).eval()
            if dtype == torch.bfloat16:
                blk = blk.bfloat16()
            ref.quantize.quantize(blk, make_args(ref, **kw, bf16=dtype == torch.bfloat16))
            key = f"{kind}/{sname}"
            info = {"dtype": str(dtype).replace("torch.", ""), "n_fwd": nfwd, "args": kw}
            with torch.no_grad():
                for i in range(nfwd):
                    h, mask = block_inputs(100 + i, dtype, H=hidden)
                    taps = {}
                    blk(h * (1.0 + 0.5 * i), mask, taps)
                    for t, v in taps.items():
                        arrays[f"{key}/fwd{i}/{t}"] = g.tensor_bits(v)
                    for k, v in _state(blk, g).items():
                        arrays[f"{key}/fwd{i}/sd/{k}"] = v
            info["modules"] = [(n, type(mod).__name__) for n, mod in blk.named_modules()]
            info["state_dict"] = {k: list(v.shape) for k, v in blk.state_dict().items()}
            info["fake_quantizers"] = sorted(n for n, mod in blk.named_modules() if type(mod).__name__ == "FusedAmaxObsFakeQuantize")
            meta[key] = info
    np.savez_compressed(os.path.join(out, "blocks.npz"), **{k.replace("/", "__"): v for k, v in arrays.items()})
    with open(os.path.join(out, "blocks.json"), "w") as f:
        json.dump(meta, f, indent=1)


def qa_batches(seed=7, n_batches=3, B=4, S=32, vocab=120):
    r = np.random.default_rng(seed)
    out = []
    for b in range(n_batches):
        ids = r.integers(3, vocab, size=(B, S)).astype(np.int64)
        keep = np.ones((B, S), dtype=np.int64)
        for row in range(B):
            pad = int(r.integers(0, S // 3))
            if pad:
                keep[row, S - pad:] = 0
                ids[row, S - pad:] = 0
        out.append((ids, keep))
    return out


def gen_qa_logits(ref, out):
    g = _helpers()
    _ref_twin_mappings(ref)
    arrays, meta = {}, {}
    batches = qa_batches()
    for i, (ids, keep) in enumerate(batches):
        arrays[f"batch{i}/input_ids"] = ids
        arrays[f"batch{i}/attention_mask"] = keep
    # (fixture name, spec, model): the BERT loop under three specs, and BASELINE.json's configs[0] at toy size -- MobileBERT with
    # posit(8,1) activations only, on the CPU
    for fname, sname, kind in (("posit8_1_act", "posit8_1_act", "bert"), ("e4m3_act_weight", "e4m3_act_weight", "bert"),
                               ("int8_qs_all", "int8_qs_all", "bert"), ("mobilebert_posit8_1_act", "posit8_1_act", "mobilebert")):
        kw, dtype, _ = SPECS[sname]
        model = mm.qa_model(kind)
        if dtype == torch.bfloat16:
            model = model.bfloat16()
        ref.quantize.quantize(model, make_args(ref, **kw, bf16=dtype == torch.bfloat16))
        starts, ends = [], []
        with torch.no_grad():                                              # run_qa_no_trainer.py:914-959
            for ids, keep in batches:
                o = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(keep))
                starts.append(o.start_logits.float())
                ends.append(o.end_logits.float())
        arrays[f"{fname}/start_logits"] = g.f32_bits(torch.cat(starts))
        arrays[f"{fname}/end_logits"] = g.f32_bits(torch.cat(ends))
        meta[fname] = {"dtype": str(dtype).replace("torch.", ""), "args": kw, "model": kind,
                       "fake_quantizers": sorted(n for n, m in model.named_modules() if type(m).__name__ == "FusedAmaxObsFakeQuantize"),
                       "state_dict": {k: list(v.shape) for k, v in model.state_dict().items()}}
        for k, v in _state(model, g).items():
            arrays[f"{fname}/sd/{k}"] = v
    np.savez_compressed(os.path.join(out, "qa_logits.npz"), **{k.replace("/", "__"): v for k, v in arrays.items()})
    with open(os.path.join(out, "qa_logits.json"), "w") as f:
        json.dump(meta, f, indent=1)


FQ_EXTRA = [
    # name, spec, input dtype, shape, extra constructor arguments, per-call input scale
    ("hist_e4m3_bf16", "e4m3", "bf16", (7, 96), dict(record_histogram=True), [1.0, 30.0, 1e-3]),
    ("hist_int8_qs_f32", "int8,qs=per_tensor_symmetric", "f32", (5, 64), dict(record_histogram=True), [1.0, 4.0]),
    ("outlier_int8_qs_bf16", "int8,qs=per_tensor_symmetric", "bf16", (6, 128), dict(outlier_threshold=2.0), [1.0, 3.0, 0.5]),
    ("outlier_e4m3_f32", "e4m3", "f32", (4, 100), dict(outlier_threshold=1.5), [1.0, 2.0]),
    ("outlier_disabled_bf16", "int8,qs=per_tensor_symmetric", "bf16", (4, 64), dict(outlier_threshold=1.0, disable=True), [1.0, 2.0]),
    ("outlier_hist_posit_bf16", "posit8_1", "bf16", (3, 5, 32), dict(outlier_threshold=4.0, record_histogram=True), [2.0, 6.0]),
]


def gen_fq_extra(ref, out):
    from dataclasses import asdict
    g = _helpers()
    rng = np.random.default_rng(77)
    meta, arrays = [], {}
    for name, spec, indt, shape, extra, sigmas in FQ_EXTRA:
        extra = dict(extra)
        disable = extra.pop("disable", False)
        kw = asdict(ref.quantizer.QuantizationSpec.from_str(spec))
        kw.update(extra)
        m = ref.fake_quantize.FusedAmaxObsFakeQuantize(**kw)
        if disable:                                        # both switches off: the outlier restore still runs
            m.disable_observer()
            m.disable_fake_quant()
        td = torch.bfloat16 if indt == "bf16" else torch.float32
        calls = []
        for ci, sg in enumerate(sigmas):
            x = (rng.standard_normal(shape) * sg).astype(np.float32)
            x.flat[:4] = [0.0, -0.0, 1.5 * sg, -7.0 * sg]
            xt = torch.from_numpy(x).to(td)
            with torch.no_grad():
                y = m(xt)
            k = f"{name}/{ci}"
            arrays[k + "/x"] = g.tensor_bits(xt) if td == torch.bfloat16 else g.f32_bits(xt)
            arrays[k + "/y"] = g.tensor_bits(y)
            arrays[k + "/histogram"] = g.f32_bits(m.histogram.detach().clone().float())
            arrays[k + "/scale"] = g.f32_bits(m.scale.detach().clone().float().reshape(-1))
            calls.append({"max_outlier_pct": float(getattr(m, "max_outlier_pct", -1.0))})
        meta.append({"name": name, "spec": spec, "in": indt, "shape": list(shape), "extra": extra, "disable": disable,
                     "n_calls": len(sigmas), "calls": calls, "buffers": sorted(n for n, _ in m.named_buffers()),
                     "state_dict": sorted(m.state_dict().keys())})
    np.savez_compressed(os.path.join(out, "fq_extra.npz"), **{k.replace("/", "__"): v for k, v in arrays.items()})
    with open(os.path.join(out, "fq_extra.json"), "w") as f:
        json.dump(meta, f, indent=1)


def gen_checkpoint(ref, out):
    """Calibrate upstream's converted block (2 forwards with live observers), take its state_dict, freeze the observers
    (run_qa_no_trainer.py:834-847: `disable_observer` after calibration), run one more forward."""
    g = _helpers()
    _ref_twin_mappings(ref)
    kw, dtype, _ = SPECS["int8_qs_all"]
    blk = mm.seeded_init_(mm.BertBlock(mm.tiny_bert_config()), 11).eval()
    ref.quantize.quantize(blk, make_args(ref, **kw))
    arrays = {}
    with torch.no_grad():
        for i in range(2):
            h, mask = block_inputs(200 + i, dtype)
            blk(h * (1.0 + i), mask)
        sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
        for mod in blk.modules():
            if type(mod).__name__ == "FusedAmaxObsFakeQuantize":
                mod.disable_observer()
        h, mask = block_inputs(300, dtype)
        taps = {}
        blk(h, mask, taps)
    for k, v in sd.items():
        arrays["sd/" + k] = v.numpy() if v.dtype in (torch.uint8, torch.int64) else g.f32_bits(v.float())
    for t, v in taps.items():
        arrays["after/" + t] = g.tensor_bits(v)
    meta = {"args": kw, "state_dict": {k: {"shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", "")} for k, v in sd.items()},
            "calibration_seeds": [200, 201], "eval_seed": 300}
    np.savez_compressed(os.path.join(out, "checkpoint.npz"), **{k.replace("/", "__"): v for k, v in arrays.items()})
    with open(os.path.join(out, "checkpoint.json"), "w") as f:
        json.dump(meta, f, indent=1)


POSIT_OPTS = [(8, 0), (8, 1), (8, 2), (16, 1), (6, 1), (16, 2)]


def gen_posit_opts(ref, out):
    """quantize_to_posit(x, nbits, es, round_to_even=False) and (..., return_pbits=True) (posit.py:27-35, 50-53, 60-65)."""
    g = _helpers()
    rng = np.random.default_rng(31)
    x = g.sample_f32(rng, 6000)
    arrays = {"x": g.f32_bits(torch.from_numpy(x))}
    for nbits, es in POSIT_OPTS:
        xt = torch.from_numpy(x)
        y0 = ref.posit.quantize_to_posit(xt, nbits, es, round_to_even=False)
        y1, pb = ref.posit.quantize_to_posit(xt, nbits, es, return_pbits=True)
        arrays[f"p{nbits}_{es}/y_no_rte"] = g.canon_nan32(g.f32_bits(y0))
        arrays[f"p{nbits}_{es}/y"] = g.canon_nan32(g.f32_bits(y1))
        arrays[f"p{nbits}_{es}/pbits"] = pb.numpy().astype(np.int32)
    np.savez_compressed(os.path.join(out, "posit_opts.npz"), **{k.replace("/", "__"): v for k, v in arrays.items()})


# ---- the attention chain of upstream's BertSelfAttention twin, tapped -------------------------------------------------------
ATTN_CASES = {
    # name: (activation spec, dtype, heads, head_dim, B, S)
    "e4m3_bf16_hd64": ("e4m3", torch.bfloat16, 2, 64, 2, 40),
    "e4m3_bf16_hd128": ("e4m3", torch.bfloat16, 2, 128, 2, 40),
    "posit8_1_bf16_hd64": ("posit8_1", torch.bfloat16, 2, 64, 2, 40),
    "posit8_1_bf16_hd128": ("posit8_1", torch.bfloat16, 2, 128, 1, 40),
    "e4m3_f32_hd64": ("e4m3", torch.float32, 2, 64, 2, 40),
    "posit8_1_f32_hd128": ("posit8_1", torch.float32, 2, 128, 1, 40),
}


def gen_attn_chain(ref, out):
    g = _helpers()
    _ref_twin_mappings(ref)
    from transformers import BertConfig
    arrays, meta = {}, {}
    for name, (spec, dtype, heads, hd, B, S) in ATTN_CASES.items():
        cfg = BertConfig(hidden_size=heads * hd, num_hidden_layers=1, num_attention_heads=heads, intermediate_size=2 * heads * hd, vocab_size=120,
                         max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        blk = mm.seeded_init_(mm.BertBlock(cfg), 21, std=0.9 / np.sqrt(heads * hd)).eval()
        if dtype == torch.bfloat16:
            blk = blk.bfloat16()
        ref.quantize.quantize(blk, make_args(ref, activation=spec, weight=None, quantize_forward="gemm", bf16=dtype == torch.bfloat16))
        att = blk.attention.self
        taps = {}

        def pre(tag):
            def hook(mod, args):
                for i, t in enumerate(args):
                    if torch.is_tensor(t):
                        taps[f"{tag}.in{i}"] = t.detach().clone()
            return hook

        def post(tag):
            def hook(mod, args, outp):
                taps[f"{tag}.out"] = outp.detach().clone()
            return hook
        hs = []
        for tag in ("qk_matmul", "attn_scaling", "softmax", "av_matmul"):
            m = getattr(att, tag)
            hs.append(m.register_forward_pre_hook(pre(tag)))          # registered after upstream's hooks: sees what the op receives
            hs.append(m.register_forward_hook(post(tag)))
        r = np.random.default_rng(300 + hd)
        h = torch.from_numpy((r.standard_normal((B, S, heads * hd)) * 1.5).astype(np.float32)).to(dtype)
        keep = torch.ones(B, S)
        keep[B - 1, S - 7:] = 0
        mask = mm.additive_mask(keep, dtype)
        with torch.no_grad():
            ctx = att(h, mask)[0]
        for hk in hs:
            hk.remove()
        taps["mask"] = mask
        taps["context"] = ctx
        for k, v in taps.items():
            arrays[f"{name}/{k}"] = g.tensor_bits(v)
        meta[name] = {"spec": spec, "dtype": str(dtype).replace("torch.", ""), "heads": heads, "head_dim": hd, "B": B, "S": S,
                      "scaling": 1.0 / float(np.sqrt(hd)), "taps": {k: list(v.shape) for k, v in taps.items()}}
    np.savez_compressed(os.path.join(out, "attn_chain.npz"), **{k.replace("/", "__"): v for k, v in arrays.items()})
    with open(os.path.join(out, "attn_chain.json"), "w") as f:
        json.dump(meta, f, indent=1)

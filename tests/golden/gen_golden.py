#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the upstream
reference (imported from /root/reference in this container only).

    python tests/golden/gen_golden.py

Outputs (all data, no reference source text):
  maps.npz          65 536-entry value maps, uint16 bf16 bit patterns, every NaN
                    rewritten to 0x7FC0 (payload is not part of the contract),
                    one array per dtype  (reference: fake_quantize.py:31-95)
  maps_sha256.json  SHA-256 of each table (little-endian uint16, index order)
  direct_fns.npz    fp32 inputs -> outputs of quantize_to_fp8_e4m3/_e5m2/
                    quantize_to_posit called directly  (fp8.py:10-67, posit.py:6-67)
  vmap.npz          bf16 / fp32 inputs -> quantized_ops.vmap outputs
                    (decomposed.py:146-163)
  quant_dequant.npz quantized_ops.quantize / dequantize vectors (decomposed.py:166-262)
  fake_quant.npz + fake_quant.json
                    multi-call traces of FusedAmaxObsFakeQuantize.forward
                    (outputs, scale and amax_history after every call)
                    (fake_quantize.py:197-248, 343-404)
  eager_trace.json + eager_trace.npz
                    quantize(model,args) on a toy model: module / state-dict keys,
                    forward outputs, 3-step training trace with backward hooks
                    (quantize.py:52-193)
  wikitext_windows.json
                    sliding-window schedule of examples/language_modeling/wikitext.py:143-165
  blocks.*, qa_logits.*, fq_extra.*, checkpoint.*
                    upstream's quantizable BERT / MobileBERT twins, QA-loop logits, histogram / outlier options and a
                    calibrated state_dict: see gen_golden_blocks.py
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

MAP_DTYPES = [
    "int8", "int6", "int4", "int3", "int2", "uint8", "uint4",
    "e4m3", "e5m2",
    "fp8_e4m3", "fp8_e5m2", "fp8_e5m3", "fp6_e3m2", "fp6_e2m3", "fp4_e2m1",
    "posit8_0", "posit8_1", "posit8_2", "posit8_3", "posit6_1", "posit4_1",
    "posit16_1", "posit16_2",
    "float16", "bfloat16", "float32",
]


def bf16_bits(t):
    assert t.dtype == torch.bfloat16
    return t.contiguous().view(torch.int16).numpy().astype(np.uint16)


def canon_nan16(bits):
    bits = bits.copy()
    nan = ((bits & 0x7F80) == 0x7F80) & ((bits & 0x007F) != 0)
    bits[nan] = 0x7FC0
    return bits


def f32_bits(t):
    assert t.dtype == torch.float32
    return t.contiguous().view(torch.int32).numpy().astype(np.uint32)


def canon_nan32(bits):
    bits = bits.copy()
    nan = ((bits & 0x7F800000) == 0x7F800000) & ((bits & 0x007FFFFF) != 0)
    bits[nan] = 0x7FC00000
    return bits


def tensor_bits(t):
    if t.dtype == torch.bfloat16:
        return canon_nan16(bf16_bits(t))
    return canon_nan32(f32_bits(t.float()))


def edge_f32():
    vals = [0.0, -0.0, 1.0, -1.0, 3.140625, -100.5, 448.0, 449.0, 464.0, 480.0, 1e9, -1e9,
            57344.0, 61440.0, 65536.0, 2.0 ** -6, 2.0 ** -7, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -10,
            2.0 ** -14, 2.0 ** -16, 2.0 ** -17, 1.0001 * 2.0 ** -17, 2.0 ** -24, 2.0 ** -126, 1e-45,
            float("inf"), float("-inf"), float("nan"), 0.5, 1.5, 2.5, 3.5, 100.5, 100.7, 126.5, 127.5,
            -127.5, -128.5, 127.49, 255.5, 4096.0, 5000.0, 2.0 ** 24, 2.0 ** 25, 2.0 ** -12, 2.0 ** -13,
            1.0625, 1.1875, 1.125, 1.25, 1.375, 3.0e38, -3.0e38]
    return np.array(vals, dtype=np.float32)


def sample_f32(rng, n):
    """Mixed-sigma normals + log-uniform magnitudes + edge values."""
    a = rng.standard_normal(n // 2).astype(np.float32) * np.float32(10.0) ** rng.uniform(-3, 3, n // 2).astype(np.float32)
    e = rng.uniform(-30, 30, n - n // 2 - 64)
    b = (np.sign(rng.standard_normal(e.size)) * 2.0 ** e).astype(np.float32)
    edges = edge_f32()
    c = np.resize(edges, 64)
    return np.concatenate([a, b, c]).astype(np.float32)


def gen_maps(ref, out):
    maps, shas = {}, {}
    for dt in MAP_DTYPES + [None]:
        m = ref.fake_quantize.get_quantization_map(dt)
        bits = canon_nan16(bf16_bits(m))
        key = "none" if dt is None else dt
        maps[key] = bits
        shas[key] = hashlib.sha256(bits.astype("<u2").tobytes()).hexdigest()
    for dt in ["nf4", "nf4_6", "nf3", "nf2_4"]:
        idx, vals = ref.fake_quantize.get_quantization_map(dt)
        maps[dt] = canon_nan16(bf16_bits(vals[idx]))
        maps[dt + "__values"] = canon_nan16(bf16_bits(vals))
        maps[dt + "__indices"] = idx.numpy().astype(np.uint16)
        shas[dt] = hashlib.sha256(maps[dt].astype("<u2").tobytes()).hexdigest()
    np.savez_compressed(os.path.join(out, "maps.npz"), **maps)
    with open(os.path.join(out, "maps_sha256.json"), "w") as f:
        json.dump(shas, f, indent=1, sort_keys=True)
    return maps


def gen_direct(ref, out):
    rng = np.random.default_rng(1234)
    x = sample_f32(rng, 8192)
    xt = torch.from_numpy(x)
    d = {"x": f32_bits(xt)}
    d["e4m3"] = canon_nan32(f32_bits(ref.fp8.quantize_to_fp8_e4m3(xt)))
    d["e5m2"] = canon_nan32(f32_bits(ref.fp8.quantize_to_fp8_e5m2(xt)))
    for nb, es in [(8, 0), (8, 1), (8, 2), (16, 1), (16, 2), (6, 1)]:
        d[f"posit{nb}_{es}"] = canon_nan32(f32_bits(ref.posit.quantize_to_posit(xt, nb, es, round_to_even=True)))
    # bf16 inputs too (function keeps input dtype)
    xb = xt.to(torch.bfloat16)
    d["xb"] = bf16_bits(xb)
    d["e4m3_b"] = canon_nan16(bf16_bits(ref.fp8.quantize_to_fp8_e4m3(xb)))
    d["e5m2_b"] = canon_nan16(bf16_bits(ref.fp8.quantize_to_fp8_e5m2(xb)))
    d["posit8_1_b"] = canon_nan16(bf16_bits(ref.posit.quantize_to_posit(xb, 8, 1)))
    np.savez_compressed(os.path.join(out, "direct_fns.npz"), **d)


def gen_vmap(ref, out):
    rng = np.random.default_rng(77)
    x32 = sample_f32(rng, 4096)
    # force some exact-bf16 values (low 16 bits == 0) and near-bf16 values (sticky fold)
    b = x32.view(np.uint32).copy()
    b[::7] &= 0xFFFF0000
    b[1::7] = (b[1::7] & 0xFFFF0000) | 1
    x32 = b.view(np.float32)
    d = {"x32": b.astype(np.uint32)}
    xb = torch.from_numpy(x32).to(torch.bfloat16)
    d["xb"] = bf16_bits(xb)
    xh = torch.from_numpy(x32).to(torch.float16)
    d["xh"] = xh.view(torch.int16).numpy().astype(np.uint16)
    for dt in ["int8", "int4", "e4m3", "e5m2", "fp8_e4m3", "fp4_e2m1", "posit8_1", "posit16_1"]:
        qmap = ref.fake_quantize.get_quantization_map(dt)
        y32 = ref.decomposed.vmap(torch.from_numpy(x32), qmap)
        yb = ref.decomposed.vmap(xb, qmap)
        yh = ref.decomposed.vmap(xh, qmap)
        assert y32.dtype == torch.float32 and yb.dtype == torch.bfloat16 and yh.dtype == torch.float16
        d[f"y32_{dt}"] = canon_nan32(f32_bits(y32))
        d[f"yb_{dt}"] = canon_nan16(bf16_bits(yb))
        d[f"yh_{dt}"] = yh.view(torch.int16).numpy().astype(np.uint16)
    np.savez_compressed(os.path.join(out, "vmap.npz"), **d)


def gen_quant_dequant(ref, out):
    rng = np.random.default_rng(99)
    x32 = (rng.standard_normal(2048) * 3).astype(np.float32)
    xb = torch.from_numpy(x32).to(torch.bfloat16)
    d = {"x32": f32_bits(torch.from_numpy(x32)), "xb": bf16_bits(xb)}
    cases = []
    q = torch.ops.quantized_ops
    for i, (dt, s) in enumerate([("int8", 0.037), ("e4m3", 1.0), ("fp8_e4m3", 0.0123), ("posit8_1", 0.5), ("int4", 0.61)]):
        qmap = ref.fake_quantize.get_quantization_map(dt)
        for tag, x in (("f32", torch.from_numpy(x32)), ("bf16", xb)):
            scale = torch.tensor([s], dtype=x.dtype)
            yq = q.quantize(x, scale, None, None, None, qmap)
            ydq = q.dequantize(yq, scale, None, None, None, None, None)
            ydq2 = q.dequantize(x, scale, None, None, None, qmap, qmap)
            name = f"c{i}_{tag}"
            d[name + "_scale"] = tensor_bits(scale)
            d[name + "_q"] = tensor_bits(yq)
            d[name + "_dq"] = tensor_bits(ydq)
            d[name + "_dq2"] = tensor_bits(ydq2)
            cases.append({"name": name, "dtype": dt, "in": tag, "scale": s})
    # zero-point variant
    qmap = ref.fake_quantize.get_quantization_map("uint8")
    x = torch.from_numpy(x32)
    scale = torch.tensor([0.05]); zp = torch.tensor([128.0])
    yq = q.quantize(x, scale, zp, None, None, qmap)
    ydq = q.dequantize(yq, scale, zp, None, None, None, None)
    d["zp_q"] = tensor_bits(yq); d["zp_dq"] = tensor_bits(ydq)
    np.savez_compressed(os.path.join(out, "quant_dequant.npz"), **d)
    with open(os.path.join(out, "quant_dequant.json"), "w") as f:
        json.dump(cases, f, indent=1)


FQ_CASES = [
    # name, spec string, input dtype, shape, sigmas per call, force_pow2
    ("e4m3_noqs_bf16", "e4m3", "bf16", (64, 96), [1.0, 30.0, 0.01], False),
    ("e4m3_noqs_f32", "e4m3", "f32", (64, 96), [1.0, 30.0, 0.01], False),
    ("posit8_1_noqs_bf16", "posit8_1", "bf16", (32, 64), [1.0, 5.0], False),
    ("posit8_2_noqs_f32", "posit8_2", "f32", (32, 64), [1.0, 5.0], False),
    ("fp8_e4m3_qs_bf16", "fp8_e4m3,qs=per_tensor_symmetric", "bf16", (48, 80), [1.0, 7.0, 0.3, 0.3, 2.0], False),
    ("fp8_e4m3_qs_f32", "fp8_e4m3,qs=per_tensor_symmetric", "f32", (48, 80), [1.0, 7.0, 0.3, 0.3, 2.0], False),
    ("int8_qs_bf16", "int8,qs=per_tensor_symmetric", "bf16", (48, 80), [1.0, 7.0, 0.3, 0.3, 2.0], False),
    ("int8_qs_f32", "int8,qs=per_tensor_symmetric", "f32", (48, 80), [1.0, 7.0, 0.3, 0.3, 2.0], False),
    ("int8_qs_ahl3_f32", "int8,qs=per_tensor_symmetric,ahl=3", "f32", (16, 32), [5, 9, 2, 1, 1, 1, 1, 7], False),
    ("int4_qs_ahl1_bf16", "int4,qs=per_tensor_symmetric,ahl=1", "bf16", (16, 32), [5, 9, 2, 1], False),
    ("fp8_e5m2_qs_qmax_bf16", "fp8_e5m2,qs=per_tensor_symmetric,qmax=57344,ahl=10", "bf16", (16, 32), [1e-3, 1e-2, 1e-3], False),
    ("e4m3_qs_pow2_bf16", "fp8_e4m3,qs=per_tensor_symmetric", "bf16", (16, 32), [1.0, 7.0, 0.3, 0.3], True),
    ("int8_qs_pow2_f32", "int8,qs=per_tensor_symmetric", "f32", (16, 32), [1.0, 7.0, 0.3, 0.3], True),
    ("posit8_1_qs_bf16", "posit8_1,qs=per_tensor_symmetric,qmax=64", "bf16", (16, 32), [1.0, 7.0, 0.3], False),
    ("int8_pc0_bf16", "int8,qs=per_channel_symmetric,ax=0", "bf16", (24, 40), [1.0, 3.0, 0.5], False),
    ("int8_pc1_f32", "int8,qs=per_channel_symmetric,ax=-1", "f32", (24, 40), [1.0, 3.0, 0.5], False),
    ("fp8_e4m3_pc1_3d_bf16", "fp8_e4m3,qs=per_channel_symmetric,ax=1", "bf16", (4, 6, 20), [1.0, 3.0, 0.5], False),
    ("int8_qs_zero_nan_f32", "int8,qs=per_tensor_symmetric,ahl=2", "f32", (8, 16), ["zero", 2.0, "nan", 3.0, 1.0, 1.0], False),
]


def gen_fake_quant(ref, out):
    from dataclasses import asdict
    rng = np.random.default_rng(2024)
    meta, arrays = [], {}
    for name, spec, indt, shape, sigmas, pow2 in FQ_CASES:
        kw = asdict(ref.quantizer.QuantizationSpec.from_str(spec))
        m = ref.fake_quantize.FusedAmaxObsFakeQuantize(**kw, force_scale_power_of_two=pow2)
        td = torch.bfloat16 if indt == "bf16" else torch.float32
        calls = []
        for ci, sg in enumerate(sigmas):
            if sg == "zero":
                x = np.zeros(shape, dtype=np.float32)
            elif sg == "nan":
                x = rng.standard_normal(shape).astype(np.float32)
                x.flat[3] = np.nan
            else:
                x = (rng.standard_normal(shape) * float(sg)).astype(np.float32)
                if ci == 0:
                    x.flat[:8] = [0.0, -0.0, 0.5, -0.5, 1.5, 2.5, -2.5, 3.5]
            xt = torch.from_numpy(x).to(td)
            with torch.no_grad():
                y = m(xt)
            assert y.dtype == td
            k = f"{name}/{ci}"
            arrays[k + "/x"] = tensor_bits(xt) if td == torch.bfloat16 else f32_bits(xt)
            arrays[k + "/y"] = tensor_bits(y)
            arrays[k + "/scale"] = f32_bits(m.scale.detach().clone().float().reshape(-1))
            arrays[k + "/hist"] = canon_nan32(f32_bits(m.amax_history.detach().clone().float().reshape(-1)))
            calls.append({"scale_shape": list(m.scale.shape), "hist_shape": list(m.amax_history.shape)})
        sd = {k: list(v.shape) for k, v in m.state_dict().items()}
        meta.append({"name": name, "spec": spec, "in": indt, "shape": list(shape), "pow2": pow2,
                     "n_calls": len(sigmas), "calls": calls, "state_dict": sd,
                     "quant_max": kw["quant_max"], "amax_history_len": kw["amax_history_len"],
                     "ch_axis": kw["ch_axis"],
                     "buffers": sorted(n for n, _ in m.named_buffers())})
    np.savez_compressed(os.path.join(out, "fake_quant.npz"), **{k.replace("/", "__"): v for k, v in arrays.items()})
    with open(os.path.join(out, "fake_quant.json"), "w") as f:
        json.dump(meta, f, indent=1)


MX_CASES = [
    # name, spec, input dtype, shape, force_pow2
    ("mx_int8_bs32_bf16", "int8,qs=microscaling,bs=32,ax=-1", "bf16", (6, 128), False),
    ("mx_int8_bs32_f32", "int8,qs=microscaling,bs=32,ax=-1", "f32", (6, 128), False),
    ("mx_fp4_bs16_bf16", "fp4_e2m1,qs=microscaling,bs=16,ax=-1", "bf16", (4, 5, 64), False),
    ("mx_fp8_pow2_bf16", "fp8_e4m3,qs=microscaling,bs=32,ax=-1", "bf16", (8, 96), True),
    ("mx_int6_pow2_f32", "int6,qs=microscaling,bs=64,ax=-1", "f32", (3, 192), True),
    ("mx_int6_ax_m2_bf16", "int6,qs=microscaling,bs=64,ax=-2", "bf16", (2, 128, 24), False),
    ("mx_int4_ragged_f32", "int4,qs=microscaling,bs=32,ax=-1", "f32", (5, 80), False),
    ("mx_int8_scaleq_bf16", "int8,qs=microscaling,bs=32,ax=-1,scale=fp8_e5m3", "bf16", (6, 128), False),
    ("gwa_uint4_bs32_f32", "uint4,qs=group_wise_affine,bs=32,ax=-1", "f32", (6, 128), False),
    ("gwa_uint8_bs64_bf16", "uint8,qs=group_wise_affine,bs=64,ax=-1", "bf16", (4, 128), False),
    ("mx_nf4_bs64_bf16", "nf4,qs=microscaling,bs=64,ax=-1", "bf16", (6, 128), False),
    ("mx_nf4_6_scaleq_bf16", "nf4_6,qs=microscaling,bs=64,ax=-1,scale=fp8_e5m3", "bf16", (6, 128), False),
    ("mx_nf4_ax_m2_f32", "nf4,qs=microscaling,bs=64,ax=-2", "f32", (2, 128, 16), False),
]


def gen_mx(ref, out):
    """Block-scaled fake-quant traces: MXFakeQuantFunction / GroupWiseAffineFakeQuantFunction through the
    module (fake_quantize.py:98-194, decomposed.py:365-448)."""
    from dataclasses import asdict
    rng = np.random.default_rng(31)
    meta, arrays = [], {}
    for name, spec, indt, shape, pow2 in MX_CASES:
        kw = asdict(ref.quantizer.QuantizationSpec.from_str(spec))
        m = ref.fake_quantize.FusedAmaxObsFakeQuantize(**kw, force_scale_power_of_two=pow2)
        td = torch.bfloat16 if indt == "bf16" else torch.float32
        x = (rng.standard_normal(shape) * 10.0 ** rng.uniform(-2, 1, shape[:-1] + (1,))).astype(np.float32)
        x.flat[:4] = [0.0, -0.0, 1.0, -3.5]
        x[..., -1, :] = 0.0 if len(shape) > 1 else x[..., -1, :]           # an all-zero row -> scale 1
        xt = torch.from_numpy(x).to(td)
        with torch.no_grad():
            y = m(xt)
        arrays[name + "/x"] = tensor_bits(xt) if td == torch.bfloat16 else f32_bits(xt)
        arrays[name + "/y"] = tensor_bits(y)
        arrays[name + "/scale"] = canon_nan32(f32_bits(m.scale.detach().float().reshape(-1)))
        e = {"name": name, "spec": spec, "in": indt, "shape": list(shape), "pow2": pow2,
             "scale_shape": list(m.scale.shape), "quant_max": kw["quant_max"], "quant_min": kw["quant_min"],
             "block_size": kw["block_size"], "ch_axis": kw["ch_axis"], "scale_dtype": kw["scale_dtype"]}
        if "group_wise" in spec:
            arrays[name + "/zp"] = canon_nan32(f32_bits(m.zero_point.detach().float().reshape(-1)))
            e["zp_shape"] = list(m.zero_point.shape)
        meta.append(e)
    np.savez_compressed(os.path.join(out, "mx.npz"), **{k.replace("/", "__"): v for k, v in arrays.items()})
    with open(os.path.join(out, "mx.json"), "w") as f:
        json.dump(meta, f, indent=1)


def _toy(ref):
    import torch.nn as nn
    fm = ref.functional_modules

    class Block(nn.Module):
        def __init__(self, d):
            super().__init__()
            self.query = nn.Linear(d, d)
            self.key = nn.Linear(d, d)
            self.value = nn.Linear(d, d)
            self.qk_matmul = fm.MatmulFunctional()
            self.attn_scaling = fm.MulFunctional()
            self.softmax = nn.Softmax(dim=-1)
            self.av_matmul = fm.MatmulFunctional()
            self.dense = nn.Linear(d, d)
            self.residual = fm.AddFunctional()
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

    return Toy


def gen_eager(ref, out):
    """quantize(model,args) traces on a toy model (no HF dependency)."""
    Toy = _toy(ref)
    arrays, meta = {}, {}
    rng = np.random.default_rng(5)

    def make_args(**kw):
        p = ref.training_args.add_qspec_args()
        a = p.parse_args([])
        for k, v in kw.items():
            setattr(a, k, v)
        return a

    def init_model(seed):
        torch.manual_seed(seed)
        m = Toy()
        # deterministic, generator-independent parameter init
        r = np.random.default_rng(seed)
        with torch.no_grad():
            for n, p in sorted(m.named_parameters()):
                p.copy_(torch.from_numpy((r.standard_normal(tuple(p.shape)) * 0.3).astype(np.float32)))
        return m

    x_np = (rng.standard_normal((3, 5, 16))).astype(np.float32)
    arrays["x"] = f32_bits(torch.from_numpy(x_np))
    sd0 = init_model(0).state_dict()
    for k, v in sd0.items():
        arrays["param/" + k] = f32_bits(v.float())

    runs = [
        ("eval_e4m3_gemm", dict(activation="e4m3", weight="e4m3", quantize_forward="gemm"), False),
        ("eval_posit_all", dict(activation="posit8_1", weight="posit8_1",
                                quantize_forward="gemm,residual,activation,layernorm,scaling"), False),
        ("eval_int8_qs_gemm", dict(activation="int8,qs=per_tensor_symmetric", weight="int8,qs=per_tensor_symmetric",
                                   quantize_forward="gemm"), False),
        ("eval_fusion", dict(activation="e4m3", weight=None, quantize_forward="gemm,residual",
                             op_fusion=["layer1.dense", "qk_matmul"]), False),
        ("eval_bf16", dict(activation="fp8_e4m3,qs=per_tensor_symmetric", weight="fp8_e4m3,qs=per_tensor_symmetric",
                           quantize_forward="gemm,scaling", bf16=True), False),
        ("train_int8_e5m2", dict(activation="int8,qs=per_tensor_symmetric", weight="int8,qs=per_tensor_symmetric",
                                 error="fp8_e5m2,qs=per_tensor_symmetric,qmax=57344,ahl=10",
                                 quantize_forward="gemm", quantize_backprop="gemm,residual"), True),
    ]
    for name, kw, train in runs:
        m = init_model(0)
        args = make_args(**kw)
        ref.quantize.quantize(m, args)
        x = torch.from_numpy(x_np)
        if kw.get("bf16"):
            x = x.bfloat16()
        info = {"args": {k: v for k, v in kw.items()}}
        if not train:
            m.eval()
            outs = []
            with torch.no_grad():
                for i in range(3):
                    y = m(x * (1.0 + i))
                    arrays[f"{name}/y{i}"] = tensor_bits(y)
            info["n_fwd"] = 3
        else:
            m.train()
            opt = torch.optim.SGD(m.parameters(), lr=0.05)
            losses = []
            for i in range(3):
                xi = (x * (1.0 + 0.5 * i)).requires_grad_(True)
                y = m(xi)
                loss = (y.float() ** 2).mean()
                opt.zero_grad()
                loss.backward()
                arrays[f"{name}/y{i}"] = tensor_bits(y.detach())
                arrays[f"{name}/gx{i}"] = tensor_bits(xi.grad)
                arrays[f"{name}/gw{i}"] = tensor_bits(m.layer0.query.weight.grad)
                opt.step()
                losses.append(float(loss))
            info["losses"] = losses
        sd = m.state_dict()
        info["state_dict"] = {k: list(v.shape) for k, v in sd.items()}
        info["modules"] = [(n, type(mod).__name__) for n, mod in m.named_modules()]
        for k, v in sd.items():
            if k.endswith(".scale") or k.endswith(".amax_history"):
                arrays[f"{name}/sd/{k}"] = canon_nan32(f32_bits(v.float().reshape(-1)))
        meta[name] = info
    np.savez_compressed(os.path.join(out, "eager_trace.npz"), **{k.replace("/", "__"): v for k, v in arrays.items()})
    with open(os.path.join(out, "eager_trace.json"), "w") as f:
        json.dump(meta, f, indent=1)


def _graph_rows(gm):
    def nm(a):
        if isinstance(a, torch.fx.Node):
            return a.name
        if isinstance(a, (list, tuple)):
            return [nm(x) for x in a]
        return a if isinstance(a, (int, float, str, bool, type(None))) else str(a)
    return [[n.op, n.name, str(n.target), nm(list(n.args))] for n in gm.graph.nodes]


def gen_pt2e(ref, out):
    """PT2E flow on a toy model: prepared graph (inserted fake-quant modules), calibrated outputs and
    scales, converted graph and outputs (quantize_pt2e.py:155-273, 323-446, 975-1002)."""
    import torch.nn as nn
    qp = ref.quantize_pt2e
    assert qp is not None, getattr(ref, "pt2e_error", None)

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

    rng = np.random.default_rng(17)
    arrays, meta = {}, {}
    xs = [(rng.standard_normal((4, 8, 16)) * (i + 1)).astype(np.float32) for i in range(4)]
    for i, x in enumerate(xs):
        arrays[f"x{i}"] = f32_bits(torch.from_numpy(x))
    runs = [("int8", dict(input_activation="int8,qs=per_tensor_symmetric", weight="int8,qs=per_tensor_symmetric", bias="int24"), None),
            ("fp8", dict(input_activation="fp8_e4m3,qs=per_tensor_symmetric", weight="fp8_e4m3,qs=per_tensor_symmetric", bias="float32"), "bfloat16"),
            ("e4m3_noqs", dict(input_activation="e4m3", weight="e4m3"), None)]
    for name, kw, out_dtype in runs:
        m = Toy().eval()
        r = np.random.default_rng(3)
        with torch.no_grad():
            for n, p in sorted(m.named_parameters()):
                p.copy_(torch.from_numpy((r.standard_normal(tuple(p.shape)) * 0.3).astype(np.float32)))
        if name == "int8":
            for n, p in m.named_parameters():
                arrays["param/" + n] = f32_bits(p.detach())
        q = qp.get_default_quantizer(**kw)
        gm = qp.prepare_pt2e(m, q, (torch.from_numpy(xs[0]),))
        info = {"kw": kw, "output_dtype": out_dtype, "prepared_graph": _graph_rows(gm),
                "fq_modules": [n for n, mod in gm.named_modules() if isinstance(mod, torch.ao.quantization.FakeQuantizeBase)]}
        with torch.no_grad():
            for i in range(3):
                gm(torch.from_numpy(xs[i]))
            y1 = gm(torch.from_numpy(xs[3]))
        arrays[f"{name}/y_prepared"] = tensor_bits(y1)
        info["scales"] = {k: [float(t) for t in v.reshape(-1)] for k, v in gm.state_dict().items() if k.endswith(".scale")}
        gc = qp.convert_pt2e(gm, out_dtype) if out_dtype else qp.convert_pt2e(gm)
        info["converted_graph"] = _graph_rows(gc)
        with torch.no_grad():
            y2 = gc(torch.from_numpy(xs[3]))
        arrays[f"{name}/y_converted"] = tensor_bits(y2)
        info["converted_buffers"] = {k: [float(t) for t in v.reshape(-1)][:4] for k, v in gc.named_buffers() if "scale" in k}
        meta[name] = info
    np.savez_compressed(os.path.join(out, "pt2e.npz"), **{k.replace("/", "__"): v for k, v in arrays.items()})
    with open(os.path.join(out, "pt2e.json"), "w") as f:
        json.dump(meta, f, indent=1)


def gen_pt2e_patterns(ref, out):
    """The remaining static patterns of the reference quantizer (xnnpack_quantizer.py:160-168 STATIC_OPS: activation,
    softmax, layer_norm next to linear / matmul / residual) reached through set_global and through set_object_type /
    set_module_name: prepared graph, inserted modules and calibrated output."""
    import torch.nn as nn
    qp = ref.quantize_pt2e
    assert qp is not None, getattr(ref, "pt2e_error", None)
    from quantized_training.quantizer.xnnpack_quantizer import XNNPACKQuantizer
    from quantized_training.quantizer.xnnpack_quantizer_utils import QuantizationConfig
    from quantized_training.quantizer.quantizer import QuantizationSpec
    from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize

    class Toy(nn.Module):
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

    def spec(s):
        q = QuantizationSpec.from_str(s)
        q.observer_or_fake_quant_ctr = FusedAmaxObsFakeQuantize.with_args(record_histogram=False, force_scale_power_of_two=False)
        return q

    def quantizer(kind):
        cfg = QuantizationConfig(spec("int8,qs=per_tensor_symmetric"), None, spec("int8,qs=per_tensor_symmetric"), None)
        cfg_e = QuantizationConfig(spec("e4m3"), None, spec("e4m3"), None)
        if kind == "global":
            return XNNPACKQuantizer().set_global(cfg)
        if kind == "object_types":
            return (XNNPACKQuantizer().set_object_type(torch.ops.aten.softmax.int, cfg)
                    .set_object_type(torch.ops.aten.layer_norm.default, cfg_e)
                    .set_object_type(torch.ops.aten.gelu.default, cfg_e)
                    .set_object_type(torch.ops.aten.relu.default, cfg))
        return XNNPACKQuantizer().set_module_name("fc1", cfg_e).set_global(cfg)     # "module_name_then_global"

    rng = np.random.default_rng(29)
    arrays, meta = {}, {}
    xs = [(rng.standard_normal((4, 8, 16)) * (i + 1)).astype(np.float32) for i in range(3)]
    for i, x in enumerate(xs):
        arrays[f"x{i}"] = f32_bits(torch.from_numpy(x))
    for kind in ("global", "object_types", "module_name_then_global"):
        m = Toy().eval()
        r = np.random.default_rng(3)
        with torch.no_grad():
            for n, p in sorted(m.named_parameters()):
                p.copy_(torch.from_numpy((r.standard_normal(tuple(p.shape)) * 0.3).astype(np.float32)))
        gm = qp.prepare_pt2e(m, quantizer(kind), (torch.from_numpy(xs[0]),))
        info = {"prepared_graph": _graph_rows(gm),
                "fq_modules": {n: mod.dtype for n, mod in gm.named_modules() if isinstance(mod, torch.ao.quantization.FakeQuantizeBase)}}
        with torch.no_grad():
            gm(torch.from_numpy(xs[0]))
            gm(torch.from_numpy(xs[1]))
            y = gm(torch.from_numpy(xs[2]))
        arrays[f"{kind}/y_prepared"] = tensor_bits(y)
        info["scales"] = {k: [float(t) for t in v.reshape(-1)] for k, v in gm.state_dict().items() if k.endswith(".scale")}
        meta[kind] = info
    np.savez_compressed(os.path.join(out, "pt2e_patterns.npz"), **{k.replace("/", "__"): v for k, v in arrays.items()})
    with open(os.path.join(out, "pt2e_patterns.json"), "w") as f:
        json.dump(meta, f, indent=1)


def gen_outlier(ref, out):
    """Outlier side path of a block-scaled linear layer: the two operators on their own (decomposed.py:450-566) and the
    converted graph + outputs of a two-layer model whose activation spec carries `outlier=` (quantize_pt2e.py:489-510,
    705-750).  The reference only handles 2-D activations here (its add of the [rows, N] side product fails on
    higher ranks), so the model input is [16, 64]."""
    import torch.nn as nn
    qp = ref.quantize_pt2e
    ops = torch.ops.quantized_ops
    rng = np.random.default_rng(31)
    arrays, meta = {}, {}
    x = torch.from_numpy((rng.standard_normal((3, 5, 32)) * 2).astype(np.float32))
    x[0, 0, 0] = float("nan")
    inl, data, idx, ptr = ops.filter_outlier(x, 3.5, 0.1)
    arrays["op/x"] = f32_bits(x)
    arrays["op/inlier"] = canon_nan32(f32_bits(inl))
    arrays["op/data"] = f32_bits(data)
    arrays["op/indices"] = idx.numpy().astype(np.int64)
    arrays["op/indptr"] = ptr.numpy().astype(np.int64)
    w = torch.from_numpy(rng.standard_normal((24, 32)).astype(np.float32))
    ws = torch.from_numpy((2.0 ** rng.integers(-3, 2, (24, 1))).astype(np.float32))
    arrays["op/w"] = f32_bits(w)
    arrays["op/ws"] = f32_bits(ws)
    arrays["op/y_plain"] = f32_bits(ops.spmm_csr(data, idx, ptr, w))
    arrays["op/y_scaled"] = f32_bits(ops.spmm_csr(data, idx, ptr, w, ws, None, 32))
    arrays["op/y_transposed"] = f32_bits(ops.spmm_csr(data, idx, ptr, w.T.contiguous(), None, None, None, True))
    meta["op"] = {"threshold": 3.5, "max_pct": 0.1, "nnz": int(ptr[-1])}

    class Toy(nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = nn.Linear(64, 128)
            self.fc2 = nn.Linear(128, 64)

        def forward(self, x):
            return self.fc2(torch.relu(self.fc1(x)))

    m = Toy().eval()
    r = np.random.default_rng(3)
    with torch.no_grad():
        for n, p in sorted(m.named_parameters()):
            p.copy_(torch.from_numpy((r.standard_normal(tuple(p.shape)) * 0.1).astype(np.float32)))
    xs = [torch.from_numpy((rng.standard_normal((16, 64)) * 2).astype(np.float32)) for _ in range(2)]
    arrays["x0"], arrays["x1"] = f32_bits(xs[0]), f32_bits(xs[1])
    kw = dict(input_activation="int8,qs=microscaling,bs=32,ax=-1,outlier=4.5", weight="int8,qs=microscaling,bs=32,ax=-1",
              force_scale_power_of_two=True)
    gm = qp.prepare_pt2e(m, qp.get_default_quantizer(**kw), (xs[0],))
    with torch.no_grad():
        gm(xs[0])
        y1 = gm(xs[1])
    arrays["y_prepared"] = tensor_bits(y1)
    gc = qp.convert_pt2e(gm)
    with torch.no_grad():
        y2 = gc(xs[1])
    arrays["y_converted"] = tensor_bits(y2)
    meta["model"] = {"kw": kw, "converted_graph": _graph_rows(gc),
                     "converted_kwargs": {n.name: {k: (v.name if isinstance(v, torch.fx.Node) else v) for k, v in n.kwargs.items()}
                                          for n in gc.graph.nodes if n.kwargs}}
    np.savez_compressed(os.path.join(out, "outlier.npz"), **{k.replace("/", "__"): v for k, v in arrays.items()})
    with open(os.path.join(out, "outlier.json"), "w") as f:
        json.dump(meta, f, indent=1)


def gen_pt2e_mx(ref, out):
    """PT2E flow with block-scaled (microscaling) specs: prepared outputs, the converted graph with
    quantize_mx / linear_mx / matmul_mx nodes, quantized weight + scale buffers and converted outputs
    (quantize_pt2e.py:94-116, 456-700; decomposed.py:265-448)."""
    import torch.nn as nn
    qp = ref.quantize_pt2e
    assert qp is not None, getattr(ref, "pt2e_error", None)

    class Toy(nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = nn.Linear(64, 128)
            self.fc2 = nn.Linear(128, 64)

        def forward(self, x):
            h = torch.relu(self.fc1(x))
            y = self.fc2(h) + x
            a = torch.matmul(y, y.transpose(-1, -2))
            return torch.matmul(torch.softmax(a, -1), y)

    rng = np.random.default_rng(23)
    arrays, meta = {}, {}
    xs = [(rng.standard_normal((2, 32, 64)) * (i + 1)).astype(np.float32) for i in range(2)]
    for i, x in enumerate(xs):
        arrays[f"x{i}"] = f32_bits(torch.from_numpy(x))
    runs = [("mxfp8", dict(input_activation="fp8_e4m3,qs=microscaling,bs=32,ax=-1", weight="fp8_e4m3,qs=microscaling,bs=32,ax=-1",
                           force_scale_power_of_two=True), torch.float32),
            ("mxfp4_bf16", dict(input_activation="fp4_e2m1,qs=microscaling,bs=32,ax=-1", weight="fp4_e2m1,qs=microscaling,bs=32,ax=-1",
                                force_scale_power_of_two=True), torch.bfloat16),
            ("mxfp6_w_int", dict(input_activation="fp6_e3m2,qs=microscaling,bs=32,ax=-1", weight="int4,qs=microscaling,bs=32,ax=-1",
                                 force_scale_power_of_two=True), torch.float32),
            ("nvfp4_like", dict(input_activation="fp4_e2m1,qs=microscaling,bs=16,ax=-1,scale=fp8_e4m3",
                                weight="fp4_e2m1,qs=microscaling,bs=16,ax=-1,scale=fp8_e4m3"), torch.float32),
            ("nf4_weight", dict(input_activation="int8,qs=microscaling,bs=32,ax=-1", weight="nf4_6,qs=microscaling,bs=32,ax=-1",
                                force_scale_power_of_two=True), torch.float32)]
    for name, kw, dt in runs:
        m = Toy().eval()
        r = np.random.default_rng(3)
        with torch.no_grad():
            for n, p in sorted(m.named_parameters()):
                p.copy_(torch.from_numpy((r.standard_normal(tuple(p.shape)) * 0.3).astype(np.float32)))
        if name == "mxfp8":
            for n, p in m.named_parameters():
                arrays["param/" + n] = f32_bits(p.detach())
        m = m.to(dt)
        x0, x1 = (torch.from_numpy(x).to(dt) for x in xs)
        try:
            q = qp.get_default_quantizer(**kw)
            gm = qp.prepare_pt2e(m, q, (x0,))
            info = {"kw": kw, "dtype": str(dt).replace("torch.", ""), "prepared_graph": _graph_rows(gm),
                    "fq_modules": {n: [mod.dtype, str(mod.ch_axis), str(mod.block_size)] for n, mod in gm.named_modules()
                                   if isinstance(mod, torch.ao.quantization.FakeQuantizeBase)}}
            with torch.no_grad():
                gm(x0)
                y1 = gm(x1)
            arrays[f"{name}/y_prepared"] = tensor_bits(y1)
            gc = qp.convert_pt2e(gm)
            info["converted_graph"] = _graph_rows(gc)
            info["converted_kwargs"] = {n.name: {k: (v.name if isinstance(v, torch.fx.Node) else v) for k, v in n.kwargs.items()}
                                        for n in gc.graph.nodes if n.kwargs}
            info["node_dtype"] = {n.name: (list(n.meta["dtype"]) if isinstance(n.meta["dtype"], tuple) else n.meta["dtype"])
                                  for n in gc.graph.nodes if n.meta.get("dtype") is not None}
            with torch.no_grad():
                y2 = gc(x1)
            arrays[f"{name}/y_converted"] = tensor_bits(y2)
            for k, v in gc.named_buffers():
                arrays[f"{name}/buf/{k}"] = tensor_bits(v)
            info["buffers"] = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in gc.named_buffers()}
        except Exception as e:  # noqa: BLE001 - record what the reference does with this spec
            info = {"kw": kw, "dtype": str(dt).replace("torch.", ""), "error": f"{type(e).__name__}: {e}"[:300]}
        meta[name] = info
    np.savez_compressed(os.path.join(out, "pt2e_mx.npz"), **{k.replace("/", "__"): v for k, v in arrays.items()})
    with open(os.path.join(out, "pt2e_mx.json"), "w") as f:
        json.dump(meta, f, indent=1)


def gen_spec(ref, out):
    """QuantizationSpec.from_str / get_quant_min_max / add_qspec_args defaults."""
    from dataclasses import asdict
    specs = ["e4m3", "int8,qs=per_tensor_symmetric", "fp8_e4m3,qs=per_tensor_symmetric,qmax=448,ahl=10",
             "posit8_1,qs=per_channel_symmetric,ax=0", "int4,qs=microscaling,bs=32,ax=-1",
             "int8,qscheme=per_tensor_symmetric,quant_max=127,amax_history_len=50,ch_axis=0,block_size=32",
             "nf4,qs=group_wise_affine,bs=(1,64),ax=(0,1),scale=fp8_e5m3", "fp4_e2m1,outlier=6.0",
             "uint8,qs=per_tensor_symmetric", "posit16_1,qs=per_tensor_symmetric", "fp6_e3m2,qs=per_tensor_symmetric"]
    res = {}
    for s in specs:
        d = asdict(ref.quantizer.QuantizationSpec.from_str(s))
        d.pop("observer_or_fake_quant_ctr")
        if d["qscheme"] is not None:
            d["qscheme"] = d["qscheme"].value
        for k in ("ch_axis", "block_size"):
            if isinstance(d[k], tuple):
                d[k] = list(d[k])
        res[s] = d
    errors = {}
    for s in ["", "int8,foo=1", "int8,qs", "int8,qs=per_tensor_symmetric,qmax=", "int4,qs=microscaling", "bogus9,qs=per_tensor_symmetric"]:
        try:
            ref.quantizer.QuantizationSpec.from_str(s)
            errors[s] = None
        except Exception as e:  # noqa: BLE001
            errors[s] = type(e).__name__
    qmm = {}
    for dt in ["int8", "int4", "uint8", "fp8_e4m3", "fp8_e5m2", "fp6_e3m2", "fp6_e2m3", "fp4_e2m1", "posit8_0",
               "posit8_1", "posit8_2", "posit16_1", "nf4", "nf4_6", "INT8"]:
        qmm[dt] = [float(v) for v in ref.quantizer.get_quant_min_max(dt)]
    p = ref.training_args.add_qspec_args()
    defaults = {k: (v if isinstance(v, (int, float, str, bool, type(None), list)) else repr(v))
                for k, v in vars(p.parse_args([])).items()}
    flags = sorted(o for a in p._actions for o in a.option_strings)
    with open(os.path.join(out, "spec.json"), "w") as f:
        json.dump({"specs": res, "errors": errors, "quant_min_max": qmm, "arg_defaults": defaults, "flags": flags},
                  f, indent=1, sort_keys=True)


def gen_windows(out):
    """Window schedule of wikitext.py:143-165 restated as data (seq_len, max_length, stride)->(begin,end,trg_len)."""
    res = {}
    for seq_len, max_length, stride in [(341469, 1024, 512), (5000, 1024, 512), (2048, 1024, 512), (1500, 1024, 512),
                                        (4096, 2048, 512), (3000, 512, 512)]:
        rows = []
        prev_end = 0
        for begin in range(0, seq_len - max_length, stride):  # upstream loop bounds
            end = min(begin + max_length, seq_len)
            trg_len = end - prev_end
            rows.append([begin, end, trg_len])
            prev_end = end
            if end == seq_len:
                break
        res[f"{seq_len},{max_length},{stride}"] = {"n": len(rows), "first": rows[:3], "last": rows[-2:],
                                                   "sum_trg": sum(r[2] for r in rows)}
    with open(os.path.join(out, "wikitext_windows.json"), "w") as f:
        json.dump(res, f, indent=1)


LORA_CASES = [
    # name, weight spec, fan_in_fan_out, rank, (in, out)
    ("lora_int8_delayed", "int8,qs=per_tensor_symmetric,ahl=3", False, 4, (48, 32)),
    ("lora_e4m3", "e4m3", False, 8, (64, 40)),
    ("lora_fp8_pow2_fanin", "fp8_e4m3,qs=per_tensor_symmetric", True, 4, (32, 48)),
]


def gen_lora(ref, out):
    """Three training steps through the LoRA QAT layer's forward (modules/qat/lora.py:34-55).  peft is not
    installed, so the layer object is assembled by hand around the reference class (its __init__ needs peft's
    base class) and only the reference's own forward runs."""
    import importlib
    from dataclasses import asdict
    import torch.nn as nn
    lora = importlib.import_module("quantized_training.modules.qat.lora")
    arrays, meta = {}, []
    for name, spec, fifo, r, (fin, fout) in LORA_CASES:
        torch.manual_seed(7)
        kw = asdict(ref.quantizer.QuantizationSpec.from_str(spec))
        layer = lora.Linear.__new__(lora.Linear)
        nn.Module.__init__(layer)
        w = torch.randn(fout, fin) * 0.1
        layer.weight = nn.Parameter((w.T.contiguous() if fifo else w).to(torch.bfloat16), requires_grad=False)
        layer.bias = nn.Parameter((torch.randn(fout) * 0.1).to(torch.bfloat16), requires_grad=False)
        a, b = nn.Linear(fin, r, bias=False), nn.Linear(r, fout, bias=False)
        with torch.no_grad():
            b.weight.copy_(torch.randn(fout, r) * 0.05)
        layer.lora_A = nn.ModuleDict({"default": a.to(torch.bfloat16)})
        layer.lora_B = nn.ModuleDict({"default": b.to(torch.bfloat16)})
        layer.scaling = {"default": 2.0}
        layer.active_adapters = ["default"]
        layer.merged = False
        layer.disable_adapters = False
        layer.fan_in_fan_out = fifo
        layer.weight_fake_quant = ref.fake_quantize.FusedAmaxObsFakeQuantize(**kw)
        arrays[f"{name}/w"] = tensor_bits(layer.weight.data)
        arrays[f"{name}/b"] = tensor_bits(layer.bias.data)
        arrays[f"{name}/A"] = tensor_bits(a.weight.data)
        arrays[f"{name}/B"] = tensor_bits(b.weight.data)
        opt = torch.optim.SGD([a.weight, b.weight], lr=0.5)
        for step in range(3):
            x = (torch.randn(5, fin) * (1.0 + step)).to(torch.bfloat16)
            y = layer(x)
            opt.zero_grad()
            y.float().square().mean().backward()
            arrays[f"{name}/{step}/x"] = tensor_bits(x)
            arrays[f"{name}/{step}/y"] = tensor_bits(y.detach())
            arrays[f"{name}/{step}/gA"] = tensor_bits(a.weight.grad)
            arrays[f"{name}/{step}/gB"] = tensor_bits(b.weight.grad)
            arrays[f"{name}/{step}/scale"] = f32_bits(layer.weight_fake_quant.scale.detach().clone().float().reshape(-1))
            opt.step()
        assert layer.weight.grad is None
        meta.append({"name": name, "spec": spec, "fan_in_fan_out": fifo, "r": r, "in": fin, "out": fout, "steps": 3,
                     "scaling": 2.0, "lr": 0.5})
    np.savez_compressed(os.path.join(out, "lora.npz"), **{k.replace("/", "__"): v for k, v in arrays.items()})
    with open(os.path.join(out, "lora.json"), "w") as f:
        json.dump(meta, f, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    torch.set_num_threads(4)
    ref = _ref_import.load_reference(with_quantize=True)
    steps = {
        "maps": lambda: gen_maps(ref, a.out),
        "direct": lambda: gen_direct(ref, a.out),
        "vmap": lambda: gen_vmap(ref, a.out),
        "qdq": lambda: gen_quant_dequant(ref, a.out),
        "fq": lambda: gen_fake_quant(ref, a.out),
        "mx": lambda: gen_mx(ref, a.out),
        "pt2e": lambda: gen_pt2e(ref, a.out),
        "pt2e_mx": lambda: gen_pt2e_mx(ref, a.out),
        "pt2e_patterns": lambda: gen_pt2e_patterns(ref, a.out),
        "outlier": lambda: gen_outlier(ref, a.out),
        "eager": lambda: gen_eager(ref, a.out),
        "lora": lambda: gen_lora(ref, a.out),
        "spec": lambda: gen_spec(ref, a.out),
        "windows": lambda: gen_windows(a.out),
    }
    import gen_golden_blocks as gb           # upstream's BERT / MobileBERT twins, QA logits, histogram / outlier, checkpoint
    steps.update({
        "blocks": lambda: gb.gen_blocks(ref, a.out),
        "qa_logits": lambda: gb.gen_qa_logits(ref, a.out),
        "fq_extra": lambda: gb.gen_fq_extra(ref, a.out),
        "checkpoint": lambda: gb.gen_checkpoint(ref, a.out),
        "posit_opts": lambda: gb.gen_posit_opts(ref, a.out),
        "attn_chain": lambda: gb.gen_attn_chain(ref, a.out),
    })
    for k, fn in steps.items():
        if a.only and k not in a.only.split(","):
            continue
        fn()
        print("generated", k)


if __name__ == "__main__":
    main()

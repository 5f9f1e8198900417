"""One-launch versions of the elementwise chains a Hugging Face LLaMA / BERT block runs between its (fake-quantized)
GEMMs.

Not part of the reference package -- its examples run HF's modeling_llama unchanged, where RMSNorm is eight torch
kernels, rotary embedding ten and SiLU * up two.  Once everything on the fake-quant path was fused these chains were
~9 of the 24 ms of a LLaMA-2-7B evaluation window, so `quantize()` swaps them for the HIP kernels of
csrc/qt_model_ops.hip when that changes nothing observable:
  * only for bf16 device tensors under torch.no_grad() (no backward is provided -- training keeps HF's code);
  * only when nothing hooks the intermediate values (an `--quantize_forward activation` hook on the SiLU, say);
  * same operation order and bf16 rounding points: SiLU * up and rotary are bit-identical to the torch chains, RMSNorm
    differs only through the summation order of its mean (isolated outputs move by one bf16 ulp).
BERT / RoBERTa blocks (the twins of modules/quantizable/attention.py) get the same treatment: `LayerNorm(dense(x) +
residual)` and the erf GELU run as one launch each, with the next Linear's stateless FP8 input fake-quantizer applied on
the way out, and query / key / value share one batched weight pass + one FP8 GEMM (fused.SiblingGroup).  LayerNorm
differs from torch's kernel only through the summation order of its mean / variance.
`QT_FUSED_MODEL_OPS=0` keeps HF's own code everywhere.
"""
import ctypes
import logging
import os

import torch

from . import _native
from .fake_quantize import FusedAmaxObsFakeQuantize, _stream_ptr

__all__ = ["apply_llama_fusions", "apply_bert_fusions", "rmsnorm", "silu_mul", "rope", "layernorm", "gelu"]


logger = logging.getLogger(__name__)
_DECLINED = set()


def _declined(what, why):
    """One log line per (rebinding, reason) when a fused forward is NOT installed, so that a slower run is explainable
    from its log (the model still runs, on Hugging Face's own code for that piece)."""
    if (what, why) not in _DECLINED:
        _DECLINED.add((what, why))
        logger.warning("quantized_training: %s keeps the Hugging Face code path: %s", what, why)


def _enabled():
    return os.environ.get("QT_FUSED_MODEL_OPS", "1") != "0"


# ---- the Hugging Face layout the rebound forwards were written against --------------------------------------------------------------
# The fused forwards below RESTATE pieces of transformers' modules (statement for statement: the image's transformers 5.x) and read
# their attributes by name.  Another transformers release may rename an attribute or change what a forward does; a restated forward
# would then compute something else without any error.  So every rebinding is preceded by a layout check -- forward parameter names
# and the attributes the restated code reads -- and a family whose layout differs keeps Hugging Face's own code, with ONE warning
# saying what differed (quantization_mappings.py:27-72 upstream pins the same classes by import).
_LLAMA_LAYOUT = {
    "LlamaRMSNorm": (["self", "hidden_states"], ("weight", "variance_epsilon")),
    "LlamaMLP": (["self", "x"], ("gate_proj", "up_proj", "down_proj", "act_fn")),
    "LlamaRotaryEmbedding": (["self", "x", "position_ids"], ()),
    "LlamaAttention": (None, ("q_proj", "k_proj", "v_proj", "o_proj", "head_dim")),
    "LlamaDecoderLayer": (None, ("input_layernorm", "post_attention_layernorm", "self_attn", "mlp")),   # (its forward: _LAYER_PARAMS below)
}
_ROPE_PARAMS = ["q", "k", "cos", "sin", "unsqueeze_dim"]


def _forward_params(cls):
    import inspect
    try:
        return list(inspect.signature(cls.forward).parameters)
    except (TypeError, ValueError):
        return None


def llama_layout_problems(model, ml):
    """What differs between `model`'s LLaMA modules and the layout the restated forwards assume ([] = nothing)."""
    problems = []
    for mod in model.modules():
        for name, (params, attrs) in _LLAMA_LAYOUT.items():
            cls = getattr(ml, name, None)
            if cls is None or not isinstance(mod, cls):
                continue
            base = cls if not getattr(type(mod), "_qt_twin", False) else type(mod).__mro__[1]
            have = _forward_params(base)
            if params is not None and have != params:
                problems.append(f"{name}.forward takes {have}, the restated one {params}")
            missing = [a for a in attrs if not hasattr(mod, a)]
            if missing:
                problems.append(f"{name} has no attribute {missing}")
    import inspect
    rope = getattr(ml, "apply_rotary_pos_emb", None)
    rope = getattr(rope, "_qt_original", rope)
    have = list(inspect.signature(rope).parameters) if rope is not None else None
    if have != _ROPE_PARAMS:
        problems.append(f"apply_rotary_pos_emb takes {have}, the routed one {_ROPE_PARAMS}")
    return sorted(set(problems))


def bert_layer_layout_problems(layer):
    """The same for one BERT / RoBERTa-shaped encoder layer (the attributes apply_bert_fusions and the fused forwards read)."""
    problems = []
    inter = layer.intermediate
    if not hasattr(inter, "intermediate_act_fn"):
        problems.append(f"{type(inter).__name__} has no attribute intermediate_act_fn")
    have = _forward_params(type(inter))
    if have != ["self", "hidden_states"]:
        problems.append(f"{type(inter).__name__}.forward takes {have}, the restated one ['self', 'hidden_states']")
    if not isinstance(getattr(inter, "dense", None), torch.nn.Module):
        problems.append(f"{type(inter).__name__}.dense is not a module")
    return problems


def _tracing(t):
    """True while torch.export / make_fx traces the caller (the module-level rotary patch is process-wide, so an export of ANOTHER model
    can pass through it): traced tensors have no storage, and a HIP launch must never end up inside an exported graph unseen."""
    from torch._subclasses.fake_tensor import FakeTensor
    is_exporting = getattr(torch.compiler, "is_exporting", None)
    return isinstance(t, FakeTensor) or (is_exporting is not None and is_exporting()) or torch._C._get_dispatch_mode(
        torch._C._TorchDispatchModeKey.PROXY) is not None


def _eligible(*tensors):
    if not _enabled() or torch.is_grad_enabled():
        return False
    return all(t.device.type == "cuda" and t.dtype == torch.bfloat16 for t in tensors) and not (tensors and _tracing(tensors[0]))


def _hooked(mod):
    return bool(mod._forward_hooks or mod._forward_pre_hooks or mod._backward_hooks or mod._backward_pre_hooks)


# ---- kernels behind tensor-level functions ------------------------------------------------------------------------
def rmsnorm(x, weight, eps):
    cols = x.shape[-1]
    x2 = x.contiguous()
    y = torch.empty_like(x2)
    _native.check(_native.lib().qt_rmsnorm_bf16(x2.data_ptr(), weight.data_ptr(), y.data_ptr(), x2.numel() // cols, cols,
                                                float(eps), _stream_ptr(x2)), "qt_rmsnorm_bf16")
    return y


def _norm_with_consumers(x2, r2, weight, eps, fqs, codes_only=False):
    """One launch for (residual add +) RMSNorm and the input fake-quantizers of ALL the Linears consuming it: returns (sum or None, y)
    where y = fq_0(result) carries the codes for every consumer (`_qt_also_done`): their hooks hand them through (fake_quantize.py)
    instead of launching a pass each over the tensor.  (Round 2 wrote one code tensor per consumer, qt_rmsnorm_consumers_bf16 -- still
    in the C ABI; with equal formats the codes are the same bytes, and sharing one tensor took 0.4 ms off the headline window.)"""
    cols = x2.shape[-1]
    n = len(fqs)
    # the consumers' formats are equal (_norm_consumer_fq checks it), so their codes are the same bytes: ONE evaluation, one code
    # tensor, shared -- each consumer's call is still handed through and counted (8 MB less to write per norm at 1024 x 4096)
    total = torch.empty_like(x2) if r2 is not None else None
    y = torch.empty_like(x2)
    y8 = torch.empty(x2.shape, dtype=torch.uint8, device=x2.device)
    f0 = fqs[0]
    if r2 is not None:
        _native.check(_native.lib().qt_add_rmsnorm_bf16(x2.data_ptr(), r2.data_ptr(), weight.data_ptr(), total.data_ptr(),
                                                        None if codes_only else y.data_ptr(), y8.data_ptr(), x2.numel() // cols, cols, float(eps),
                                                        ctypes.byref(f0._qt_format), _stream_ptr(x2)), "qt_add_rmsnorm_bf16")
    else:
        _native.check(_native.lib().qt_rmsnorm_fq8_bf16(x2.data_ptr(), weight.data_ptr(), None if codes_only else y.data_ptr(), y8.data_ptr(),
                                                        x2.numel() // cols, cols, float(eps), ctypes.byref(f0._qt_format), _stream_ptr(x2)),
                      "qt_rmsnorm_fq8_bf16")
    codes = _fp8_view(y8, f0)
    y._qt_fp8 = codes
    y._qt_fq_done_by = f0
    y._qt_also_done = [(f, codes) for f in fqs[1:]]
    y._qt_ver = y._version
    if codes_only:
        _mark_lazy(y)                                 # (model_fusions.codes_only_ok: every consumer multiplies the codes)
    return total, y


def rmsnorm_fq(x, weight, eps, fq, codes_only=False):
    """RMSNorm with `fq` applied to the result: the first consumer's input fake-quantizer, or the list of all consumers' (then one
    launch evaluates them all, _norm_with_consumers)."""
    cols = x.shape[-1]
    x2 = x.contiguous()
    if isinstance(fq, tuple) and fq and fq[0] == "map":
        got = rmsnorm_map(x2, None, weight, eps, fq[1])
        return got[1] if got is not None else rmsnorm(x2, weight, eps)
    if isinstance(fq, (list, tuple)):
        if len(fq) > 1:
            return _norm_with_consumers(x2, None, weight, eps, fq, codes_only)[1]
        fq = fq[0]
    y = torch.empty_like(x2)
    y8 = torch.empty(x2.shape, dtype=torch.uint8, device=x2.device)
    _native.check(_native.lib().qt_rmsnorm_fq8_bf16(x2.data_ptr(), weight.data_ptr(), None if codes_only else y.data_ptr(), y8.data_ptr(),
                                                    x2.numel() // cols, cols, float(eps), ctypes.byref(fq._qt_format),
                                                    _stream_ptr(x2)), "qt_rmsnorm_fq8_bf16")
    y._qt_fp8 = _fp8_view(y8, fq)
    y._qt_fq_done_by = fq
    y._qt_ver = y._version
    if codes_only:
        _mark_lazy(y)
    return y


def add_rmsnorm(x, residual, norm, fq=None, codes_only=False):
    """(bf16(x + residual), RMSNorm of that sum) in one launch; with `fq` the norm result carries the first consumer's
    fake-quant exactly as rmsnorm_fq's does (a list: all consumers', as there)."""
    cols = x.shape[-1]
    x2, r2 = x.contiguous(), residual.contiguous()
    if isinstance(fq, tuple) and fq and fq[0] == "map":
        got = rmsnorm_map(x2, r2, norm.weight, norm.variance_epsilon, fq[1])
        if got is not None:
            return got
        fq = None
    if isinstance(fq, (list, tuple)):
        if len(fq) > 1:
            return _norm_with_consumers(x2, r2, norm.weight, norm.variance_epsilon, fq, codes_only)
        fq = fq[0]
    total = torch.empty_like(x2)
    y = torch.empty_like(x2)
    y8 = torch.empty(x2.shape, dtype=torch.uint8, device=x2.device) if fq is not None else None
    codes_only = bool(codes_only and fq is not None)
    _native.check(_native.lib().qt_add_rmsnorm_bf16(
        x2.data_ptr(), r2.data_ptr(), norm.weight.data_ptr(), total.data_ptr(), None if codes_only else y.data_ptr(),
        y8.data_ptr() if y8 is not None else None, x2.numel() // cols, cols, float(norm.variance_epsilon),
        ctypes.byref(fq._qt_format) if fq is not None else None, _stream_ptr(x2)), "qt_add_rmsnorm_bf16")
    if fq is not None:
        y._qt_fp8 = _fp8_view(y8, fq)
        y._qt_fq_done_by = fq
        y._qt_ver = y._version
        if codes_only:
            _mark_lazy(y)
    return total, y


def _add_rmsnorm_or_none(x, residual, norm):
    """The residual add in front of a LlamaRMSNorm absorbed into its kernel, or None (gradients needed, hooks on the norm,
    other dtypes / devices, QT_FUSED_MODEL_OPS=0)."""
    w = getattr(norm, "weight", None)
    if (norm is None or w is None or norm.__dict__.get("_qt_hf_forward") is None or _hooked(norm) or not _eligible(x, residual, w)
            or x.shape != residual.shape or x.shape[-1] % 8 != 0 or x.shape[-1] > 16384 or x.numel() == 0 or not w.is_contiguous()):
        return None
    fq = _norm_consumer_fq(norm, allow_all=True, allow_map=True)
    lazy = fq is not None and not (isinstance(fq, tuple) and fq and fq[0] == "map") and codes_only_ok(norm.__dict__.get("_qt_consumers"), norm)
    return add_rmsnorm(x, residual, norm, fq, codes_only=lazy)


def consumer_fq_map(linear):
    """consumer_fq for stateless TABLE formats (posit, fpN, ... without `qs`): the input fake-quantizer of a QAT Linear when a producing
    kernel may apply it in its row form (FusedAmaxObsFakeQuantize.map_producer_format), else None."""
    holder = getattr(linear, "activation_pre_process", None)
    hooked_once = len(linear._forward_pre_hooks) == 1 or linear.__dict__.get("_qt_prepared")
    if holder is None or not hooked_once or "0" not in holder or len(holder) != 1 or os.environ.get("QT_FUSED_PRODUCER_MAP", "1") == "0":
        return None
    fq = holder["0"]
    if not isinstance(fq, FusedAmaxObsFakeQuantize) or not fq.stateless_map() or fq._qt_format.kind != _native.QT_FMT_LUT:
        return None
    return fq


def _mark_done(y, fqs):
    """y = fq(result) for every fake-quantizer in `fqs` (one stateless format: their calls are idempotent repeats): the first hands y
    through, the others find themselves in `_qt_also_done` (fake_quantize.py, FusedAmaxObsFakeQuantize.forward)."""
    y._qt_fq_done_by = fqs[0]
    if len(fqs) > 1:
        y._qt_also_done = [(f, None) for f in fqs[1:]]
    y._qt_ver = y._version
    return y


def rmsnorm_map(x, residual, weight, eps, fqs, sum_fq=None):
    """(sum or None, y): RMSNorm (+ the residual add in front) with the consumers' stateless table-format fake-quantizer applied in its row
    form (qt_rmsnorm_map_bf16), or None when the device map does not carry the row words."""
    pf = fqs[0].map_producer_format(x.device)
    if pf is None:
        return None
    fmt, qmap = pf
    cols = x.shape[-1]
    x2 = x.contiguous()
    r2 = residual.contiguous() if residual is not None else None
    total = torch.empty_like(x2) if r2 is not None else None
    y = torch.empty_like(x2)
    _native.check(_native.lib().qt_rmsnorm_map_bf16(x2.data_ptr(), r2.data_ptr() if r2 is not None else None, weight.data_ptr(),
                                                    total.data_ptr() if total is not None else None, y.data_ptr(), x2.numel() // cols, cols,
                                                    float(eps), ctypes.byref(fmt), qmap.data_ptr(), int(sum_fq is not None), _stream_ptr(x2)),
                  "qt_rmsnorm_map_bf16")
    if sum_fq is not None:                           # (PT2E graphs) the residual stream's fake-quantizer, same format: written through the map
        _mark_done(total, [sum_fq])
    return total, _mark_done(y, fqs)


def silu_mul_map(gate, up, fq):
    """SiLU * up with the down projection's stateless table-format input fake-quantizer in its row form, or None."""
    pf = fq.map_producer_format(gate.device)
    if pf is None:
        return None
    fmt, qmap = pf
    g, rows, cols, rs_g = _rows_view(gate)
    u, _, _, rs_u = _rows_view(up)
    y = torch.empty(gate.shape, dtype=gate.dtype, device=gate.device)
    _native.check(_native.lib().qt_silu_mul_map_bf16(g.data_ptr(), u.data_ptr(), y.data_ptr(), rows, cols, rs_g, rs_u, ctypes.byref(fmt),
                                                     qmap.data_ptr(), _stream_ptr(g)), "qt_silu_mul_map_bf16")
    return _mark_done(y, [fq])


def rope_map(q, k, cos, sin, fq_q, fq_k, inner_q=False, inner_k=False, value_job=None):
    """Rotary embedding with qk_matmul's two stateless table-format input fake-quantizers (one format) in the same pass: contiguous
    [B, H, S, D] outputs marked as done for them, or None.

    value_job = (attn, value, fq_v), set when the table-format attention core is what will consume these tensors (_rows_attention_plan):
    the launch then also carries that kernel's value pass (qt_rope_map_value) and leaves V^T with the attention module for the core's
    call (fused.attention_rows_or_none, which counts fq_v's call when it takes it)."""
    pf = fq_q.map_producer_format(q.device)
    if pf is None or fq_k.map_producer_format(q.device) is None or fq_k.dtype != fq_q.dtype:
        return None
    fmt, qmap = pf
    B, Hq, S, D = q.shape
    Hk = k.shape[1]
    q_out = torch.empty((B, Hq, S, D), dtype=q.dtype, device=q.device)
    k_out = torch.empty((B, Hk, S, D), dtype=k.dtype, device=k.device)
    if value_job is not None:
        from . import fused
        attn, value, fq_v = value_job
        vt = torch.empty((B, Hk, D, S), dtype=torch.bfloat16, device=q.device)
        wjob = _o_proj_weight_job(attn, q, fq_q, qmap)
        W, wq = (wjob[1], torch.empty_like(wjob[1])) if wjob is not None else (None, None)
        _native.check(_native.lib().qt_rope_map_value_weight(
            q.data_ptr(), k.data_ptr(), cos.data_ptr(), sin.data_ptr(), q_out.data_ptr(), k_out.data_ptr(), B, S, Hq, Hk, D, _row_stride(q),
            _row_stride(k), ctypes.byref(fmt), qmap.data_ptr(), int(bool(inner_q)), int(bool(inner_k)), value.data_ptr(), vt.data_ptr(),
            value.stride(0), value.stride(1), value.stride(2), W.data_ptr() if W is not None else None, wq.data_ptr() if wq is not None else None,
            W.numel() if W is not None else 0, _stream_ptr(q)), "qt_rope_map_value_weight")
        attn.__dict__["_qt_vt_rows"] = (fused.value_key(value), fq_v, vt)
        if wjob is not None:
            # the output projection's weight_fake_quant(W) call finds its result (valid for the very next call on this weight at this
            # version, counted there: fake_quantize.py, `_qt_pre`)
            wjob[0].__dict__["_qt_pre"] = (W.data_ptr(), W._version, wq)
        return _mark_done(q_out, [fq_q]), _mark_done(k_out, [fq_k])
    _native.check(_native.lib().qt_rope_map_bf16(q.data_ptr(), k.data_ptr(), cos.data_ptr(), sin.data_ptr(), q_out.data_ptr(), k_out.data_ptr(),
                                                 B, S, Hq, Hk, D, _row_stride(q), _row_stride(k), ctypes.byref(fmt), qmap.data_ptr(),
                                                 int(bool(inner_q)), int(bool(inner_k)), _stream_ptr(q)), "qt_rope_map_bf16")
    return _mark_done(q_out, [fq_q]), _mark_done(k_out, [fq_k])


def _norm_consumer_fq(norm, allow_all=False, allow_map=False):
    """The fake-quantizer the norm kernel may apply: every Linear fed by this norm must quantize its input with the
    same stateless format (then the first one's pass is fused here and the siblings', run on the already quantized
    tensor, reproduce it -- the formats are idempotent)."""
    consumers = norm.__dict__.get("_qt_consumers")
    if not consumers or os.environ.get("QT_FUSED_PRODUCER_FQ", "1") == "0":
        return None
    fqs = [consumer_fq(lin) for lin in consumers]
    if any(f is None for f in fqs):
        if allow_map:                                # table formats: one row-form evaluation serves every consumer (RMSNorm kernels only)
            mfqs = [consumer_fq_map(lin) for lin in consumers]
            if all(f is not None for f in mfqs) and len({f.dtype for f in mfqs}) == 1:
                return ("map", mfqs)
        return None
    f0 = fqs[0]._qt_format
    for f in fqs[1:]:
        g = f._qt_format
        if (g.kind, g.p0, g.p1, g.flo, g.fhi) != (f0.kind, f0.p0, f0.p1, f0.flo, f0.fhi):
            return None
    if allow_all and 2 <= len(fqs) <= 3 and len({id(f) for f in fqs}) == len(fqs):
        return fqs                                   # all of them in the norm's launch (_norm_with_consumers)
    return fqs[0]


def _rows_view(t):
    """(tensor, rows, cols, row stride) of a [.., cols] tensor whose rows are equally spaced in memory (contiguous, or
    column slices of a wider buffer); other layouts are made contiguous first."""
    cols = t.shape[-1]
    rows = t.numel() // cols
    if t.dim() >= 2 and t.stride(-1) == 1 and t.data_ptr() % 16 == 0:
        rs = t.stride(-2)
        ok = rs >= cols and rs % 8 == 0
        inner = 1                                        # rows spanned by one step of the dimension being checked
        for d in range(t.dim() - 2, -1, -1):
            ok = ok and (t.shape[d] == 1 or t.stride(d) == rs * inner)
            inner *= t.shape[d]
        if ok:
            return t, rows, cols, rs
    t = t.contiguous()
    return t, rows, cols, cols


def silu_mul(gate, up):
    g, rows, cols, rs_g = _rows_view(gate)
    u, _, _, rs_u = _rows_view(up)
    y = torch.empty(gate.shape, dtype=gate.dtype, device=gate.device)
    _native.check(_native.lib().qt_silu_mul_bf16(g.data_ptr(), u.data_ptr(), y.data_ptr(), rows, cols, rs_g, rs_u,
                                                 _stream_ptr(g)), "qt_silu_mul_bf16")
    return y


def consumer_fq(linear):
    """The input fake-quantizer of a QAT Linear when a producing kernel may apply it (see
    FusedAmaxObsFakeQuantize.producer_fusable), else None.  It exists only after the layer's first call."""
    holder = getattr(linear, "activation_pre_process", None)
    hooked_once = len(linear._forward_pre_hooks) == 1 or linear.__dict__.get("_qt_prepared")    # pt2e_fusion.PreparedLinear: the fake-quantizer is a graph node
    if holder is None or not hooked_once or "0" not in holder or len(holder) != 1:
        return None
    fq = holder["0"]
    if not isinstance(fq, FusedAmaxObsFakeQuantize) or not fq.producer_fusable():
        return None
    return fq


def codes_only_ok(linears, producer=None):
    """True when a producer kernel may write ONLY the FP8 codes of its fake-quantized result for these consumers (the bf16 tensor stays
    unwritten, `_qt_lazy`): every consumer is a QAT Linear whose own forward runs (it multiplies the codes, and asks
    fake_quantize.materialize_lazy for the values on every other route), reached through its single input hook, under no_grad; and
    nobody hooked the producing module (a forward hook would be handed the unwritten tensor)."""
    from . import fused
    from .modules.qat.linear import Linear as QATLinear
    if os.environ.get("QT_CODES_ONLY", "1") == "0" or torch.is_grad_enabled() or not fused.fp8_gemm_enabled() or not linears:
        return False
    if producer is not None and _hooked(producer):
        return False
    for lin in linears:
        if not isinstance(lin, QATLinear) or type(lin).forward is not QATLinear.forward or lin._forward_hooks or lin.__dict__.get("_qt_prepared"):
            return False
        fq = consumer_fq(lin)
        if fq is None or fq._forward_hooks or fq._forward_pre_hooks:
            return False
    for lin in linears:
        consumer_fq(lin).__dict__["_qt_lazy_ok"] = True          # (its hook then hands a lazy tensor through instead of decoding it)
    return True


def _mark_lazy(t):
    """t's values were not written (its FP8 codes were).  QT_LAZY_POISON=1 (tests): fill it with NaN, so that a read that bypasses
    fake_quantize.materialize_lazy cannot go unnoticed."""
    if os.environ.get("QT_LAZY_POISON", "0") == "1":
        stamped = getattr(t, "_qt_ver", None) == t._version
        t.fill_(float("nan"))
        if stamped:
            t._qt_ver = t._version                    # (the fill is not a modification of the result the hand-over describes)
    t._qt_lazy = True
    from .fake_quantize import note_lazy
    note_lazy(t)
    return t


def _fp8_view(t8, fq):
    return t8.view(torch.float8_e5m2 if fq._qt_format.p0 == 2 else torch.float8_e4m3fn)


def silu_mul_fq(gate, up, fq):
    """SiLU * up with `fq` (the down-projection's input fake-quantizer) applied in the same pass; the result is
    marked so that the hook returns it unchanged."""
    g, rows, cols, rs_g = _rows_view(gate)
    u, _, _, rs_u = _rows_view(up)
    y = torch.empty(gate.shape, dtype=gate.dtype, device=gate.device)
    y8 = torch.empty(gate.shape, dtype=torch.uint8, device=gate.device)
    _native.check(_native.lib().qt_silu_mul_fq8_bf16(g.data_ptr(), u.data_ptr(), y.data_ptr(), y8.data_ptr(), rows, cols,
                                                     rs_g, rs_u, ctypes.byref(fq._qt_format), _stream_ptr(g)),
                  "qt_silu_mul_fq8_bf16")
    y._qt_fp8 = _fp8_view(y8, fq)
    y._qt_fq_done_by = fq
    y._qt_ver = y._version
    return y


def transpose_fq(out, fq):
    """out [B, H, S, D] (what the P.V product returns) -> contiguous [B, S, H, D] with `fq` (the output projection's
    input fake-quantizer) applied in the same pass; replaces `.transpose(1, 2).contiguous()` + the hook's own pass.
    HF reshapes the result before the projection sees it, so the hand-over goes through fq.expect_prequantized."""
    B, H, S, D = out.shape
    src = out.transpose(1, 2)                                   # [B, S, H, D] view of the [B, H, S, D] buffer
    y = torch.empty((B, S, H, D), dtype=out.dtype, device=out.device)
    y8 = torch.empty((B, S, H, D), dtype=torch.uint8, device=out.device)
    _native.check(_native.lib().qt_fake_quant_rows_bf16_fp8(out.data_ptr(), y.data_ptr(), y8.data_ptr(), B, S, H, D,
                                                            src.stride(0), src.stride(1), src.stride(2),
                                                            ctypes.byref(fq._qt_format), _stream_ptr(out)),
                  "qt_fake_quant_rows_bf16_fp8")
    fq.expect_prequantized(y, _fp8_view(y8, fq))
    return y


def attention_output(module, out):
    """[B, H, S, D] attention result -> the [B, S, H, D] tensor HF's attention block expects, fused with the output
    projection's input fake-quantizer when that is a stateless FP8 one (LLaMA-style blocks with `o_proj`)."""
    proj = getattr(module, "o_proj", None)
    if (proj is not None and os.environ.get("QT_FUSED_PRODUCER_FQ", "1") != "0" and _eligible(out) and out.dim() == 4
            and out.is_contiguous() and out.shape[-1] % 8 == 0 and out.numel() > 0):
        fq = consumer_fq(proj)
        if fq is not None:
            return transpose_fq(out, fq)
    return out.transpose(1, 2).contiguous()


def _row_stride(x):
    """x [B, H, S, D] as the transposed view of a [B, S, H, D]-ordered buffer whose (b, s) rows may be wider than H * D
    (a column slice of a fused projection's output): the row stride in elements, or None if x is something else."""
    B, H, S, D = x.shape
    sb, sh, ss, sd = x.stride()
    if sd != 1 or sh != D or ss < H * D or ss % 8 or (B > 1 and sb != S * ss) or x.data_ptr() % 16:
        return None
    return ss


def rope(q, k, cos, sin):
    """q [B, Hq, S, D], k [B, Hk, S, D] as the transposed views of [B, S, H, D] buffers that HF's attention holds;
    returns tensors with the same shape and memory order, as the torch chain would."""
    B, Hq, S, D = q.shape
    Hk = k.shape[1]
    q_out = torch.empty((B, S, Hq, D), dtype=q.dtype, device=q.device)
    k_out = torch.empty((B, S, Hk, D), dtype=k.dtype, device=k.device)
    _native.check(_native.lib().qt_rope_bf16(q.data_ptr(), k.data_ptr(), cos.data_ptr(), sin.data_ptr(), q_out.data_ptr(),
                                             k_out.data_ptr(), B, S, Hq, Hk, D, _row_stride(q), _row_stride(k),
                                             _stream_ptr(q)), "qt_rope_bf16")
    return q_out.transpose(1, 2), k_out.transpose(1, 2)


def rope_fq(q, k, cos, sin, fq_q, fq_k, value_job=None):
    """Rotary embedding with qk_matmul's two input fake-quantizers applied in the same pass; outputs are contiguous
    [B, H, S, D] (the layout those hooks write) and marked as done for fq_q / fq_k.

    value_job = (attn, value, fq_v), set when the FP8 attention kernel is what will consume these tensors (_fp8_attention_plan): the
    launch then also carries that kernel's value-code pass (qt_rope_fq_value), and writes the FP8 codes ONLY -- the kernel multiplies
    codes; should anything else ask for the bf16 values after all (the fake-quantizers' hand-over is the one door to them), they are
    decoded from the codes then, exactly (fake_quantize.materialize_lazy)."""
    B, Hq, S, D = q.shape
    Hk = k.shape[1]
    q_out = torch.empty((B, Hq, S, D), dtype=q.dtype, device=q.device)
    k_out = torch.empty((B, Hk, S, D), dtype=k.dtype, device=k.device)
    q8 = torch.empty((B, Hq, S, D), dtype=torch.uint8, device=q.device)
    k8 = torch.empty((B, Hk, S, D), dtype=torch.uint8, device=k.device)
    if value_job is None:
        _native.check(_native.lib().qt_rope_fq_bf16(q.data_ptr(), k.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                                    q_out.data_ptr(), k_out.data_ptr(), q8.data_ptr(), k8.data_ptr(), B, S, Hq,
                                                    Hk, D, _row_stride(q), _row_stride(k), ctypes.byref(fq_q._qt_format),
                                                    ctypes.byref(fq_k._qt_format), _stream_ptr(q)), "qt_rope_fq_bf16")
    else:
        from . import fused
        attn, value, fq_v = value_job
        vt8 = torch.empty((B, Hk, D, S), dtype=torch.uint8, device=q.device)
        _native.check(_native.lib().qt_rope_fq_value(q.data_ptr(), k.data_ptr(), cos.data_ptr(), sin.data_ptr(), None, None, q8.data_ptr(),
                                                     k8.data_ptr(), B, S, Hq, Hk, D, _row_stride(q), _row_stride(k),
                                                     ctypes.byref(fq_q._qt_format), ctypes.byref(fq_k._qt_format), value.data_ptr(),
                                                     vt8.data_ptr(), value.stride(0), value.stride(1), value.stride(2),
                                                     ctypes.byref(fq_v._qt_format), _stream_ptr(q)), "qt_rope_fq_value")
        attn.__dict__["_qt_vt8"] = (fused.value_key(value), fq_v, vt8)
        q_out._qt_lazy = k_out._qt_lazy = True               # bf16 values not written: see the docstring
    q_out._qt_fq_done_by = fq_q
    q_out._qt_ver = q_out._version
    k_out._qt_fq_done_by = fq_k
    k_out._qt_ver = k_out._version
    q_out._qt_fp8 = _fp8_view(q8, fq_q)                 # Q.K^T can then run as an FP8 GEMM (functional_modules.py)
    q_out._qt_ver = q_out._version
    k_out._qt_fp8 = _fp8_view(k8, fq_k)
    k_out._qt_ver = k_out._version
    return q_out, k_out


# ---- BERT-style blocks -----------------------------------------------------------------------------------------------
def layernorm(x, norm, residual=None, fq=None, codes_only=False):
    """LayerNorm(x [+ residual]) in one launch.  With `fq` (the first consuming Linear's input fake-quantizer) the kernel
    also writes fq(y) as bf16 + FP8 code and leaves them for that fake-quantizer's next call; y itself stays
    unquantized because the next residual connection reads it too.  codes_only (model_fusions.codes_only_ok): fq(y) is left as codes,
    its bf16 tensor allocated but not written (`_qt_lazy`: decoded on demand, exactly)."""
    cols = x.shape[-1]
    x2 = x.contiguous()
    r2 = residual.contiguous() if residual is not None else None
    y = torch.empty_like(x2)
    if isinstance(fq, (list, tuple)):
        if len(fq) > 1:
            # every consuming Linear's input fake-quantizer in this launch (qt_layernorm_consumers_bf16): each then finds its own codes
            # (and the shared fake-quantized values) waiting for its next call
            fqs, n = fq, len(fq)
            yq = torch.empty_like(x2)
            # equal formats (checked by _norm_consumer_fq): the consumers' codes are the same bytes -- one evaluation, one tensor
            y8 = torch.empty(x2.shape, dtype=torch.uint8, device=x2.device)
            _native.check(_native.lib().qt_layernorm_bf16(
                x2.data_ptr(), r2.data_ptr() if r2 is not None else None, norm.weight.data_ptr(), norm.bias.data_ptr(), y.data_ptr(),
                None if codes_only else yq.data_ptr(), y8.data_ptr(), x2.numel() // cols, cols, float(norm.eps),
                ctypes.byref(fqs[0]._qt_format), _stream_ptr(x2)), "qt_layernorm_bf16")
            if codes_only:
                _mark_lazy(yq)
            for f in fqs:
                f.expect_prequantized(y, _fp8_view(y8, f), replacement=yq)
            return y
        fq = fq[0]
    yq = y8 = None
    if fq is not None:
        yq = torch.empty_like(x2)
        y8 = torch.empty(x2.shape, dtype=torch.uint8, device=x2.device)
    _native.check(_native.lib().qt_layernorm_bf16(
        x2.data_ptr(), r2.data_ptr() if r2 is not None else None, norm.weight.data_ptr(), norm.bias.data_ptr(), y.data_ptr(),
        yq.data_ptr() if (yq is not None and not codes_only) else None, y8.data_ptr() if y8 is not None else None, x2.numel() // cols, cols,
        float(norm.eps), ctypes.byref(fq._qt_format) if fq is not None else None, _stream_ptr(x2)), "qt_layernorm_bf16")
    if fq is not None:
        if codes_only:
            _mark_lazy(yq)
        fq.expect_prequantized(y, _fp8_view(y8, fq), replacement=yq)
    return y


def gelu(x, fq=None, codes_only=False):
    """erf-form GELU; with `fq` (the consuming Linear's input fake-quantizer) applied in the same pass (result marked).  codes_only
    (model_fusions.codes_only_ok): only the FP8 codes are written, the bf16 tensor is `_qt_lazy`."""
    x2 = x.contiguous()
    y = torch.empty_like(x2)
    y8 = torch.empty(x2.shape, dtype=torch.uint8, device=x2.device) if fq is not None else None
    lazy = bool(codes_only and fq is not None)
    _native.check(_native.lib().qt_gelu_bf16(x2.data_ptr(), None if lazy else y.data_ptr(), y8.data_ptr() if y8 is not None else None, x2.numel(),
                                             ctypes.byref(fq._qt_format) if fq is not None else None, _stream_ptr(x2)),
                  "qt_gelu_bf16")
    if fq is not None:
        y._qt_fp8 = _fp8_view(y8, fq)
        y._qt_fq_done_by = fq
        y._qt_ver = y._version
        if lazy:
            _mark_lazy(y)
    return y


def _layernorm_ok(norm, x, residual=None):
    w, b = getattr(norm, "weight", None), getattr(norm, "bias", None)
    if type(norm) is not torch.nn.LayerNorm or w is None or b is None or _hooked(norm) or len(norm.normalized_shape) != 1:
        return False
    ts = (x, w, b) if residual is None else (x, residual, w, b)
    cols = x.shape[-1]
    return (_eligible(*ts) and cols == norm.normalized_shape[0] and cols % 8 == 0 and cols <= 16384 and x.numel() > 0
            and w.is_contiguous() and b.is_contiguous() and (residual is None or residual.shape == x.shape))


def add_layernorm_or_none(block, hidden, residual):
    """`block.LayerNorm(block.residual(hidden, residual))` of a BERT-style output block in one launch, or None when the
    add is hooked (`--quantize_forward residual`), gradients are needed, or the tensors are not bf16 device tensors."""
    add = getattr(block, "residual", None)
    norm = getattr(block, "LayerNorm", None)
    if norm is None or add is None or _hooked(add) or not _layernorm_ok(norm, hidden, residual):
        return None
    return layernorm(hidden, norm, residual, _norm_consumer_fq(norm, allow_all=True), codes_only=codes_only_ok(norm.__dict__.get("_qt_consumers"), norm))


def _layernorm_forward(self, x):
    if torch.is_grad_enabled() and x.is_cuda:
        from . import train_fusions
        y = train_fusions.layernorm_or_none(self, x)      # a training step: LayerNorm + the consumers' input quantizers in one launch
        if y is not None:
            return y
    if _layernorm_ok(self, x):
        return layernorm(x, self, None, _norm_consumer_fq(self, allow_all=True), codes_only=codes_only_ok(self.__dict__.get("_qt_consumers"), self))
    return self._qt_hf_forward(x)


def _is_erf_gelu(act):
    return (type(act).__name__ == "GELUActivation" and getattr(act, "act", None) is torch.nn.functional.gelu) or \
        (type(act) is torch.nn.GELU and act.approximate == "none")


def _intermediate_forward(self, hidden_states):
    act = self.intermediate_act_fn
    if isinstance(act, torch.nn.Module) and not _hooked(act) and _is_erf_gelu(act):
        h = self.dense(hidden_states)
        if torch.is_grad_enabled() and h.is_cuda:
            from . import train_fusions
            y = train_fusions.gelu_or_none(self, h)           # a training step: GELU + the output dense's input quantizer in one launch
            if y is not None:
                return y
        if _eligible(h) and h.numel() % 8 == 0 and h.numel() > 0:
            consumer = self.__dict__.get("_qt_consumer")
            fq = consumer_fq(consumer) if consumer is not None and os.environ.get("QT_FUSED_PRODUCER_FQ", "1") != "0" else None
            return gelu(h, fq, codes_only=fq is not None and codes_only_ok([consumer], self))
        return act(h)
    return self._qt_hf_forward(hidden_states)


def _embedding_forward(self, input):
    if torch.is_grad_enabled() and input.is_cuda:
        from . import train_fusions
        y = train_fusions.embedding_or_none(self, input)     # a training step: torch's lookup, the weight gradient as two parallel launches
        if y is not None:
            return y
    return torch.nn.Embedding.forward(self, input)


def apply_bert_fusions(model):
    """Called by quantize(): BERT / RoBERTa encoder layers (anything shaped like HF's BertLayer) get the one-launch
    LayerNorm / GELU forwards and a q / k / v sibling group.  Returns the number of layers touched."""
    layers = []
    for mod in model.modules():
        att = getattr(mod, "attention", None)
        inner = getattr(att, "self", None) if att is not None else None
        if (inner is not None and all(hasattr(inner, n) for n in ("query", "key", "value"))
                and hasattr(getattr(att, "output", None), "LayerNorm") and hasattr(getattr(mod, "intermediate", None), "dense")
                and hasattr(getattr(mod, "output", None), "LayerNorm") and hasattr(mod.output, "dense")):
            layers.append(mod)
    problems = sorted({p for mod in layers for p in bert_layer_layout_problems(mod)})
    if problems:
        _declined("BERT-style launch fusions (LayerNorm / GELU forwards, q / k / v groups)", "; ".join(problems))
        return 0
    prev_norm = None
    for mod in model.modules():                                   # the embedding LayerNorm feeds the first layer
        if type(mod).__name__.endswith("Embeddings") and isinstance(getattr(mod, "LayerNorm", None), torch.nn.LayerNorm):
            prev_norm = mod.LayerNorm
            if isinstance(getattr(mod, "dropout", None), torch.nn.Dropout):
                prev_norm.__dict__["_qt_dropout_after"] = mod.dropout      # HF: embeddings = dropout(LayerNorm(...)): active, the consumers read another tensor
            if layers:
                for emb in mod.children():                        # word / position / token-type tables: their weight gradient in a training step
                    if type(emb) is torch.nn.Embedding:
                        _bind(emb, _embedding_forward)
            break
    for i, mod in enumerate(layers):
        inner = mod.attention.self
        qkv = [inner.query, inner.key, inner.value]
        if all(hasattr(l, "weight_fake_quant") for l in qkv) and "_qt_sibling_group" not in inner.query.__dict__:
            from .fused import SiblingGroup
            group = SiblingGroup(qkv)
            for lin in qkv:
                lin.__dict__["_qt_sibling_group"] = group
        inner.__dict__["_qt_out_proj"] = mod.attention.output.dense   # consumes the attention core's result (fused._attention_fp8_or_none)
        if prev_norm is not None:
            prev_norm.__dict__["_qt_consumers"] = qkv
        mod.attention.output.LayerNorm.__dict__["_qt_consumers"] = [mod.intermediate.dense]
        mod.output.LayerNorm.__dict__.pop("_qt_consumers", None)  # set by the next layer (the last one has no Linear behind it)
        mod.intermediate.__dict__["_qt_consumer"] = mod.output.dense
        _bind(mod.intermediate, _intermediate_forward)
        for norm in (prev_norm, mod.attention.output.LayerNorm, mod.output.LayerNorm):
            if type(norm) is torch.nn.LayerNorm:
                _bind(norm, _layernorm_forward)
        prev_norm = mod.output.LayerNorm
    return len(layers)


# The attention block whose forward is running (set by hooks on the converted LlamaAttention modules): HF calls the
# module-level apply_rotary_pos_emb from inside it, and the fused rotary needs that block's qk_matmul fake-quantizers.
_CURRENT_ATTN = []


def _attn_enter(module, args, kwargs):
    _CURRENT_ATTN.append(module)
    module.__dict__["_qt_cacheless"] = kwargs.get("past_key_values") is None and kwargs.get("past_key_value") is None


def _attn_exit(module, args, kwargs, output):
    if _CURRENT_ATTN and _CURRENT_ATTN[-1] is module:
        _CURRENT_ATTN.pop()


def _qk_fqs(attn, table=False):
    """(fq_q, fq_k) of attn.qk_matmul when both may be applied by the rotary kernel, else None.  table: stateless table formats (row
    form) instead of exact FP8 ones."""
    mm = getattr(attn, "qk_matmul", None)
    holder = getattr(mm, "activation_pre_process", None) if mm is not None else None
    if holder is None or len(mm._forward_pre_hooks) != 1 or mm._forward_hooks or set(holder.keys()) != {"0", "1"}:
        return None
    fq_q, fq_k = holder["0"], holder["1"]
    for f in (fq_q, fq_k):
        if not isinstance(f, FusedAmaxObsFakeQuantize):
            return None
        if table:
            if not (f.stateless_map() and f._qt_format.kind == _native.QT_FMT_LUT) or os.environ.get("QT_FUSED_PRODUCER_MAP", "1") == "0":
                return None
        elif not f.producer_fusable():
            return None
    if getattr(attn, "num_key_value_groups", 1) != 1:
        return None                  # keys are repeated (new tensors) between the rotary and qk_matmul
    return fq_q, fq_k


# ---- module-level swaps -------------------------------------------------------------------------------------------
def _rmsnorm_forward(self, hidden_states):
    pre = getattr(hidden_states, "_qt_prenormed", None)
    if pre is not None and pre[0] is self and not _hooked(self):
        return pre[1]                    # the previous block's kernel added its residual and normalised for this norm
    w = self.weight
    if (_eligible(hidden_states, w) and hidden_states.shape[-1] % 8 == 0 and hidden_states.shape[-1] <= 16384
            and hidden_states.numel() > 0 and w.is_contiguous()):
        fq = _norm_consumer_fq(self, allow_all=True, allow_map=True)
        if fq is not None:
            lazy = not (isinstance(fq, tuple) and fq and fq[0] == "map") and codes_only_ok(self.__dict__.get("_qt_consumers"), self)
            return rmsnorm_fq(hidden_states, w, self.variance_epsilon, fq, codes_only=lazy)
        return rmsnorm(hidden_states, w, self.variance_epsilon)
    return self._qt_hf_forward(hidden_states)


def _only_pre_hooks(mod):
    return not (mod._forward_hooks or mod._backward_hooks or mod._backward_pre_hooks)


def _run_pre_hooks(mod, x):
    """What Module.__call__ does with the forward pre-hooks quantize() registered (quantize.py:128-140: the input fake-quantizers):
    the possibly replaced positional input."""
    args = (x,)
    for hook_id, hook in mod._forward_pre_hooks.items():
        if hook_id in mod._forward_pre_hooks_with_kwargs:
            res = hook(mod, args, {})
            if res is not None:
                args = res[0]
        else:
            res = hook(mod, args)
            if res is not None:
                args = res if isinstance(res, tuple) else (res,)
    return args[0]


def _fused_mlp_or_none(self, x):
    """gate_proj, up_proj, SiLU * up and the down projection's input fake-quantizer as ONE launch (qt_mlp_fq8_bf16) when both
    projections are QAT Linears on the FP8 route whose fused GEMM is the measured choice for this shape.  Every fake-quant call of
    the module chain still happens and is counted once: the two input fake-quantizers run as hooks (the second one on the already
    quantized tensor, as before), the two weight fake-quantizers and the consumer's inside the kernel."""
    from . import fused
    from .fake_quantize import STATS, handover_valid
    from .modules.qat.linear import Linear as QATLinear
    gate, up, down = self.gate_proj, self.up_proj, self.down_proj
    if os.environ.get("QT_FQ8_MLP", "1") == "0" or not fused.fq8_gemm_enabled() or fused._WEIGHT_CACHE["on"]:
        return None
    if not (isinstance(gate, QATLinear) and isinstance(up, QATLinear) and _eligible(x) and _only_pre_hooks(gate) and _only_pre_hooks(up)):
        return None
    if os.environ.get("QT_FUSED_PRODUCER_FQ", "1") == "0":
        return None
    fq_out = consumer_fq(down)
    for l in (gate, up):
        f = l.weight_fake_quant
        if not (isinstance(f, FusedAmaxObsFakeQuantize) and f.fp8_exact()):
            return None
    if fq_out is None or gate.weight.shape != up.weight.shape or x.shape[-1] != gate.weight.shape[1]:
        return None
    # The route is decided before any hook runs (a hook must not run twice): x has to arrive with the FP8 code its producer
    # (the RMSNorm kernel, for its first consumer) attached.
    x8 = getattr(x, "_qt_fp8", None) if handover_valid(x) else None
    prepared = bool(self.__dict__.get("_qt_prepared"))           # pt2e_fusion.PreparedMLP: no hooks, x comes out of the fake-quantizer's node
    if x8 is None or (getattr(x, "_qt_fq_done_by", None) is None and not prepared) or not fused.fq8_route_is_fused(x8.reshape(-1, x8.shape[-1]), [gate]):
        return None
    x8f = x8.reshape(-1, x8.shape[-1])
    for l in (gate, up):
        l.weight_fake_quant._move_to(x.device)
    if not fused.mlp_route_is_one_launch(x8f, gate, up, fq_out):
        return None
    xg = _run_pre_hooks(gate, x)                               # gate's input fake-quantizer: hands the producer's result through, counted
    x8 = getattr(xg, "_qt_fp8", None) if handover_valid(xg) else None
    xu = _run_pre_hooks(up, x)                                 # up's own input fake-quantizer: same values, still computed and counted
    if x8 is None:
        return _after_hooks_unfused(self, xg, xu)
    for l in (gate, up):
        l.weight_fake_quant._move_to(x.device)
    lazy = codes_only_ok([down], self)
    got = fused.hip_mlp_fq8_or_none(x8.reshape(-1, x8.shape[-1]), gate, up, fq_out, codes_only=lazy)
    if got is None:
        return _after_hooks_unfused(self, xg, xu)
    STATS.add(gate.weight.numel())
    STATS.add(up.weight.numel())
    h, h8 = got
    y = h.reshape(*x.shape[:-1], gate.weight.shape[0])
    y._qt_fp8 = _fp8_view(h8.reshape(y.shape), fq_out)
    y._qt_fq_done_by = fq_out
    y._qt_ver = y._version
    if lazy:
        _mark_lazy(y)
    return y


def _after_hooks_unfused(self, xg, xu):
    """Both projections' input hooks have run but the fused kernel cannot be used after all: finish with the Linears' own forwards
    (no hooks again) and the one-launch SiLU * up."""
    gate = self.gate_proj.forward(xg)
    up = self.up_proj.forward(xu)
    fq = consumer_fq(self.down_proj)
    return silu_mul_fq(gate, up, fq) if fq is not None else silu_mul(gate, up)


def _mlp_forward(self, x):
    if not _hooked(self.act_fn) and getattr(self.act_fn, "__class__", None).__name__ in ("SiLU", "SiLUActivation"):
        y = _fused_mlp_or_none(self, x)
        if y is not None:
            return self.down_proj(y)
        gate = self.gate_proj(x)
        up = self.up_proj(x)
        if _eligible(gate, up) and gate.shape == up.shape and gate.shape[-1] % 8 == 0 and gate.numel() > 0:
            fq = consumer_fq(self.down_proj)
            if fq is not None and os.environ.get("QT_FUSED_PRODUCER_FQ", "1") != "0":
                return self.down_proj(silu_mul_fq(gate, up, fq))
            mfq = consumer_fq_map(self.down_proj) if os.environ.get("QT_FUSED_PRODUCER_FQ", "1") != "0" else None
            if mfq is not None:
                y = silu_mul_map(gate, up, mfq)
                if y is not None:
                    return self.down_proj(y)
            return self.down_proj(silu_mul(gate, up))
        return self.down_proj(self.act_fn(gate) * up)
    return self._qt_hf_forward(x)


_LAYER_PARAMS = ["self", "hidden_states", "attention_mask", "position_ids", "past_key_values", "use_cache",
                 "position_embeddings", "kwargs"]


def _drop_layer_handovers(layer):
    """One-shot hand-overs a block's kernels leave for each other, dropped at the end of the block whichever route ran (each holds
    10 - 56 MB per layer at 13B widths): the sibling groups' [M, sum N] products once their members have taken their slices, the
    value pass the rotary launch wrote ahead (`_qt_vt_rows`), the o projection's pre-quantized weight (`_qt_pre`)."""
    attn, mlp = getattr(layer, "self_attn", None), getattr(layer, "mlp", None)
    for owner, names in ((attn, ("q_proj", "k_proj", "v_proj", "o_proj")), (mlp, ("gate_proj", "up_proj"))):
        if owner is None:
            continue
        for name in names:
            lin = getattr(owner, name, None)
            if lin is None:
                continue
            group = lin.__dict__.get("_qt_sibling_group")
            if group is not None:
                group.stash = None
    if attn is not None:
        attn.__dict__.pop("_qt_vt_rows", None)
        o = getattr(attn, "o_proj", None)
        fq = getattr(o, "weight_fake_quant", None) if o is not None else None
        if fq is not None and not torch.is_grad_enabled():
            fq.__dict__["_qt_pre"] = None


def _decoder_layer_forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_values=None, use_cache=False,
                           position_embeddings=None, **kwargs):
    """transformers' LlamaDecoderLayer.forward, statement for statement, with each `residual + hidden_states` absorbed into
    the RMSNorm kernel that reads the sum next -- this block's post-attention norm, and the NEXT block's input norm (or the
    model's final norm), whose result travels with the returned tensor (`_qt_prenormed`) until that norm is called."""
    residual = hidden_states
    hidden_states = self.input_layernorm(hidden_states)
    hidden_states, _ = self.self_attn(hidden_states=hidden_states, attention_mask=attention_mask, position_ids=position_ids,
                                      past_key_values=past_key_values, use_cache=use_cache,
                                      position_embeddings=position_embeddings, **kwargs)
    fused = _add_rmsnorm_or_none(hidden_states, residual, self.post_attention_layernorm)
    if fused is not None:
        residual, hidden_states = fused
    else:
        hidden_states = residual + hidden_states
        residual = hidden_states
        hidden_states = self.post_attention_layernorm(hidden_states)
    hidden_states = self.mlp(hidden_states)
    _drop_layer_handovers(self)
    nxt = self.__dict__.get("_qt_next_norm")
    fused = _add_rmsnorm_or_none(hidden_states, residual, nxt) if nxt is not None else None
    if fused is None:
        return residual + hidden_states
    out, normed = fused
    out._qt_prenormed = (nxt, normed)
    return out


def _bind(module, fn):
    """Keep the class (isinstance checks, state-dict keys, hooks) and replace only this instance's forward."""
    if getattr(module, "_qt_hf_forward", None) is None:
        module._qt_hf_forward = module.forward
        module.forward = fn.__get__(module, type(module))


def _rotary_table_forward(self, x, position_ids):
    """LlamaRotaryEmbedding.forward builds cos / sin from the positions with about a dozen small torch kernels on every forward.  A
    caller that passes the SAME position tensor again (same storage, same version -- the window harness's static arange) gets the
    tensors computed for it the first time; the cached entry keeps that position tensor alive, so its address cannot come back as
    another tensor's.  Anything else (a fresh tensor per forward, as transformers builds by default) is computed as before."""
    if torch.is_grad_enabled() or not _enabled():
        return self._qt_hf_forward(x, position_ids)
    key = (position_ids.data_ptr(), position_ids._version, tuple(position_ids.shape), position_ids.dtype, x.dtype, x.device)
    hit = self.__dict__.get("_qt_table")
    if hit is not None and hit[0] == key and hit[2] is position_ids:
        return hit[1]
    out = self._qt_hf_forward(x, position_ids)
    if not (x.is_cuda and torch.cuda.is_current_stream_capturing()):     # (tensors made inside a capture belong to the graph's pool)
        self.__dict__["_qt_table"] = (key, out, position_ids)
    return out


_ROPE_PATCHED = {"done": False}


def _fp8_attention_plan(attn, q, qk_fqs):
    """(attn, value, fq_v) when the attention core of this call will be qt_attention_fp8 as far as can be told here -- the rotary
    kernel's outputs go nowhere else (no KV cache), the four fake-quantizers are stateless FP8 ones of one format, the shapes are the
    kernel's, and the value projection is the q / k / v sibling group's slice (so it exists already) -- else None."""
    if (os.environ.get("QT_ROPE_VALUE_LAUNCH", "1") == "0" or os.environ.get("QT_FP8_ATTENTION_KERNEL", "1") == "0"
            or os.environ.get("QT_FP8_ATTENTION", "1") == "0" or os.environ.get("QT_FUSED_ATTENTION", "auto") == "0"
            or not attn.__dict__.get("_qt_cacheless", False)):
        return None
    vproj = getattr(attn, "v_proj", None)
    group = vproj.__dict__.get("_qt_sibling_group") if vproj is not None else None
    av = getattr(attn, "av_matmul", None)
    holder = getattr(av, "activation_pre_process", None) if av is not None else None
    if (group is None or group.stash is None or holder is None or set(holder.keys()) != {"0", "1"} or vproj not in group.layers
            or len(av._forward_pre_hooks) != 1 or av._forward_hooks):
        return None
    fqs = (*qk_fqs, holder["0"], holder["1"])
    if not all(isinstance(f, FusedAmaxObsFakeQuantize) and f.producer_fusable() for f in fqs) or len({f._qt_format.key() for f in fqs}) != 1:
        return None
    B, H, S, D = q.shape
    Ns = [l.weight.shape[0] for l in group.layers]
    idx = group.layers.index(vproj)
    y = group.stash[1]
    if D not in (64, 128) or S % 128 != 0 or S > 1024 or B * H > 65535 or Ns[idx] != H * D or y.shape[0] != B * S or not group.stash[2][idx]:
        return None                                            # (grouped-query heads, other lengths: the ordinary order)
    off = sum(Ns[:idx])
    value = y[:, off:off + Ns[idx]].view(B, S, H, D).transpose(1, 2)
    if value.dtype != torch.bfloat16 or any(st % 8 for st in value.stride()[:3]) or value.data_ptr() % 16:
        return None
    return attn, value, holder["1"]


def _o_proj_weight_job(attn, q, fq_q, qmap):
    """(weight fake-quantizer, W) when the output projection behind this attention block will run as weight pass + library GEMM with a
    stateless table-format weight fake-quantizer of the rotary launch's format: that pass then rides in the launch (qt_rope_map_value_weight)
    -- HBM-bound work beside two latency-bound jobs -- else None."""
    from . import fused
    from .modules.qat.linear import Linear as QATLinear
    proj = getattr(attn, "o_proj", None)
    if (os.environ.get("QT_ROPE_WEIGHT_PASS", "1") == "0" or not isinstance(proj, QATLinear) or type(proj).forward is not QATLinear.forward
            or torch.is_grad_enabled() or fused._WEIGHT_CACHE["on"] or not fused._fqt_weight_ok(proj)):
        return None
    fqw, W = proj.weight_fake_quant, proj.weight
    if str(fqw.dtype) != str(fq_q.dtype) or fqw._forward_hooks or fqw._forward_pre_hooks:
        return None
    if W.dtype != torch.bfloat16 or not W.is_contiguous() or W.numel() % 8 or W.data_ptr() % 16 or W.device != q.device:
        return None
    fqw._move_to(q.device)
    if fqw.qmap is None or fqw.qmap.data_ptr() != qmap.data_ptr():             # one cached device map per dtype: the same table, or no deal
        return None
    B, H, S, D = q.shape
    if fused.fqt_gemm_mode() != "0" and fused.fqt_route_is_fused(B * S, [W.shape[0]], W.shape[1], q.device):
        return None                                                             # the value-map GEMM converts the weight itself
    return fqw, W


def _rows_attention_plan(attn, q, qk_fqs):
    """(attn, value, fq_v) when the attention core of this call will be qt_attention_rows_bf16 as far as can be told here -- the same
    conditions as _fp8_attention_plan with the four fake-quantizers stateless TABLE formats of one dtype in their row form, head_dim 128
    -- else None."""
    if (os.environ.get("QT_ROPE_VALUE_LAUNCH", "1") == "0" or os.environ.get("QT_FUSED_ATTENTION", "auto") == "0"
            or not attn.__dict__.get("_qt_cacheless", False)):
        return None
    vproj = getattr(attn, "v_proj", None)
    group = vproj.__dict__.get("_qt_sibling_group") if vproj is not None else None
    av = getattr(attn, "av_matmul", None)
    holder = getattr(av, "activation_pre_process", None) if av is not None else None
    if (group is None or group.stash is None or holder is None or set(holder.keys()) != {"0", "1"} or vproj not in group.layers
            or len(av._forward_pre_hooks) != 1 or av._forward_hooks):
        return None
    fqs = (*qk_fqs, holder["0"], holder["1"])
    for f in fqs:
        if not (isinstance(f, FusedAmaxObsFakeQuantize) and f.stateless_map() and f._qt_format.kind == _native.QT_FMT_LUT
                and f.map_producer_format(q.device) is not None) or f._forward_hooks or f._forward_pre_hooks:
            return None
    if len({str(f.dtype) for f in fqs}) != 1:
        return None
    B, H, S, D = q.shape
    Ns = [l.weight.shape[0] for l in group.layers]
    idx = group.layers.index(vproj)
    y = group.stash[1]
    if D != 128 or S % 128 != 0 or S > 1024 or B * H > 65535 or Ns[idx] != H * D or y.shape[0] != B * S or not group.stash[2][idx]:
        return None
    off = sum(Ns[:idx])
    value = y[:, off:off + Ns[idx]].view(B, S, H, D).transpose(1, 2)
    if value.dtype != torch.bfloat16 or any(st % 8 for st in value.stride()[:3]) or value.data_ptr() % 16:
        return None
    return attn, value, holder["1"]


def _patch_rope():
    """HF's LlamaAttention.forward calls the module-level apply_rotary_pos_emb; route it through the fused kernel when
    the tensors are the layout it produces ([B, S, H, D] buffers seen as [B, H, S, D], cos / sin [B, S, D])."""
    if _ROPE_PATCHED["done"]:
        return
    try:
        from transformers.models.llama import modeling_llama as ml
        original = ml.apply_rotary_pos_emb
    except Exception as e:  # noqa: BLE001
        _declined("rotary embedding", f"transformers.models.llama.modeling_llama.apply_rotary_pos_emb not found ({e})")
        return

    def apply_rotary_pos_emb(q, k, cos, sin, unsqueeze_dim=1):
        ok = (unsqueeze_dim == 1 and q.dim() == 4 and k.dim() == 4 and cos.dim() == 3 and _eligible(q, k, cos, sin)
              and q.shape[-1] % 16 == 0 and q.shape[-1] == k.shape[-1] == cos.shape[-1] and cos.shape == sin.shape
              and cos.shape[0] in (1, q.shape[0]) and cos.shape[1] == q.shape[2] and k.shape[2] == q.shape[2]
              and _row_stride(q) is not None and _row_stride(k) is not None
              and cos.is_contiguous() and sin.is_contiguous() and q.numel() > 0 and k.shape[0] == q.shape[0])
        if ok:
            if cos.shape[0] != q.shape[0]:                      # position ids shared by the batch
                cos, sin = cos.expand(q.shape[0], -1, -1).contiguous(), sin.expand(q.shape[0], -1, -1).contiguous()
            if _CURRENT_ATTN and os.environ.get("QT_FUSED_PRODUCER_FQ", "1") != "0":
                fqs = _qk_fqs(_CURRENT_ATTN[-1])
                if fqs is not None:
                    return rope_fq(q, k, cos, sin, *fqs, value_job=_fp8_attention_plan(_CURRENT_ATTN[-1], q, fqs))
                mfqs = _qk_fqs(_CURRENT_ATTN[-1], table=True)
                if mfqs is not None:
                    got = rope_map(q, k, cos, sin, *mfqs, value_job=_rows_attention_plan(_CURRENT_ATTN[-1], q, mfqs))
                    if got is not None:
                        return got
            return rope(q, k, cos, sin)
        return original(q, k, cos, sin, unsqueeze_dim)

    apply_rotary_pos_emb._qt_original = original
    ml.apply_rotary_pos_emb = apply_rotary_pos_emb
    _ROPE_PATCHED["done"] = True


def apply_llama_fusions(model):
    """Called by quantize(): swap the forwards of every LlamaRMSNorm / LlamaMLP of `model` and route rotary embedding
    through the fused kernel.  Returns the number of modules touched."""
    try:
        from transformers.models.llama import modeling_llama as ml
    except Exception:  # noqa: BLE001
        return 0
    if not any(isinstance(mod, (ml.LlamaRMSNorm, ml.LlamaMLP, ml.LlamaAttention)) for mod in model.modules()):
        return 0
    problems = llama_layout_problems(model, ml)
    if problems:
        _declined("LLaMA launch fusions (RMSNorm / MLP / rotary forwards, q / k / v groups)", "; ".join(problems))
        return 0
    n = 0
    for mod in model.modules():
        if isinstance(mod, ml.LlamaRMSNorm):
            _bind(mod, _rmsnorm_forward)
            n += 1
        elif isinstance(mod, ml.LlamaMLP):
            _bind(mod, _mlp_forward)
            n += 1
        elif isinstance(mod, ml.LlamaRotaryEmbedding):
            _bind(mod, _rotary_table_forward)
        elif isinstance(mod, ml.LlamaAttention) and not getattr(mod, "_qt_ctx_hooks", False):
            mod.register_forward_pre_hook(_attn_enter, with_kwargs=True)
            mod.register_forward_hook(_attn_exit, with_kwargs=True, always_call=True)
            mod._qt_ctx_hooks = True
    # residual adds absorbed into the norm behind them: only if this transformers version's decoder layer is the one
    # _decoder_layer_forward restates (same parameters) and the model exposes its blocks and final norm the usual way
    import inspect
    have = list(inspect.signature(ml.LlamaDecoderLayer.forward).parameters)
    same_layer = have == _LAYER_PARAMS
    if n and not same_layer:
        _declined("residual add + RMSNorm (LlamaDecoderLayer.forward)",
                  f"this transformers version's decoder layer takes {have}, the restated one {_LAYER_PARAMS}")
    for mod in model.modules():
        layers, final = getattr(mod, "layers", None), getattr(mod, "norm", None)
        if same_layer and isinstance(mod, ml.LlamaModel) and not (
                isinstance(final, ml.LlamaRMSNorm) and layers is not None and all(isinstance(l, ml.LlamaDecoderLayer) for l in layers)):
            _declined("residual add + RMSNorm (LlamaDecoderLayer.forward)",
                      "the LlamaModel's .layers / .norm are not plain LlamaDecoderLayer / LlamaRMSNorm modules")
        if (same_layer and isinstance(mod, ml.LlamaModel) and isinstance(final, ml.LlamaRMSNorm) and layers is not None
                and all(isinstance(l, ml.LlamaDecoderLayer) for l in layers)):
            n_used = mod.config.num_hidden_layers
            for i, layer in enumerate(layers[:n_used]):
                layer.__dict__["_qt_next_norm"] = layers[i + 1].input_layernorm if i + 1 < n_used else final
                _bind(layer, _decoder_layer_forward)
    # which Linears consume each norm's output (plain references kept out of the module tree)
    for mod in model.modules():
        if isinstance(mod, ml.LlamaDecoderLayer):
            att, mlp = mod.self_attn, mod.mlp
            mod.input_layernorm.__dict__["_qt_consumers"] = [att.q_proj, att.k_proj, att.v_proj]
            if all(hasattr(l, "weight_fake_quant") for l in (att.q_proj, att.k_proj, att.v_proj)):
                from .fused import SiblingGroup
                group = SiblingGroup([att.q_proj, att.k_proj, att.v_proj])
                for lin in group.layers:
                    lin.__dict__["_qt_sibling_group"] = group
            mod.post_attention_layernorm.__dict__["_qt_consumers"] = [mlp.gate_proj, mlp.up_proj]
            # gate / up as one sibling group: on the FP8 routes it measured slower -- one N = 22016 GEMM is no faster than two N = 11008
            # ones and SiLU * up then reads strided rows: 16.64 vs 16.48 ms per window; what pays there is the one-launch MLP front half,
            # qt_mlp_fq8_bf16.  On the value-map GEMM (qt_linear_fqt_ws_bf16) one launch is two full rounds of 6.75-group tiles instead
            # of two launches that each fill 84 % of the chip: 32.68 -> 32.46 ms per configs[3] window (same-box A/B,
            # tools/ab_13b_routes.py) -- the group is taken by that route only.  (Folding SiLU * up and the consumer's fake-quantizer
            # into that launch's epilogue as well was built and measured: a tie with this -- at one workgroup per CU the epilogue's
            # ~45 vector instructions per value run with the matrix pipe idle and cost what the separate HBM-bound launch costs.)
            if all(hasattr(l, "weight_fake_quant") for l in (mlp.gate_proj, mlp.up_proj)):
                from .fused import SiblingGroup
                group = SiblingGroup([mlp.gate_proj, mlp.up_proj], value_map_only=True)
                for lin in group.layers:
                    lin.__dict__["_qt_sibling_group"] = group
    if n:
        _patch_rope()
    return n

"""Quantizable twins of the Hugging Face attention / output blocks.

The reference re-implements whole HF blocks so that QK^T, score scaling, softmax, AV and the
residual add become hookable sub-modules (upstream modules/quantizable/modeling_bert.py:59-62,118,
142,148,158,185-190; modeling_mobilebert.py:52-55,80-93,117-176; modeling_llama.py:131-134,228,
244-246 -- the LLaMA twin is disabled upstream because it imports symbols removed from HF).

Here the HF modules are converted IN PLACE instead, which survives HF version drift:
  * attention blocks get the sub-modules ``qk_matmul``, ``attn_scaling``, ``softmax``,
    ``av_matmul`` (same names as upstream, so hook and state-dict keys match) and a private config
    view that routes HF's ``ALL_ATTENTION_FUNCTIONS`` dispatch to ``quantizable_attention_forward``;
  * output blocks (``LayerNorm(dense(x) + residual)``) get a ``residual`` AddFunctional and a
    forward that uses it.
Every class keeps upstream's ``from_observed(float_module)`` constructor.
"""
import torch
from torch import nn

from .functional_modules import AddFunctional, MatmulFunctional, MulFunctional

__all__ = [
    "quantizable_attention_forward", "QuantizableAttentionCore", "Fp32Softmax",
    "BertSelfAttention", "BertSelfOutput", "BertOutput",
    "MobileBertSelfAttention", "MobileBertSelfOutput", "FFNOutput", "MobileBertOutput",
    "LlamaAttention",
]

ATTN_IMPL_NAME = "qt_quantizable"


class Fp32Softmax(nn.Softmax):
    """softmax computed in fp32 and cast back (what HF's LLaMA eager attention does)."""

    def forward(self, input):
        return nn.functional.softmax(input, dim=self.dim, dtype=torch.float32).to(input.dtype)


def _repeat_kv(x, n_rep):
    if n_rep == 1:
        return x
    b, h, s, d = x.shape
    return x[:, :, None, :, :].expand(b, h, n_rep, s, d).reshape(b, h * n_rep, s, d)


def _additive_mask(mask, dtype):
    """HF builds a BOOLEAN keep-mask when the model was loaded with its default (sdpa) attention
    implementation -- always under stream capture, and in eager mode whenever the batch holds padding.
    The upstream blocks add the mask to the scores (modeling_bert.py:142-145), so turn a boolean mask
    into the additive form once per forward (every layer receives the same tensor object)."""
    if mask is None or mask.dtype != torch.bool:
        return mask
    cached = getattr(mask, "_qt_additive", None)
    if cached is not None and cached.dtype == dtype:
        return cached
    keep = mask
    if keep.dim() == 4 and keep.shape[2] > 1 and keep.stride(2) == 0:
        keep = keep[:, :, :1]                   # query-broadcast view: keep it broadcast
    add = torch.zeros(keep.shape, dtype=dtype, device=keep.device).masked_fill_(~keep, torch.finfo(dtype).min)
    mask._qt_additive = add
    return add


_CAUSAL = {}


def _causal_mask(q_len, k_len, dtype, device):
    """Additive causal mask [1, 1, q_len, k_len] (query i sees keys <= i + k_len - q_len).  HF's default (sdpa) mask
    preparation hands the attention block `attention_mask=None` for unpadded batches and expects the kernel to apply
    `is_causal` itself (`_ignore_causal_mask_sdpa`); the quantizable path adds masks to the scores like the upstream
    blocks do (modeling_llama.py:228-246 upstream), so it has to build that mask -- without it a model loaded the
    usual way would attend bidirectionally."""
    key = (q_len, k_len, dtype, str(device))
    m = _CAUSAL.get(key)
    if m is None:
        if len(_CAUSAL) > 16:
            _CAUSAL.clear()
        m = torch.full((q_len, k_len), torch.finfo(dtype).min, dtype=dtype, device=device).triu(k_len - q_len + 1)[None, None]
        _CAUSAL[key] = m
    return m


def quantizable_attention_forward(module, query, key, value, attention_mask, scaling=None, dropout=0.0, **kwargs):
    """Drop-in for HF's ``eager_attention_forward`` that goes through the module's hookable ops:
    ``av_matmul(softmax(attn_scaling(qk_matmul(q, k^T), scale) + mask), v)``."""
    from ..qat.linear import flush_forward
    flush_forward()                            # projections whose forward products are still pending (a training step: one launch for q / k / v)
    n_rep = getattr(module, "num_key_value_groups", 1)
    key = _repeat_kv(key, n_rep)
    value = _repeat_kv(value, n_rep)
    if scaling is None:
        scaling = query.size(-1) ** -0.5
    attention_mask = _additive_mask(attention_mask, query.dtype)
    causal = kwargs.get("is_causal")
    if causal is None:
        causal = getattr(module, "is_causal", False)
    if attention_mask is None and query.shape[2] > 1 and causal:
        attention_mask = _causal_mask(query.shape[2], key.shape[2], query.dtype, query.device)
    from ...fused import fused_attention_or_none, fused_scores_to_probs_or_none
    core = fused_attention_or_none(module, query, key, value, attention_mask, scaling, dropout)
    if core is not None:
        return core, None                      # probabilities are never materialised on this path
    # The fused core declined (a hooked sub-module, dropout while training, a mask it cannot address, ...).  A producer that
    # expected it may have written only the FP8 codes of q / k (model_fusions.rope_fq, `_qt_lazy`): give the tensors their
    # values BEFORE any view is taken -- a view does not carry the lazy state, and the matmul below would read unwritten memory.
    from ...fake_quantize import materialize_lazy
    materialize_lazy(query)
    materialize_lazy(key)
    if torch.is_grad_enabled() and query.is_cuda:
        from ...train_fusions import attention_or_none
        out = attention_or_none(module, query, key, value, attention_mask, scaling, dropout)     # a training step: the whole core, one launch
        if out is not None:
            return out, None
    key_t = key.transpose(2, 3)
    if getattr(key, "_qt_fq_done_by", None) is not None and getattr(key, "_qt_ver", None) == key._version:
        key_t._qt_ver = key._qt_ver                          # a view shares the version counter
        key_t._qt_fq_done_by = key._qt_fq_done_by          # fake-quant is elementwise: done for K means done for K^T
        if getattr(key, "_qt_fp8", None) is not None:
            key_t._qt_fp8_of_transpose = key._qt_fp8        # FP8 code of K itself ([B, H, S, D], contiguous)
    scores = module.qk_matmul(query, key_t)
    fused = fused_scores_to_probs_or_none(module, scores, attention_mask, scaling, dropout, value)
    if fused is not None:
        probs, out = fused
        from ...model_fusions import attention_output
        return attention_output(module, out), probs
    probs = None
    if torch.is_grad_enabled() and scores.is_cuda:
        from ...train_fusions import softmax_or_none
        probs = softmax_or_none(module, scores, attention_mask, scaling, dropout)      # a training step: scaling, mask, softmax and
        if probs is not None:                                                          # av_matmul's input quantizer in one launch
            out = module.av_matmul(probs, value)
            return out.transpose(1, 2).contiguous(), probs
    scores = module.attn_scaling(scores, scaling)
    if attention_mask is not None:
        scores = scores + attention_mask[..., : key.shape[-2]]
    probs = module.softmax(scores)
    probs = nn.functional.dropout(probs, p=dropout, training=module.training)
    out = module.av_matmul(probs, value)
    return out.transpose(1, 2).contiguous(), probs


def _register_interface():
    from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS
    if ATTN_IMPL_NAME not in ALL_ATTENTION_FUNCTIONS:
        ALL_ATTENTION_FUNCTIONS.register(ATTN_IMPL_NAME, quantizable_attention_forward)


class _AttnConfigView:
    """The module's config with ``_attn_implementation`` pinned to the quantizable interface.
    Only the attention block sees it; mask construction keeps using the model's real config."""

    _attn_implementation = ATTN_IMPL_NAME

    def __init__(self, base):
        object.__setattr__(self, "_base", base)

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "_base"), name)


class QuantizableAttentionCore:
    """Shared conversion for every HF attention block that dispatches through
    ``ALL_ATTENTION_FUNCTIONS.get_interface(self.config._attn_implementation, ...)``
    (BERT, RoBERTa, MobileBERT, LLaMA, ... in the installed transformers)."""

    softmax_cls = nn.Softmax

    @classmethod
    def from_observed(cls, other):
        assert hasattr(other, "config"), "The float module must have 'config'"
        _register_interface()
        if not hasattr(other, "qk_matmul"):
            other.qk_matmul = MatmulFunctional()
            other.av_matmul = MatmulFunctional()
            other.attn_scaling = MulFunctional()
            other.softmax = cls.softmax_cls(dim=-1)
        base = other.config._base if isinstance(other.config, _AttnConfigView) else other.config
        other.config = _AttnConfigView(base)
        return other


class BertSelfAttention(QuantizableAttentionCore):
    """BERT / RoBERTa self-attention; softmax in the model dtype (upstream modeling_bert.py:62,148)."""


class MobileBertSelfAttention(QuantizableAttentionCore):
    """upstream modeling_mobilebert.py:52-55,80-93"""


class LlamaAttention(QuantizableAttentionCore):
    """HF LLaMA upcasts the softmax to fp32; grouped-query heads are repeated before QK^T."""

    softmax_cls = Fp32Softmax


# ---- output blocks: LayerNorm(dense(x) + residual) ------------------------------------------------
def _bert_output_forward(self, hidden_states, input_tensor):
    hidden_states = self.dropout(self.dense(hidden_states))
    if torch.is_grad_enabled() and hidden_states.is_cuda:
        from ...train_fusions import add_layernorm_or_none as train_add_layernorm
        fused = train_add_layernorm(self, hidden_states, input_tensor)      # a training step: the residual add inside the LayerNorm launch
        if fused is not None:
            return fused
    from ...model_fusions import add_layernorm_or_none
    fused = add_layernorm_or_none(self, hidden_states, input_tensor)      # one launch on device under no_grad
    if fused is not None:
        return fused
    return self.LayerNorm(self.residual(hidden_states, input_tensor))


def _mobilebert_self_output_forward(self, hidden_states, residual_tensor):
    out = self.dense(hidden_states)
    if not self.use_bottleneck:
        out = self.dropout(out)
    return self.LayerNorm(self.residual(out, residual_tensor))


def _ffn_output_forward(self, hidden_states, residual_tensor):
    return self.LayerNorm(self.residual(self.dense(hidden_states), residual_tensor))


def _mobilebert_output_forward(self, intermediate_states, residual_tensor_1, residual_tensor_2):
    out = self.dense(intermediate_states)
    if not self.use_bottleneck:
        out = self.dropout(out)
        return self.LayerNorm(out + residual_tensor_1)       # not hooked upstream either (modeling_mobilebert.py:163-165)
    out = self.LayerNorm(self.residual(out, residual_tensor_1))
    return self.bottleneck(out, residual_tensor_2)


_TWIN_CACHE = {}


def _swap_forward(module, forward):
    """Re-class ``module`` to a subclass of its own HF class whose forward is ``forward``."""
    base = type(module)
    if getattr(base, "_qt_twin", False):
        return module
    # The twin's forward RESTATES the Hugging Face forward it replaces (with the residual add as a hooked module).  If this transformers
    # release's block takes other arguments, or lacks a sub-module the restated code calls, the restatement is of another block: refuse
    # loudly instead of computing something else (upstream pins the classes by copying their source, quantization_mappings.py:27-72).
    import inspect
    have, want = list(inspect.signature(base.forward).parameters), list(inspect.signature(forward).parameters)
    missing = [a for a in ("dense", "LayerNorm") if not hasattr(module, a)]
    if have != want or missing:
        raise NotImplementedError(
            f"quantized_training: {base.__module__}.{base.__name__} does not have the layout its quantizable twin restates "
            f"(forward takes {have}, the twin {want}; missing sub-modules {missing}): this transformers release is not supported for "
            f"--quantize_forward/--quantize_backprop residual")
    key = (base, forward)
    twin = _TWIN_CACHE.get(key)
    if twin is None:
        twin = type(base.__name__, (base,), {"forward": forward, "_qt_twin": True, "__module__": __name__})
        _TWIN_CACHE[key] = twin
    module.__class__ = twin
    return module


class _ResidualOutput:
    forward_fn = None

    @classmethod
    def from_observed(cls, other):
        if not hasattr(other, "residual"):
            other.residual = AddFunctional()
        return _swap_forward(other, cls.forward_fn)


class BertSelfOutput(_ResidualOutput):
    """upstream modeling_bert.py:174-190"""
    forward_fn = _bert_output_forward


class BertOutput(_ResidualOutput):
    """upstream modeling_bert.py:201-214"""
    forward_fn = _bert_output_forward


class MobileBertSelfOutput(_ResidualOutput):
    """upstream modeling_mobilebert.py:109-123"""
    forward_fn = _mobilebert_self_output_forward


class FFNOutput(_ResidualOutput):
    """upstream modeling_mobilebert.py:190-200"""
    forward_fn = _ffn_output_forward


class MobileBertOutput(_ResidualOutput):
    """upstream modeling_mobilebert.py:148-186 (its inner OutputBottleneck gets a residual too, :134-146)"""
    forward_fn = _mobilebert_output_forward

    @classmethod
    def from_observed(cls, other):
        inner = getattr(other, "bottleneck", None)
        if inner is not None and not hasattr(inner, "residual"):
            inner.residual = AddFunctional()
            _swap_forward(inner, _mobilebert_self_output_forward_bottleneck)
        return super().from_observed(other)


def _mobilebert_self_output_forward_bottleneck(self, hidden_states, residual_tensor):
    out = self.dropout(self.dense(hidden_states))
    return self.LayerNorm(self.residual(out, residual_tensor))

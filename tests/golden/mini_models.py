"""Small drivers around Hugging Face's BERT / MobileBERT building blocks, used by gen_golden.py (where upstream's
`quantize()` swaps the blocks for ITS twins, modules/quantizable/modeling_bert.py / modeling_mobilebert.py) and by the
tests (where this repo's `quantize()` converts the same blocks in place).

Why not whole HF models: the upstream twins keep the transformers-4 call signature (`self.self(hidden_states,
attention_mask, head_mask, ...)`), which the installed transformers' `BertAttention.forward` no longer uses, so a
converted `BertForQuestionAnswering` cannot run upstream here.  These drivers call the blocks the way both sides accept:
`attention.self(hidden, additive_mask)[0]`, `attention.output(context, hidden)`, ...  Nothing here is upstream code.
"""
import numpy as np
import torch
from torch import nn


def seeded_init_(module, seed, std=0.25):
    """Generator-independent parameter values (numpy), in sorted parameter-name order."""
    r = np.random.default_rng(seed)
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            v = r.standard_normal(tuple(p.shape)).astype(np.float32) * std
            if name.endswith("LayerNorm.weight") or name.endswith("norm.weight"):
                v = 1.0 + 0.1 * v
            p.copy_(torch.from_numpy(v))
    return module


def additive_mask(attention_mask, dtype):
    """[B, S] keep-mask (1 = token, 0 = padding) -> additive [B, 1, 1, S] mask as BERT builds it."""
    m = attention_mask[:, None, None, :].to(dtype)
    return (1.0 - m) * torch.finfo(dtype).min


class BertBlock(nn.Module):
    """One encoder layer out of HF parts, module names as in `BertLayer` (attention.self, attention.output, intermediate,
    output), so hook / state-dict names read like the real model's."""

    def __init__(self, config):
        super().__init__()
        from transformers.models.bert import modeling_bert as hb
        self.config = config
        self.attention = nn.Module()
        self.attention.self = hb.BertSelfAttention(config)
        self.attention.output = hb.BertSelfOutput(config)
        self.intermediate = hb.BertIntermediate(config)
        self.output = hb.BertOutput(config)

    def forward(self, hidden, mask, taps=None):
        ctx = self.attention.self(hidden, mask)[0]
        att = self.attention.output(ctx, hidden)
        inter = self.intermediate(att)
        out = self.output(inter, att)
        if taps is not None:
            taps.update(context=ctx, attention_output=att, intermediate=inter, output=out)
        return out


class QAOutput:
    def __init__(self, start_logits, end_logits):
        self.start_logits, self.end_logits = start_logits, end_logits


class TinyBertQA(nn.Module):
    """embeddings -> n x BertBlock -> qa_outputs; forward(input_ids, attention_mask) returns start / end logits like
    `BertForQuestionAnswering` (what the SQuAD evaluation loop reads, run_qa_no_trainer.py:914-959 upstream)."""

    def __init__(self, config, layers=2):
        super().__init__()
        from transformers.models.bert import modeling_bert as hb
        self.config = config
        self.embeddings = hb.BertEmbeddings(config)
        self.layer = nn.ModuleList([BertBlock(config) for _ in range(layers)])
        self.qa_outputs = nn.Linear(config.hidden_size, 2)

    def forward(self, input_ids=None, attention_mask=None, **kwargs):
        h = self.embeddings(input_ids=input_ids)
        mask = additive_mask(attention_mask, h.dtype) if attention_mask is not None else None
        for blk in self.layer:
            h = blk(h, mask)
        logits = self.qa_outputs(h)
        return QAOutput(logits[..., 0].contiguous(), logits[..., 1].contiguous())


class MobileBertBlock(nn.Module):
    """One MobileBERT layer out of HF parts (bottleneck, attention.self / .output, ffn[i].intermediate / .output,
    intermediate, output), wired like `MobileBertLayer.forward`."""

    def __init__(self, config):
        super().__init__()
        from transformers.models.mobilebert import modeling_mobilebert as hm
        self.config = config
        self.use_bottleneck = config.use_bottleneck
        if self.use_bottleneck:
            self.bottleneck = hm.Bottleneck(config)
        self.attention = nn.Module()
        self.attention.self = hm.MobileBertSelfAttention(config)
        self.attention.output = hm.MobileBertSelfOutput(config)
        self.ffn = nn.ModuleList()
        for _ in range(config.num_feedforward_networks - 1):
            f = nn.Module()
            f.intermediate = hm.MobileBertIntermediate(config)
            f.output = hm.FFNOutput(config)
            self.ffn.append(f)
        self.intermediate = hm.MobileBertIntermediate(config)
        self.output = hm.MobileBertOutput(config)

    def forward(self, hidden, mask, taps=None):
        if self.use_bottleneck:
            q, k, v, layer_input = self.bottleneck(hidden)
        else:
            q = k = v = layer_input = hidden
        ctx = self.attention.self(q, k, v, mask)[0]
        att = self.attention.output(ctx, layer_input)
        x = att
        for f in self.ffn:
            x = f.output(f.intermediate(x), x)
        inter = self.intermediate(x)
        out = self.output(inter, x, hidden)
        if taps is not None:
            taps.update(context=ctx, attention_output=att, ffn_output=x, output=out)
        return out


class TinyMobileBertQA(nn.Module):
    """MobileBERT embeddings -> n x MobileBertBlock -> qa_outputs: the shape of BASELINE.json's configs[0] (MobileBERT SQuAD
    evaluation, posit(8,1) activations, CPU), at toy size."""

    def __init__(self, config, layers=2):
        super().__init__()
        from transformers.models.mobilebert import modeling_mobilebert as hm
        self.config = config
        self.embeddings = hm.MobileBertEmbeddings(config)
        self.layer = nn.ModuleList([MobileBertBlock(config) for _ in range(layers)])
        self.qa_outputs = nn.Linear(config.hidden_size, 2)

    def forward(self, input_ids=None, attention_mask=None, **kwargs):
        h = self.embeddings(input_ids=input_ids)
        mask = additive_mask(attention_mask, h.dtype) if attention_mask is not None else None
        for blk in self.layer:
            h = blk(h, mask)
        logits = self.qa_outputs(h)
        return QAOutput(logits[..., 0].contiguous(), logits[..., 1].contiguous())


def qa_model(kind):
    """The seeded QA model of a qa_logits fixture: kind "bert" or "mobilebert"."""
    if kind == "mobilebert":
        return seeded_init_(TinyMobileBertQA(tiny_mobilebert_config()), 5, std=0.2).eval()
    return seeded_init_(TinyBertQA(tiny_bert_config()), 3, std=0.2).eval()


def tiny_bert_config():
    from transformers import BertConfig
    return BertConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, vocab_size=120,
                      max_position_embeddings=48, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)


def bert_hd64_config():
    """head_dim 64 (the width the single-launch attention kernel takes), hidden 256."""
    from transformers import BertConfig
    return BertConfig(hidden_size=256, num_hidden_layers=1, num_attention_heads=4, intermediate_size=512, vocab_size=120,
                      max_position_embeddings=48, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)


def tiny_mobilebert_config():
    from transformers import MobileBertConfig
    return MobileBertConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=96, vocab_size=120,
                            embedding_size=32, intra_bottleneck_size=32, max_position_embeddings=48, num_feedforward_networks=2,
                            hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)

"""The launch-fusion planners read Hugging Face modules by class and attribute name (model_fusions.py, train_fusions.py, the quantizable
twins).  Upstream pins those layouts by vendoring the model sources (src/quantized_training/modules/modeling_*.py,
quantization_mappings.py:27-72); this package binds to the installed transformers instead, so a layout it was not written against must
make the planners DECLINE with one warning -- never rebind a forward that restates another block.  These tests feed mutated module
trees.  Also here: the one-shot hand-over caches of the training step and the QT_TRAIN_DEBUG mask."""
import copy
import logging

import pytest
import torch

import quantized_training as qt
from quantized_training import model_fusions, train_fusions


def _args(*flags):
    return qt.add_qspec_args().parse_args(list(flags))


def _tiny_llama():
    from transformers import LlamaConfig, LlamaModel
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4, vocab_size=97,
                      max_position_embeddings=32, attn_implementation="eager")
    return LlamaModel(cfg).eval()


def _bound(model):
    return [n for n, m in model.named_modules() if getattr(m, "_qt_hf_forward", None) is not None]


def test_llama_fusions_bind_on_the_layout_they_were_written_for(caplog):
    m = _tiny_llama()
    with caplog.at_level(logging.WARNING, logger=model_fusions.__name__):
        qt.quantize(m, _args("--activation", "e4m3", "--weight", "e4m3", "--quantize_forward", "gemm"))
    assert len(_bound(m)) >= 2 * 3 + 1                      # per layer two norms + the MLP (+ the decoder layers), the final norm
    assert not [r for r in caplog.records if "keeps the Hugging Face code path" in r.getMessage() and "LLaMA launch fusions" in r.getMessage()]


@pytest.mark.parametrize("mutation", ["mlp_signature", "mlp_attribute", "norm_attribute", "rope_signature"])
def test_llama_fusions_decline_on_another_layout(mutation, monkeypatch, caplog):
    from transformers.models.llama import modeling_llama as ml
    m = _tiny_llama()
    ref_in = torch.randint(0, 97, (2, 8), generator=torch.Generator().manual_seed(1))
    if mutation == "mlp_signature":                          # another release: the MLP takes (hidden_states, gate_bias)
        def forward(self, hidden_states, gate_bias=None):
            return self.down_proj(self.act_fn(self.gate_proj(hidden_states)) * self.up_proj(hidden_states))
        monkeypatch.setattr(ml.LlamaMLP, "forward", forward)
    elif mutation == "mlp_attribute":                        # ... or fuses gate and up into one projection
        for layer in m.layers:
            mlp = layer.mlp
            mlp.gate_up_proj = mlp.up_proj
            del mlp.up_proj
            mlp.forward = (lambda self_: (lambda x: self_.down_proj(self_.act_fn(self_.gate_proj(x)) * self_.gate_up_proj(x))))(mlp)
    elif mutation == "norm_attribute":                       # ... or renames the epsilon
        for mod in m.modules():
            if isinstance(mod, ml.LlamaRMSNorm):
                mod.eps = mod.variance_epsilon
                del mod.variance_epsilon
                mod.forward = (lambda self_: (lambda h: h))(mod)
    else:                                                    # ... or passes position ids to the rotary helper (transformers 4.3x)
        orig = getattr(ml.apply_rotary_pos_emb, "_qt_original", ml.apply_rotary_pos_emb)

        def apply_rotary_pos_emb(q, k, cos, sin, position_ids=None, unsqueeze_dim=1):
            return orig(q, k, cos, sin, unsqueeze_dim)
        monkeypatch.setattr(ml, "apply_rotary_pos_emb", apply_rotary_pos_emb)
    model_fusions._DECLINED.clear()
    with caplog.at_level(logging.WARNING, logger=model_fusions.__name__):
        qt.quantize(m, _args("--activation", "e4m3", "--weight", "e4m3", "--quantize_forward", "gemm"))
        with torch.no_grad():
            out = m(ref_in, use_cache=False).last_hidden_state
        qt_again = _tiny_llama()                              # a second model: the warning is not repeated
        if mutation == "mlp_attribute":
            for layer in qt_again.layers:
                layer.mlp.gate_up_proj = layer.mlp.up_proj
                del layer.mlp.up_proj
        elif mutation == "norm_attribute":
            for mod in qt_again.modules():
                if isinstance(mod, ml.LlamaRMSNorm):
                    mod.eps = mod.variance_epsilon
                    del mod.variance_epsilon
        qt.quantize(qt_again, _args("--activation", "e4m3", "--weight", "e4m3", "--quantize_forward", "gemm"))
    assert _bound(m) == [] and _bound(qt_again) == []         # nothing rebound: Hugging Face's (here: the mutated) code runs
    hits = [r.getMessage() for r in caplog.records if "LLaMA launch fusions" in r.getMessage()]
    assert len(hits) == 1, hits
    assert torch.isfinite(out).all()
    for layer in m.layers:                                    # no sibling groups, no consumer lists planned from half a layout
        for lin in (layer.self_attn.q_proj, layer.self_attn.k_proj):
            assert "_qt_sibling_group" not in lin.__dict__
        assert "_qt_consumers" not in layer.input_layernorm.__dict__


def test_bert_fusions_decline_when_the_intermediate_block_differs(caplog):
    from transformers import BertConfig, BertModel
    torch.manual_seed(0)
    cfg = BertConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, vocab_size=99, max_position_embeddings=32)
    good, bad = BertModel(cfg).eval(), BertModel(cfg).eval()
    for layer in bad.encoder.layer:                           # another release names the activation differently
        layer.intermediate.act = layer.intermediate.intermediate_act_fn
        del layer.intermediate.intermediate_act_fn
        layer.intermediate.forward = (lambda self_: (lambda h: self_.act(self_.dense(h))))(layer.intermediate)
    model_fusions._DECLINED.clear()
    flags = _args("--activation", "e4m3", "--weight", "e4m3", "--quantize_forward", "gemm")
    with caplog.at_level(logging.WARNING, logger=model_fusions.__name__):
        qt.quantize(good, flags)
        assert not caplog.records
        qt.quantize(bad, flags)
    assert any("intermediate" in n for n in _bound(good))
    assert _bound(bad) == []
    assert len([r for r in caplog.records if "BERT-style launch fusions" in r.getMessage()]) == 1
    for layer in bad.encoder.layer:
        assert "_qt_sibling_group" not in layer.attention.self.query.__dict__
    ids = torch.randint(3, 99, (2, 8), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        assert torch.isfinite(bad(ids).last_hidden_state).all()


def test_quantizable_twin_refuses_a_block_it_does_not_restate():
    """The residual twins restate HF's output blocks; a block whose forward takes other arguments is another block."""
    from transformers import BertConfig
    from transformers.models.bert import modeling_bert as mb
    cfg = BertConfig(hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64, vocab_size=50)

    class OtherOutput(mb.BertSelfOutput):
        def forward(self, hidden_states, input_tensor, gate):              # a gated residual: not what the twin computes
            return self.LayerNorm(self.dropout(self.dense(hidden_states)) * gate + input_tensor)
    from quantized_training.modules.quantizable import BertSelfOutput
    blk = OtherOutput(cfg)
    with pytest.raises(NotImplementedError, match="does not have the layout its quantizable twin restates"):
        BertSelfOutput.from_observed(blk)
    ok = BertSelfOutput.from_observed(mb.BertSelfOutput(cfg))
    assert getattr(type(ok), "_qt_twin", False) and hasattr(ok, "residual")


def test_training_plan_declines_a_mutated_tree(caplog):
    from transformers import RobertaConfig, RobertaForSequenceClassification
    torch.manual_seed(0)
    cfg = RobertaConfig(hidden_size=64, num_hidden_layers=1, num_attention_heads=4, intermediate_size=128, vocab_size=100, max_position_embeddings=20,
                        num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = RobertaForSequenceClassification(cfg).train()
    qt.quantize(m, _args("--activation", "int8,qs=per_tensor_symmetric", "--weight", "int8,qs=per_tensor_symmetric", "--error",
                         "fp8_e5m2,qs=per_tensor_symmetric,qmax=57344,ahl=10", "--quantize_forward", "gemm", "--quantize_backprop", "gemm,residual"))
    g = torch.Generator().manual_seed(1)
    batch = {"input_ids": torch.randint(3, 100, (2, 8), generator=g), "labels": torch.randint(0, 2, (2,), generator=g)}
    m(**batch).loss.backward()                                # creates the lazily built fake-quantizers
    assert train_fusions.plan(m) >= 4                         # two output-block chains, the q / k / v inputs, single-member chains
    out = m.roberta.encoder.layer[0].output
    out.norm = out.LayerNorm                                  # a layout the planner does not know
    del out.LayerNorm
    train_fusions._PLAN_WARNED = False
    with caplog.at_level(logging.WARNING, logger=train_fusions.__name__):
        assert train_fusions.plan(m) == 0
        assert train_fusions.plan(m) == 0
    assert len([r for r in caplog.records if "launch fusions are OFF" in r.getMessage()]) == 1
    from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize
    assert not any("_qt_chain" in f.__dict__ for f in m.modules() if isinstance(f, FusedAmaxObsFakeQuantize))


def test_colsum_hand_over_never_matches_another_tensor():
    """train_fusions.put_colsum / take_colsum: an entry holds its gradient tensor, so the address cannot be reused while the
    entry exists; 64 entries evict the OLDEST, never all."""
    train_fusions._COLSUM.clear()
    g = torch.randn(4, 8)
    train_fusions.put_colsum(g, "sums of g")
    ptr = g.data_ptr()
    del g                                                     # round 5: the entry outlived its tensor and the address was handed out again
    others = [torch.randn(4, 8) for _ in range(32)]
    assert all(t.data_ptr() != ptr for t in others)           # the entry keeps the storage alive
    assert all(train_fusions.take_colsum(t) is None for t in others)
    keep = []
    for i in range(70):
        t = torch.randn(2, 8)
        keep.append(t)
        train_fusions.put_colsum(t, i)
    assert len(train_fusions._COLSUM) == 64
    assert train_fusions.take_colsum(keep[-1]) == 69 and train_fusions.take_colsum(keep[-1]) is None      # one shot
    assert train_fusions.take_colsum(keep[10]) == 10 and train_fusions.take_colsum(keep[0]) is None       # the oldest went, the others stayed
    t = keep[20]
    t.add_(1.0)                                               # another version of the same tensor is another gradient
    assert train_fusions.take_colsum(t) is None
    train_fusions._COLSUM.clear()


def test_train_debug_mask(monkeypatch):
    monkeypatch.delenv("QT_TRAIN_DEBUG", raising=False)
    assert train_fusions.enabled() and train_fusions.producers_enabled() and train_fusions.fanin_enabled()
    monkeypatch.setenv("QT_TRAIN_DEBUG", "4")
    assert train_fusions.enabled() and not train_fusions.producers_enabled() and not train_fusions.fanin_enabled()
    monkeypatch.setenv("QT_TRAIN_DEBUG", "0x10")
    assert train_fusions.producers_enabled() and not train_fusions.fanin_enabled()
    monkeypatch.setenv("QT_TRAIN_DEBUG", "1")
    assert not train_fusions.enabled() and not train_fusions.producers_enabled()
    monkeypatch.setenv("QT_TRAIN_DEBUG", "chains")
    with pytest.raises(ValueError, match="integer mask"):
        train_fusions.enabled()

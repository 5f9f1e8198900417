"""quantize(model, args) on Hugging Face model families built from tiny configs (no checkpoints):
the in-place quantizable twins expose the same hookable sub-modules and hook names as the reference's
re-implemented blocks (upstream modules/quantizable/modeling_bert.py:59-62, modeling_mobilebert.py:52-55,
134-176) and the converted models still compute the float model when nothing is quantized."""
import pytest
import torch

import quantized_training as qt
from quantized_training.modules.quantizable import AddFunctional, MatmulFunctional, MulFunctional

transformers = pytest.importorskip("transformers")


def _args(*flags):
    return qt.add_qspec_args().parse_args(list(flags))


def _bert():
    from transformers import BertConfig, BertForQuestionAnswering
    torch.manual_seed(0)
    cfg = BertConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, vocab_size=100,
                     max_position_embeddings=64)
    return BertForQuestionAnswering(cfg).eval()


def _roberta():
    from transformers import RobertaConfig, RobertaForSequenceClassification
    torch.manual_seed(0)
    cfg = RobertaConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, vocab_size=100,
                        max_position_embeddings=66, num_labels=2)
    return RobertaForSequenceClassification(cfg).eval()


def _mobilebert():
    from transformers import MobileBertConfig, MobileBertForQuestionAnswering
    torch.manual_seed(0)
    cfg = MobileBertConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                           vocab_size=100, embedding_size=32, intra_bottleneck_size=32, max_position_embeddings=64)
    return MobileBertForQuestionAnswering(cfg).eval()


IDS = torch.randint(3, 100, (2, 16), generator=torch.Generator().manual_seed(0))


def _first_logits(model):
    with torch.no_grad():
        out = model(IDS)
    return out.start_logits if hasattr(out, "start_logits") else out.logits


@pytest.mark.parametrize("build,prefix", [(_bert, "bert.encoder.layer.0."), (_roberta, "roberta.encoder.layer.0.")])
def test_bert_family_twins_and_hook_names(build, prefix):
    m = build()
    ref = _first_logits(m)
    qt.quantize(m, _args("--activation", "posit8_1", "--weight", "posit8_1",
                         "--quantize_forward", "gemm,residual,activation,layernorm,scaling"))
    out = _first_logits(m)
    assert torch.isfinite(out).all() and 0 < float((out - ref).abs().max()) < 0.5 * float(ref.abs().max()) + 0.5
    mods = dict(m.named_modules())
    att = mods[prefix + "attention.self"]
    assert isinstance(att.qk_matmul, MatmulFunctional) and isinstance(att.av_matmul, MatmulFunctional)
    assert isinstance(att.attn_scaling, MulFunctional) and isinstance(att.softmax, torch.nn.Softmax)
    assert isinstance(mods[prefix + "attention.output"].residual, AddFunctional)
    assert isinstance(mods[prefix + "output"].residual, AddFunctional)
    expected = [
        "attention.self.query.activation_pre_process.0", "attention.self.key.activation_pre_process.0",
        "attention.self.value.activation_pre_process.0", "attention.self.qk_matmul.activation_pre_process.0",
        "attention.self.qk_matmul.activation_pre_process.1", "attention.self.attn_scaling.activation_pre_process.0",
        "attention.self.softmax.activation_pre_process.0", "attention.self.av_matmul.activation_pre_process.0",
        "attention.self.av_matmul.activation_pre_process.1", "attention.output.dense.activation_pre_process.0",
        "attention.output.residual.activation_pre_process.0", "attention.output.residual.activation_pre_process.1",
        "attention.output.LayerNorm.activation_pre_process.0", "intermediate.dense.activation_pre_process.0",
        "output.dense.activation_pre_process.0", "output.residual.activation_pre_process.0",
        "output.residual.activation_pre_process.1", "output.LayerNorm.activation_pre_process.0",
    ]
    for e in expected:
        assert prefix + e in mods, e
    sd = m.state_dict()
    assert prefix + "attention.self.query.weight_fake_quant.scale" in sd
    assert prefix + "attention.self.qk_matmul.activation_pre_process.1.amax_history" in sd


def test_mobilebert_twins():
    m = _mobilebert()
    ref = _first_logits(m)
    qt.quantize(m, _args("--activation", "posit8_1", "--quantize_forward", "gemm,residual,scaling"))
    out = _first_logits(m)
    assert torch.isfinite(out).all() and float((out - ref).abs().max()) > 0
    mods = dict(m.named_modules())
    p = "mobilebert.encoder.layer.0."
    for e in ["attention.self.qk_matmul.activation_pre_process.1", "attention.self.attn_scaling.activation_pre_process.0",
              "attention.output.residual.activation_pre_process.1", "output.bottleneck.residual.activation_pre_process.0",
              "output.residual.activation_pre_process.1", "ffn.0.output.residual.activation_pre_process.0",
              "bottleneck.input.dense.activation_pre_process.0"]:
        assert p + e in mods, e


def test_conversion_alone_preserves_the_float_model():
    """Twins without any fake-quantizer (weights only spec = identity on activations) compute exactly the HF model."""
    from quantized_training.quantization_mappings import TRANSFORMER_MODULE_MAPPINGS
    for build in (_bert, _mobilebert, _roberta):
        m = build()
        ref = _first_logits(m)
        qt.propagate_config(m, "config", m.config)
        qt.convert(m, inplace=True, custom_module_class_mapping=TRANSFORMER_MODULE_MAPPINGS)
        assert torch.allclose(_first_logits(m), ref, atol=1e-6, rtol=1e-5)


def test_padded_batch_with_the_default_attention_implementation():
    """A model built with HF's default attention implementation hands the blocks a BOOLEAN keep-mask when the batch
    holds padding; the twins must treat it as the additive mask the upstream blocks add to the scores."""
    from quantized_training.quantization_mappings import TRANSFORMER_MODULE_MAPPINGS
    att = torch.ones_like(IDS)
    att[0, 9:] = 0
    att[1, 13:] = 0
    for build in (_bert, _roberta):
        m = build()
        with torch.no_grad():
            ref = m(IDS, attention_mask=att)
        ref = ref.start_logits if hasattr(ref, "start_logits") else ref.logits
        qt.propagate_config(m, "config", m.config)
        qt.convert(m, inplace=True, custom_module_class_mapping=TRANSFORMER_MODULE_MAPPINGS)
        with torch.no_grad():
            out = m(IDS, attention_mask=att)
        out = out.start_logits if hasattr(out, "start_logits") else out.logits
        assert torch.allclose(out, ref, atol=1e-5, rtol=1e-5), (out - ref).abs().max()


def test_op_fusion_skips_named_modules_and_weights_only():
    m = _bert()
    qt.quantize(m, _args("--activation", "e4m3", "--weight", "e4m3", "--quantize_forward", "gemm,residual",
                         "--op_fusion", "attention.output.residual,intermediate"))
    _first_logits(m)
    mods = dict(m.named_modules())
    assert "bert.encoder.layer.0.attention.output.residual.activation_pre_process" not in mods
    assert "bert.encoder.layer.0.intermediate.dense.activation_pre_process" not in mods
    assert "bert.encoder.layer.0.output.residual.activation_pre_process.0" in mods
    m = _bert()
    qt.quantize(m, _args("--weight", "int8,qs=per_tensor_symmetric"))          # activation None -> no forward hooks
    _first_logits(m)
    assert not any("activation_pre_process" in n for n, _ in m.named_modules())
    assert float(m.bert.encoder.layer[0].attention.self.query.weight_fake_quant.amax_history[0]) > 0


def test_llama_attention_twin_and_training_backward_hooks():
    from quantized_training import harness
    m = harness.build_causal_lm("llama-tiny", device="cpu", dtype=torch.float32, seed=1)
    qt.quantize(m, _args("--activation", "int8,qs=per_tensor_symmetric", "--weight", "int8,qs=per_tensor_symmetric",
                         "--error", "fp8_e5m2,qs=per_tensor_symmetric,qmax=57344,ahl=10",
                         "--quantize_forward", "gemm", "--quantize_backprop", "gemm"))
    m.train()
    ids = torch.randint(0, 512, (2, 24), generator=torch.Generator().manual_seed(0))
    # the first backward runs every error fake-quantizer at scale 1 (delayed scaling), which flushes tiny
    # gradients; drivers always do a warm-up fwd+bwd after quantize(), so check the second
    for _ in range(2):
        m.zero_grad()
        loss = m(ids, labels=ids).loss
        loss.backward()
    mods = dict(m.named_modules())
    assert "model.layers.0.self_attn.qk_matmul.activation_pre_process.1" in mods
    assert "model.layers.0.self_attn.q_proj.error_pre_process.0" in mods
    assert "model.layers.0.self_attn.av_matmul.error_pre_process.0" in mods
    g = m.model.layers[0].self_attn.q_proj.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
    assert float(mods["model.layers.0.self_attn.q_proj.error_pre_process.0"].amax_history[0]) > 0


def test_llama_default_attention_stays_causal_after_quantize():
    """A LLaMA built the usual way (HF's default attention implementation, sdpa) hands the attention block
    `attention_mask=None` for unpadded batches and relies on the kernel's `is_causal`; the quantizable path must apply
    the causal mask itself.  With nothing quantized the converted model equals the float model, and logits at position
    t do not move when later tokens change."""
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      vocab_size=100, max_position_embeddings=64)
    m = LlamaForCausalLM(cfg).eval()                      # no attn_implementation="eager"
    ids = torch.randint(3, 100, (2, 12), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = m(ids).logits
    qt.quantize(m, _args())
    with torch.no_grad():
        out = m(ids).logits
        ids2 = ids.clone()
        ids2[:, -1] = (ids2[:, -1] + 7) % 97 + 3
        out2 = m(ids2).logits
    assert float((out - ref).abs().max()) < 1e-5
    assert float((out2[:, :-1] - out[:, :-1]).abs().max()) < 1e-6
    assert float((out2[:, -1] - out[:, -1]).abs().max()) > 1e-4


def test_declined_llama_rebinding_is_logged_once(monkeypatch, caplog):
    """When this transformers version's decoder layer is not the one model_fusions restates, the residual-add + RMSNorm
    rebinding is skipped: one warning says so (once per reason), the other rebindings still happen and the model runs."""
    import logging
    from transformers import LlamaConfig, LlamaForCausalLM
    from quantized_training import model_fusions as mf
    monkeypatch.setattr(mf, "_LAYER_PARAMS", ["self", "hidden_states", "something_else"])
    mf._DECLINED.clear()
    cfg = LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                      vocab_size=100, max_position_embeddings=64)
    m = LlamaForCausalLM(cfg).eval()
    with caplog.at_level(logging.WARNING, logger=mf.logger.name):
        assert mf.apply_llama_fusions(m) > 0
        assert mf.apply_llama_fusions(m) > 0              # a second conversion does not repeat the line
    lines = [r.getMessage() for r in caplog.records if "keeps the Hugging Face code path" in r.getMessage()]
    assert len(lines) == 1 and "LlamaDecoderLayer.forward" in lines[0] and "something_else" in lines[0]
    assert not hasattr(m.model.layers[0], "_qt_hf_forward")          # decoder layer untouched
    assert hasattr(m.model.layers[0].mlp, "_qt_hf_forward")          # the MLP rebinding still made
    with torch.no_grad():
        assert torch.isfinite(m(torch.randint(3, 100, (1, 8))).logits).all()

"""Diagnostic: the pytest sequence (h256 model first, then the full-width layer) -- is shared scratch left dirty?"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import quantized_training as qt
from quantized_training import harness, train_fusions
from test_gpu_models import _args, _TRAIN_FLAGS
from transformers import RobertaConfig, RobertaForSequenceClassification

flags = _args(*_TRAIN_FLAGS)


def scratch_state(tag):
    dirty = {str(k): int(v.count_nonzero()) for k, v in train_fusions._SCRATCH.items()}
    print(f"   [{tag}] chain scratch nonzero bytes {dirty}; _COLSUM {len(train_fusions._COLSUM)} _PENDING {len(train_fusions._PENDING)}", flush=True)


def case(cfg, B, S, V, nrep):
    torch.manual_seed(0)
    base = RobertaForSequenceClassification(cfg).bfloat16()
    g = torch.Generator().manual_seed(1)
    batches = [{"input_ids": torch.randint(3, V, (B, S), generator=g).cuda(), "labels": torch.randint(0, 2, (B,), generator=g).cuda()} for _ in range(6)]

    def run(mode):
        m = copy.deepcopy(base).cuda()
        qt.quantize(m, flags)
        opt = torch.optim.AdamW(m.parameters(), lr=2e-5, fused=True, capturable=True)
        m.train()
        if mode == "eager":
            for i, b in enumerate([batches[0]] * 3 + batches[1:]):
                if i == 1:
                    train_fusions.ensure_planned(m)
                opt.zero_grad(set_to_none=True)
                loss = m(**b).loss
                loss.backward()
                torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0, error_if_nonfinite=False)
                opt.step()
        else:
            step = harness.GraphedTrainStep(m, opt)
            step.capture(batches[0], warmup=3)
            for b in batches[1:]:
                step.replay(b)
        torch.cuda.synchronize()
        return {n: p.detach().clone() for n, p in m.named_parameters()}
    for rep in range(nrep):
        e = run("eager")
        scratch_state("after eager")
        gr = run("graph")
        scratch_state("after graph")
        bad = [k for k in e if not torch.equal(e[k], gr[k])]
        print(f"hidden {cfg.hidden_size} rep {rep}: parameters differing {len(bad)} {bad[:4]}", flush=True)


small = RobertaConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512, vocab_size=500, max_position_embeddings=70,
                      num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
big = RobertaConfig(hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=3072, vocab_size=1000, max_position_embeddings=132,
                    num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
toy = RobertaConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, vocab_size=100, max_position_embeddings=66,
                    num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
if os.environ.get("WITH_TOY", "1") == "1":
    case(toy, 8, 16, 100, 1)
case(small, 8, 64, 500, 1)
case(big, 16, 128, 1000, 3)

"""Diagnostic: eager vs graph, one step after warm-up: which gradients differ, how many bias gradients were handed over (colsums)?"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import quantized_training as qt
from quantized_training import harness, train_fusions
from test_gpu_models import _args, _TRAIN_FLAGS
from transformers import RobertaConfig, RobertaForSequenceClassification

torch.manual_seed(0)
cfg = RobertaConfig(hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=3072, vocab_size=1000,
                    max_position_embeddings=132, num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
B, S, V = 16, 128, 1000
base = RobertaForSequenceClassification(cfg).bfloat16()
g = torch.Generator().manual_seed(1)
batches = [{"input_ids": torch.randint(3, V, (B, S), generator=g).cuda(), "labels": torch.randint(0, 2, (B,), generator=g).cuda()} for _ in range(6)]
flags = _args(*_TRAIN_FLAGS)


class Probe(harness.GraphedTrainStep):
    def _step(self, batch):
        train_fusions.STATS.reset()
        return super()._step(batch)


def run(mode):
    m = copy.deepcopy(base).cuda()
    qt.quantize(m, flags)
    opt = torch.optim.AdamW(m.parameters(), lr=2e-5, fused=True, capturable=True)
    m.train()
    T = train_fusions.STATS
    if mode == "eager":
        for i, b in enumerate([batches[0]] * 3 + batches[1:2]):
            if i == 1:
                train_fusions.ensure_planned(m)
            opt.zero_grad(set_to_none=True)
            T.reset()
            loss = m(**b).loss
            loss.backward()
            stats = (T.colsums, T.chains, T.misses, T.fanins, T.deferred, len(train_fusions._COLSUM), len(train_fusions._PENDING))
            torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0, error_if_nonfinite=False)
            opt.step()
    else:
        step = Probe(m, opt)
        step.capture(batches[0], warmup=3)
        stats = (T.colsums, T.chains, T.misses, T.fanins, T.deferred, len(train_fusions._COLSUM), len(train_fusions._PENDING))
        step.replay(batches[1])
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None}
    return grads, stats


for gemm in ("1", "0"):
    os.environ["QT_TRAIN_GEMM"] = gemm
    for rep in range(3):
        e, es = run("eager")
        gr, gs = run("graph")
        bad = [(k, float((e[k] - gr[k]).abs().max()), float(e[k].abs().max())) for k in e if not torch.equal(e[k], gr[k])]
        print(f"QT_TRAIN_GEMM={gemm} rep {rep}: eager stats {es} graph stats {gs}; gradients differing {len(bad)}: {bad[:6]}", flush=True)
os.environ["QT_TRAIN_GEMM"] = "1"
os.environ["QT_TRAIN_DEBUG"] = "2"
e2, _ = run("eager")
os.environ["QT_TRAIN_DEBUG"] = "0"
e, _ = run("eager")
gr, _ = run("graph")
for k in ("roberta.encoder.layer.0.intermediate.dense.bias", "roberta.encoder.layer.0.output.dense.bias"):
    print(k, "eager == no-colsum eager:", torch.equal(e[k], e2[k]), " graph == no-colsum eager:", torch.equal(gr[k], e2[k]), " eager == graph:", torch.equal(e[k], gr[k]))

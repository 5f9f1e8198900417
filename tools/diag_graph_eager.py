"""Diagnostic: eager loop vs GraphedTrainStep on one full-width RoBERTa layer -- which fake-quantizers' state differs, and is either mode
run-to-run deterministic?"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import quantized_training as qt
from quantized_training import harness, train_fusions
from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize
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


def run(mode, nreplay=5, batched=True):
    torch.manual_seed(4321)
    m = copy.deepcopy(base).cuda()
    qt.quantize(m, flags)
    opt = torch.optim.AdamW(m.parameters(), lr=2e-5, fused=True, capturable=True)
    m.train()
    losses = []
    if mode == "eager":
        for i, b in enumerate([batches[0]] * 3 + batches[1:1 + nreplay]):
            if i == 1:
                train_fusions.ensure_planned(m)
            opt.zero_grad(set_to_none=True)
            loss = m(**b).loss
            loss.backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0, error_if_nonfinite=False)
            opt.step()
            losses.append(float(loss.detach()))
        losses = losses[3:]
    else:
        step = harness.GraphedTrainStep(m, opt, batch_scale_updates=batched, batch_weight_passes=batched)
        step.capture(batches[0], warmup=3)
        for b in batches[1:1 + nreplay]:
            losses.append(float(step.replay(b)))
    state = {n: (mod.scale.detach().float().cpu().reshape(-1), mod.amax_history.detach().float().cpu().reshape(-1)) for n, mod in m.named_modules()
             if isinstance(mod, FusedAmaxObsFakeQuantize)}
    return losses, state


def diff(a, b, what):
    bad = [k for k in a[1] if not (torch.equal(a[1][k][0], b[1][k][0]) and torch.equal(a[1][k][1], b[1][k][1]))]
    print(f"== {what}: losses equal {a[0] == b[0]}; {len(bad)} of {len(a[1])} fake-quantizers differ")
    for k in bad[:16]:
        print(f"   {k}\n      scale {a[1][k][0].tolist()} | {b[1][k][0].tolist()}\n      hist  {[f'{v:.6g}' for v in a[1][k][1].tolist()[:6]]}\n            {[f'{v:.6g}' for v in b[1][k][1].tolist()[:6]]}")


for n in (1, 5):
    e1, e2 = run("eager", n), run("eager", n)
    diff(e1, e2, f"eager vs eager, {n} steps after warm-up")
    g1, g2 = run("graph", n), run("graph", n)
    diff(g1, g2, f"graph vs graph, {n} replays")
    diff(e1, g1, f"eager vs graph, {n}")
    g3 = run("graph", n, batched=False)
    diff(e1, g3, f"eager vs graph without batched scale updates / weight passes, {n}")
for mask, name in ((8, "no fused attention"), (16, "no fan-in"), (2, "no colsum")):
    os.environ["QT_TRAIN_DEBUG"] = str(mask)
    diff(run("eager", 5), run("graph", 5), f"eager vs graph, 5, QT_TRAIN_DEBUG={mask} ({name})")
    del os.environ["QT_TRAIN_DEBUG"]

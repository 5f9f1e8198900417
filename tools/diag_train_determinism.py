"""Diagnostic: is the training step run-to-run deterministic?  N eager runs of the full-width RoBERTa layer; per run, a checksum of every
fake-quantizer call's INPUT in call order over the last step; the first call whose checksum differs from run 0 is where the runs part."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import quantized_training as qt
from quantized_training import harness, train_fusions
from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize
from test_gpu_models import _args, _TRAIN_FLAGS
from transformers import RobertaConfig, RobertaForSequenceClassification

N = int(os.environ.get("RUNS", "24"))
torch.manual_seed(0)
cfg = RobertaConfig(hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=3072, vocab_size=1000,
                    max_position_embeddings=132, num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
B, S, V = 16, 128, 1000
base = RobertaForSequenceClassification(cfg).bfloat16()
g = torch.Generator().manual_seed(1)
batches = [{"input_ids": torch.randint(3, V, (B, S), generator=g).cuda(), "labels": torch.randint(0, 2, (B,), generator=g).cuda()} for _ in range(6)]
flags = _args(*_TRAIN_FLAGS)


def run(hooks=True):
    m = copy.deepcopy(base).cuda()
    qt.quantize(m, flags)
    opt = torch.optim.AdamW(m.parameters(), lr=2e-5, fused=True, capturable=True)
    m.train()
    calls = []
    for i, b in enumerate([batches[0]] * 3 + batches[1:3]):
        if i == 1:
            train_fusions.ensure_planned(m)
        if i == 4 and hooks:
            for n, mod in m.named_modules():
                if isinstance(mod, FusedAmaxObsFakeQuantize):
                    mod.register_forward_hook(lambda mod, a, o, n=n: calls.append((n, int(a[0].detach().contiguous().view(torch.int16).long().sum()),
                                                                                   int((o if not isinstance(o, tuple) else o[0]).detach().contiguous().view(torch.int16).long().sum()))))
        opt.zero_grad(set_to_none=True)
        loss = m(**b).loss
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0, error_if_nonfinite=False)
        opt.step()
    params = {n: int(p.detach().view(torch.int16).long().sum()) for n, p in m.named_parameters()}
    grads = {n: int(p.grad.detach().view(torch.int16).long().sum()) for n, p in m.named_parameters() if p.grad is not None}
    return calls, params, grads


for hooks in (True, False):
    ref = run(hooks)
    nbad = 0
    for r in range(1, N):
        got = run(hooks)
        first = next((i for i, (a, b) in enumerate(zip(ref[0], got[0])) if a != b), None)
        gbad = [k for k in ref[2] if ref[2][k] != got[2][k]]
        pbad = [k for k in ref[1] if ref[1][k] != got[1][k]]
        if first is not None or gbad or pbad:
            nbad += 1
            print(f"run {r} (hooks {hooks}): first differing call {first} {ref[0][first][0] if first is not None else ''} "
                  f"(input differs {ref[0][first][1] != got[0][first][1] if first is not None else ''}); grads differing {len(gbad)} {gbad[:6]}; params differing {len(pbad)}")
    print(f"hooks {hooks}: {nbad} of {N - 1} runs differ from run 0; calls per step {len(ref[0])}")

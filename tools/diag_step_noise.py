"""Diagnostic: ONE training step of the full-width RoBERTa layer repeated N times from an identical state (parameters, optimizer state and
every fake-quantizer buffer restored before each repetition): checksums of every fake-quantizer call's input and output in call order, of
every gradient and of the loss.  Reports the repetitions that differ from the first and the FIRST call at which they part -- the launch
right in front of that call is where a run-to-run difference enters."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import quantized_training as qt
from quantized_training import train_fusions
from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize
from test_gpu_models import _args, _TRAIN_FLAGS
from transformers import RobertaConfig, RobertaForSequenceClassification

N = int(os.environ.get("RUNS", "400"))
torch.manual_seed(0)
cfg = RobertaConfig(hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=3072, vocab_size=1000,
                    max_position_embeddings=132, num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
B, S, V = 16, 128, 1000
m = RobertaForSequenceClassification(cfg).bfloat16().cuda()
g = torch.Generator().manual_seed(1)
batches = [{"input_ids": torch.randint(3, V, (B, S), generator=g).cuda(), "labels": torch.randint(0, 2, (B,), generator=g).cuda()} for _ in range(4)]
qt.quantize(m, _args(*_TRAIN_FLAGS))
opt = torch.optim.AdamW(m.parameters(), lr=2e-5, fused=True, capturable=True)
m.train()


def step(b):
    opt.zero_grad(set_to_none=True)
    loss = m(**b).loss
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0, error_if_nonfinite=False)
    opt.step()
    return loss


for i in range(3):
    step(batches[0])
    if i == 0:
        train_fusions.ensure_planned(m)
torch.cuda.synchronize()
snap_model = {k: v.detach().clone() for k, v in m.state_dict().items()}
snap_bufs = {n: (mod.scale.detach().clone(), mod.amax_history.detach().clone()) for n, mod in m.named_modules() if isinstance(mod, FusedAmaxObsFakeQuantize)}
snap_opt = copy.deepcopy(opt.state_dict())
calls = []
for n, mod in m.named_modules():
    if isinstance(mod, FusedAmaxObsFakeQuantize) and os.environ.get("HOOKS", "1") == "1":
        mod.register_forward_hook(lambda mod, a, o, n=n: calls.append((n, int(a[0].detach().contiguous().view(torch.int16).long().sum()),
                                                                      int((o if not isinstance(o, tuple) else o[0]).detach().contiguous().view(torch.int16).long().sum()))))


def once():
    with torch.no_grad():
        for k, v in m.state_dict().items():
            v.copy_(snap_model[k])
        for n, mod in m.named_modules():
            if isinstance(mod, FusedAmaxObsFakeQuantize):
                mod.scale.copy_(snap_bufs[n][0]); mod.amax_history.copy_(snap_bufs[n][1])
    opt.load_state_dict(copy.deepcopy(snap_opt))
    calls.clear()
    loss = step(batches[1])
    torch.cuda.synchronize()
    grads = [(n, int(p.grad.detach().view(torch.int16).long().sum())) for n, p in m.named_parameters() if p.grad is not None]
    params = [(n, int(p.detach().view(torch.int16).long().sum())) for n, p in m.named_parameters()]
    params += [("amax:" + n, int(mod.amax_history.detach().view(torch.int32).long().sum())) for n, mod in m.named_modules() if isinstance(mod, FusedAmaxObsFakeQuantize)]
    return list(calls), grads, params, float(loss)


ref = once()
nbad = 0
for r in range(1, N):
    got = once()
    first = next((i for i, (a, b) in enumerate(zip(ref[0], got[0])) if a != b), None)
    gbad = [a[0] for a, b in zip(ref[1], got[1]) if a != b]
    pbad = [a[0] for a, b in zip(ref[2], got[2]) if a != b]
    if first is not None or gbad or pbad or ref[3] != got[3]:
        nbad += 1
        what = ""
        if first is not None:
            what = f"first differing fake-quantizer call #{first} of {len(ref[0])}: {ref[0][first][0]} ({'input' if ref[0][first][1] != got[0][first][1] else 'output only'})"
        print(f"repetition {r}: loss {ref[3]} vs {got[3]}; {what}; gradients differing {len(gbad)} {gbad[:5]}; parameters differing {len(pbad)} {pbad[:4]}", flush=True)
print(f"{nbad} of {N - 1} repetitions of the same step differ from the first ({len(ref[0])} fake-quantizer calls per step)")
print("call order:", [c[0].replace('roberta.encoder.layer.0.', '') for c in ref[0]])

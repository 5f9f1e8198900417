"""Diagnostic: does any launch of the training step read memory nobody wrote?  Every torch.empty* allocation made by train_fusions.py is
filled with a poison (NaN for floating types, 0xFF bytes otherwise); the eager loop's per-call checksums, gradients and parameters must
equal the unpoisoned run's.  (Eager runs are run-to-run deterministic because the caching allocator hands the same stale blocks out in
the same order; a captured graph's private pool does not, which is where such a read shows up as a rare mismatch.)"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import quantized_training as qt
from quantized_training import harness, train_fusions
from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize
from test_gpu_models import _args, _TRAIN_FLAGS
from transformers import RobertaConfig, RobertaForSequenceClassification


class Poisoned:
    def __init__(self, kind):
        self.kind = kind

    def __getattr__(self, name):
        return getattr(torch, name)

    def _poison(self, t):
        if t.is_floating_point():
            t.fill_(float("nan") if self.kind == "nan" else 3.0e4)
        else:
            t.fill_(255 if t.dtype == torch.uint8 else -1)
        return t

    def empty(self, *a, **k):
        return self._poison(torch.empty(*a, **k))

    def empty_like(self, *a, **k):
        return self._poison(torch.empty_like(*a, **k))

    def empty_strided(self, *a, **k):
        return self._poison(torch.empty_strided(*a, **k))


torch.manual_seed(0)
cfg = RobertaConfig(hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=3072, vocab_size=1000,
                    max_position_embeddings=132, num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
B, S, V = 16, 128, 1000
base = RobertaForSequenceClassification(cfg).bfloat16()
g = torch.Generator().manual_seed(1)
batches = [{"input_ids": torch.randint(3, V, (B, S), generator=g).cuda(), "labels": torch.randint(0, 2, (B,), generator=g).cuda()} for _ in range(6)]
for b in batches:
    b["attention_mask"] = torch.ones(B, S, dtype=torch.long, device="cuda")
    b["attention_mask"][::3, 100:] = 0
flags = _args(*_TRAIN_FLAGS)


def run():
    m = copy.deepcopy(base).cuda()
    qt.quantize(m, flags)
    opt = torch.optim.AdamW(m.parameters(), lr=2e-5, fused=True, capturable=True)
    m.train()
    calls = []
    for i, b in enumerate([batches[0]] * 2 + batches[1:3]):
        if i == 1:
            train_fusions.ensure_planned(m)
        if i == 2:
            for n, mod in m.named_modules():
                if isinstance(mod, FusedAmaxObsFakeQuantize):
                    mod.register_forward_hook(lambda mod, a, o, n=n: calls.append((n, int(a[0].detach().contiguous().view(torch.int16).long().sum()),
                                                                                   int((o if not isinstance(o, tuple) else o[0]).detach().contiguous().view(torch.int16).long().sum()))))
        opt.zero_grad(set_to_none=True)
        loss = m(**b).loss
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0, error_if_nonfinite=False)
        opt.step()
    grads = {n: int(p.grad.detach().view(torch.int16).long().sum()) for n, p in m.named_parameters() if p.grad is not None}
    params = {n: int(p.detach().view(torch.int16).long().sum()) for n, p in m.named_parameters()}
    state = {n: (mod.scale.detach().float().cpu().reshape(-1).tolist(), mod.amax_history.detach().float().cpu().reshape(-1).tolist())
             for n, mod in m.named_modules() if isinstance(mod, FusedAmaxObsFakeQuantize)}
    return calls, grads, params, state


for hooked in (True, False):
    if not hooked:
        # without hooks on the fake-quantizers the launches keep g / dS / dS' on the chip and the deferred calls run: another set of buffers
        _orig = FusedAmaxObsFakeQuantize.register_forward_hook
        FusedAmaxObsFakeQuantize.register_forward_hook = lambda self, *a, **k: None
    ref = run()
    for kind in ("nan", "big"):
        mods = [mm for nn_, mm in sys.modules.items() if nn_.startswith("quantized_training") and getattr(mm, "torch", None) is torch]
        for mm in mods:
            mm.torch = Poisoned(kind)
        got = run()
        for mm in mods:
            mm.torch = torch
        first = next((i for i, (a, b) in enumerate(zip(ref[0], got[0])) if a != b), None)
        gbad = [k for k in ref[1] if ref[1][k] != got[1][k]]
        pbad = [k for k in ref[2] if ref[2][k] != got[2][k]]
        sbad = [k for k in ref[3] if ref[3][k] != got[3][k]]
        print(f"hooked {hooked} poison {kind}: first differing fake-quantizer call {first} {ref[0][first][0] if first is not None else '-'}; "
              f"gradients differing {len(gbad)} {gbad[:8]}; parameters differing {len(pbad)} {pbad[:8]}; quantizer states differing {len(sbad)} {sbad[:8]}", flush=True)

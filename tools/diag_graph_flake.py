"""Diagnostic: which launch fusion makes GraphedTrainStep replays differ from the (deterministic) eager loop now and then?"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import quantized_training as qt
from quantized_training import harness, train_fusions
from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize
from test_gpu_models import _args, _TRAIN_FLAGS
from transformers import RobertaConfig, RobertaForSequenceClassification

N = int(os.environ.get("RUNS", "16"))
torch.manual_seed(0)
cfg = RobertaConfig(hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=3072, vocab_size=1000,
                    max_position_embeddings=132, num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
B, S, V = 16, 128, 1000
base = RobertaForSequenceClassification(cfg).bfloat16()
g = torch.Generator().manual_seed(1)
batches = [{"input_ids": torch.randint(3, V, (B, S), generator=g).cuda(), "labels": torch.randint(0, 2, (B,), generator=g).cuda()} for _ in range(6)]
flags = _args(*_TRAIN_FLAGS)
ORDER = []


def run(mode, batched=True):
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
            from quantized_training import optim
            optim.clip_and_step(m.parameters(), opt, 1.0)              # (what harness.train_steps calls: the in-tree step end unless QT_TRAIN_DEBUG has bit 256)
    else:
        step = harness.GraphedTrainStep(m, opt, batch_scale_updates=batched, batch_weight_passes=batched)
        step.capture(batches[0], warmup=3)
        for b in batches[1:]:
            step.replay(b)
    torch.cuda.synchronize()
    state = {n: (mod.scale.detach().float().cpu().reshape(-1).tolist(), mod.amax_history.detach().float().cpu().reshape(-1).tolist())
             for n, mod in m.named_modules() if isinstance(mod, FusedAmaxObsFakeQuantize)}
    params = {n: int(p.detach().view(torch.int16).long().sum()) for n, p in m.named_parameters()}
    return state, params


CASES = ((0, True), (0, False), (8, True), (16, True), (32, True), (64, True), (2, True), (4, True), (1, True))
if os.environ.get("MASKS"):                     # e.g. MASKS=0,256: these masks only, batched launches
    CASES = tuple((int(m), True) for m in os.environ["MASKS"].split(","))
for mask, batched in CASES:
    os.environ["QT_TRAIN_DEBUG"] = str(mask)
    ref = run("eager")
    nbad, shown = 0, 0
    for r in range(N):
        got = run("graph", batched)
        sbad = [k for k in ref[0] if ref[0][k] != got[0][k]]
        pbad = [k for k in ref[1] if ref[1][k] != got[1][k]]
        if sbad or pbad:
            nbad += 1
            if shown < 2:
                shown += 1
                slots = {k: [i for i, (a, b) in enumerate(zip(ref[0][k][1], got[0][k][1])) if a != b] for k in sbad}
                print(f"   mask {mask} batched {batched} run {r}: {len(sbad)} quantizers differ (history slots {sorted({s for v in slots.values() for s in v})}), "
                      f"{len(pbad)} parameters differ; e.g. {sbad[:14]}")
    print(f"QT_TRAIN_DEBUG={mask} batched={batched}: {nbad} of {N} graph runs differ from the eager loop", flush=True)

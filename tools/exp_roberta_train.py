"""BASELINE config 5 shape: RoBERTa-base sequence classifier, MRPC-style batches [16, 128], int8 activations + weights
(delayed scaling) and E5M2 gradients (quantized backward), AdamW + clip 1.0 -- time per training step on device."""
import sys, time, torch
sys.path.insert(0, "quantized-training_amd")
import quantized_training as qt
from quantized_training import harness
from quantized_training.fake_quantize import STATS
from transformers import RobertaConfig, RobertaForSequenceClassification
torch.manual_seed(0)
dtype = torch.bfloat16 if "--bf16" in sys.argv else torch.float32
m = RobertaForSequenceClassification(RobertaConfig(num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)).cuda().to(dtype)
flags = ["--activation", "int8,qs=per_tensor_symmetric", "--weight", "int8,qs=per_tensor_symmetric",
         "--error", "fp8_e5m2,qs=per_tensor_symmetric,qmax=57344,ahl=10", "--quantize_forward", "gemm", "--quantize_backprop", "gemm,residual"]
if dtype == torch.bfloat16:
    flags.append("--bf16")
qt.quantize(m, qt.add_qspec_args().parse_args(flags))
g = torch.Generator().manual_seed(1)
batches = [{"input_ids": torch.randint(3, 50000, (16, 128), generator=g), "labels": torch.randint(0, 2, (16,), generator=g)} for _ in range(12)]
opt = torch.optim.AdamW(m.parameters(), lr=2e-5)
harness.train_steps(m, batches[:3], opt)
torch.cuda.synchronize()
STATS.reset()
t = time.perf_counter()
losses = harness.train_steps(m, batches[3:], opt)
torch.cuda.synchronize()
el = (time.perf_counter() - t) / 9
print(f"eager {dtype}: {el * 1e3:.2f} ms per step, {STATS.elements / 9 / 1e6:.1f} M quantized elements per step, {STATS.calls / 9:.0f} fake-quant calls, "
      f"{STATS.elements / 9 / el / 1e9:.1f} G elements/s, loss {losses[0]:.4f} -> {losses[-1]:.4f}")

# the same step as a replayed hipGraph
opt2 = torch.optim.AdamW(m.parameters(), lr=2e-5, fused=True, capturable=True) if "--fused" in sys.argv else torch.optim.AdamW(m.parameters(), lr=2e-5, capturable=True)
step = harness.GraphedTrainStep(m, opt2)
dev_batches = [{k: v.cuda() for k, v in b.items()} for b in batches]
step.capture(dev_batches[0], warmup=3)
for b in dev_batches[:3]:
    step.replay(b)
torch.cuda.synchronize()
t = time.perf_counter()
for b in dev_batches[3:]:
    loss = step.replay(b)
torch.cuda.synchronize()
el = (time.perf_counter() - t) / 9
print(f"graph {dtype}: {el * 1e3:.2f} ms per step, {765.0 / el / 1e3:.1f} G elements/s, last loss {float(loss):.4f}")

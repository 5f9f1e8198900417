"""Diagnostic: GraphedTrainStep capture of a RoBERTa classifier with L layers, with and without the grouped q / k / v backward."""
import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd")); sys.path.insert(0, ROOT)
import torch
import quantized_training as qt
from quantized_training import harness, train_fusions
from transformers import RobertaConfig, RobertaForSequenceClassification

L = int(sys.argv[1]); vocab = int(sys.argv[2]) if len(sys.argv) > 2 else 50265
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = RobertaForSequenceClassification(RobertaConfig(num_labels=2, num_hidden_layers=L, vocab_size=vocab, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)).to(dev).bfloat16()
qt.quantize(model, qt.add_qspec_args().parse_args(["--activation", "int8,qs=per_tensor_symmetric", "--weight", "int8,qs=per_tensor_symmetric", "--error",
                                                   "fp8_e5m2,qs=per_tensor_symmetric,qmax=57344,ahl=10", "--quantize_forward", "gemm", "--quantize_backprop", "gemm,residual", "--bf16"]))
g = torch.Generator().manual_seed(1)
batches = [{"input_ids": torch.randint(3, vocab, (16, 128), generator=g).to(dev), "labels": torch.randint(0, 2, (16,), generator=g).to(dev)} for _ in range(6)]
opt = torch.optim.AdamW(model.parameters(), lr=2e-5, fused=True, capturable=True)
harness.train_steps(model, batches[:2], torch.optim.AdamW(model.parameters(), lr=2e-5))
train_fusions.STATS.reset()
step = harness.GraphedTrainStep(model, opt)
step.capture(batches[0], warmup=3)
print("captured; qkv groups in the captured pass:", train_fusions.STATS.qkv_groups, "pending linear grads", len(train_fusions._LINEAR_GRADS), flush=True)
for i in range(3):
    loss = step.replay(batches[3 + i])
torch.cuda.synchronize()
print("layers", L, "mask", os.environ.get("QT_TRAIN_DEBUG", "0"), "ok, loss", float(loss))

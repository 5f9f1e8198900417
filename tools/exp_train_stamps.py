"""GPU experiment (tuning build): where the chain / LayerNorm-backward kernels of the configs[4] training step spend their time.

    QT_HIP_LIB=tools/build/libqt_hip_tuning.so python tools/exp_train_stamps.py [layers]

Every workgroup (first 512) of fq_chain_kernel / ln_train_bwd_kernel stamps s_memrealtime (100 MHz, chip-wide) in wave 0 and in its
last wave at a few points (QT_EW_STAMPS, csrc/qt_elementwise.hip).  The step is captured into a hipGraph as bench.py does and replayed;
each captured launch keeps its own stamp region, so the last replay's stamps of every launch are read back together.  Printed per
launch: tag (0x1NS. chain, 0x2NS. LayerNorm backward), grid, the launch's span (first start -> last end), the spread of workgroup
starts, and the median over workgroups of each phase (us after the workgroup's own start; the last four columns of a chain launch: round 1
loads arrived / round 1 done / round 2 loads arrived / round 2 done; QT_CHAIN_ABLATE=1 / 2 / 3 times the launches without their GELU
arithmetic / fake-quantizer stages / both).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("QT_HIP_LIB", os.path.join(ROOT, "tools", "build", "libqt_hip_tuning.so"))

import torch  # noqa: E402
import quantized_training as qt  # noqa: E402
from quantized_training import harness  # noqa: E402
from transformers import RobertaConfig, RobertaForSequenceClassification  # noqa: E402

NL = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
REGION = 512 * 32
stamps = torch.zeros(256 * REGION, dtype=torch.int64, device=dev)
torch.manual_seed(0)
model = RobertaForSequenceClassification(RobertaConfig(num_labels=2, num_hidden_layers=NL, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)).to(dev).bfloat16()
qt.quantize(model, qt.add_qspec_args().parse_args(["--activation", "int8,qs=per_tensor_symmetric", "--weight", "int8,qs=per_tensor_symmetric", "--error",
                                                   "fp8_e5m2,qs=per_tensor_symmetric,qmax=57344,ahl=10", "--quantize_forward", "gemm", "--quantize_backprop", "gemm,residual", "--bf16"]))
g = torch.Generator().manual_seed(1)
batches = [{"input_ids": torch.randint(3, 50265, (16, 128), generator=g).to(dev), "labels": torch.randint(0, 2, (16,), generator=g).to(dev)} for _ in range(6)]
opt = torch.optim.AdamW(model.parameters(), lr=2e-5, fused=True, capturable=True)
harness.train_steps(model, batches[:2], torch.optim.AdamW(model.parameters(), lr=2e-5))
step = harness.GraphedTrainStep(model, opt)
step.capture(batches[0], warmup=2)
os.environ["QT_EW_STAMPS"] = hex(stamps.data_ptr())       # from here on every launch takes a region: re-capture with the regions in the arguments
step2 = harness.GraphedTrainStep(model, opt)
step2.capture(batches[0], warmup=0)
del os.environ["QT_EW_STAMPS"]
for i in range(5):
    step2.replay(batches[1 + i])
torch.cuda.synchronize()
s = stamps.cpu().view(256, 512, 32)
print("launch  tag    grid   span_us  start_spread_us | wave 0: phases (median us after own start) | last wave: phases")
rows = []
for li in range(256):
    head = int(s[li, 0, 7])
    if head == 0:
        continue
    tag, grid = head >> 32, head & 0xFFFFFFFF
    n = min(grid, 512)
    r = s[li, :n].double() / 100.0       # us
    t0 = r[:, 0]
    w0 = torch.cat([r[:, :7], r[:, 8:12]], dim=1)
    w7 = torch.cat([r[:, 16:23], r[:, 24:28]], dim=1)
    ends = torch.maximum(w0.max(dim=1).values, w7.max(dim=1).values)
    span = float(ends.max() - t0.min())
    spread = float(t0.max() - t0.min())

    def phases(w):
        out = []
        for k in range(1, w.shape[1]):
            col = w[:, k]
            ok = col > 0
            out.append(f"{float((col[ok] - t0[ok]).median()):6.2f}" if ok.any() else "     -")
        return " ".join(out)
    rows.append((float(t0.min()), f"{li:4d}  {tag:#05x} {grid:5d}  {span:7.2f}  {spread:7.2f} | {phases(w0)} | {phases(w7)}"))
for _, line in sorted(rows):
    print(line)

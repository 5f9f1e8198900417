"""Diagnostic: per-tap code-step shares of one LLaMA-2-7B decoder layer, device default route vs CPU (tests/test_gpu_models.py helpers)."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import quantized_training as qt
from oracle import qt_oracle as o
from test_gpu_models import _tap_fake_quantizers, _args
from transformers import LlamaConfig, LlamaModel

torch.manual_seed(0)
cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=1, num_attention_heads=32, num_key_value_heads=32,
                  vocab_size=2048, max_position_embeddings=1024, attn_implementation="eager")
base = LlamaModel(cfg).eval().bfloat16()
ids = torch.randint(0, 2048, (1, 1024), generator=torch.Generator().manual_seed(2))


def build(dev, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    m = copy.deepcopy(base).to(dev)
    qt.quantize(m, _args("--activation", "e4m3", "--weight", "e4m3", "--bf16", "--quantize_forward", "gemm"))
    taps, out = _tap_fake_quantizers(m, lambda: m(ids.to(dev), use_cache=False))
    for k in (env or {}):
        del os.environ[k]
    return taps, out.last_hidden_state.float().cpu()


qmap = o.get_quantization_map("e4m3")
vals = o.bf16_to_f32(qmap)
grid = np.unique(vals[np.isfinite(vals)].astype(np.float64))


def report(name, ref, got):
    print(f"== {name}")
    for k in ref[0]:
        if k not in got[0]:
            print(f"   {k}: not tapped on this route")
            continue
        for a, b in zip(ref[0][k], got[0][k]):
            a, b = a.numpy().astype(np.float64).ravel(), b.numpy().astype(np.float64).ravel()
            steps = np.abs(np.searchsorted(grid, a) - np.searchsorted(grid, b))
            nz = float((a != 0).mean())
            print(f"   {k:60s} {tuple(ref[0][k][0].shape)}: one step {float((steps == 1).mean()):.4f}  further {float((steps > 1).mean()):.2e}  "
                  f"nonzero share {nz:.3f}  rms {np.sqrt((a * a).mean()):.3e}")
    d = (got[1] - ref[1]).abs()
    s = float(ref[1].abs().max())
    print(f"   hidden: rms {float(d.pow(2).mean().sqrt()) / s:.3e} max {float(d.max()) / s:.3e} (of max |h| {s:.3e})")


cpu = build("cpu")
dev = build("cuda")
report("device default vs cpu", cpu, dev)
plain = build("cuda", {"QT_FP8_GEMM": "0", "QT_FUSED_MODEL_OPS": "0", "QT_FUSED_SOFTMAX": "0", "QT_FUSED_ATTENTION": "0"})
report("device plain vs cpu", cpu, plain)
report("device default vs device plain", plain, dev)

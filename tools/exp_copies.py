"""Where do the device-to-device copies in a quantized LLaMA layer come from?  Logs every Tensor.copy_ / contiguous /
clone that actually moves data during one window forward, with the calling frames inside this package / transformers."""
import collections
import sys
import traceback

import torch

sys.path.insert(0, "quantized-training_amd")
import quantized_training as qt
from quantized_training import harness

m = harness.build_causal_lm("llama-2-7b", device="cuda", seed=0, num_layers=2)
qt.quantize(m, qt.add_qspec_args().parse_args(["--activation", "e4m3", "--weight", "e4m3", "--bf16", "--quantize_forward", "gemm"]))
ids = torch.randint(0, 32000, (1, 1024), device="cuda")
log = collections.Counter()


def where():
    fr = [f for f in traceback.extract_stack()[:-2] if "quantized_training" in f.filename or "transformers" in f.filename]
    return " <- ".join(f"{f.filename.split('/')[-1]}:{f.lineno}" for f in fr[-3:][::-1])


orig_contig, orig_copy, orig_clone = torch.Tensor.contiguous, torch.Tensor.copy_, torch.Tensor.clone


def contiguous(self, *a, **k):
    if not self.is_contiguous():
        log[("contiguous", tuple(self.shape), str(self.dtype), where())] += 1
    return orig_contig(self, *a, **k)


def copy_(self, src, *a, **k):
    log[("copy_", tuple(self.shape), str(self.dtype), where())] += 1
    return orig_copy(self, src, *a, **k)


def clone(self, *a, **k):
    log[("clone", tuple(self.shape), str(self.dtype), where())] += 1
    return orig_clone(self, *a, **k)


with torch.no_grad():
    harness.window_nll(m, ids, 512)
    harness.window_nll(m, ids, 512)
    torch.Tensor.contiguous, torch.Tensor.copy_, torch.Tensor.clone = contiguous, copy_, clone
    harness.window_nll(m, ids, 512)
    torch.Tensor.contiguous, torch.Tensor.copy_, torch.Tensor.clone = orig_contig, orig_copy, orig_clone
for k, v in log.most_common():
    print(v, k)

"""Runs only bench.py's value-map GEMM leg (qt_linear_fqt_bf16 at the LLaMA-2-13B q / k / v shape, weights rotating beyond the
Infinity Cache) so that rocprofv3 --kernel-trace / --pmc passes see just that kernel."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    print(json.dumps(bench.fqt_gemm_leg(dev, "posit8_2", 1024, [5120, 5120, 5120], 5120)))

"""Runs only bench.py's roofline leg (the fused fake-quant pass over a > 256 MiB pool) so that
rocprofv3 --pmc passes see just that kernel."""
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
    print(json.dumps(bench.roofline_leg(dev)))

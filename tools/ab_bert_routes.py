"""A/B on ONE box: the configs[1] batch (BERT-base, E4M3) with single Linear shapes moved to the fused FP8 GEMM."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "quantized-training_amd"))
sys.argv = ["bench.py", "--workload", "bert-base-squad-e4m3", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-roofline"]
from quantized_training import fused
ab = os.environ.get("AB", "default")
if ab != "default":
    n, k = (int(v) for v in ab.split("x"))
    fused._FQ8_TABLE[(6144, n, k)] = True
import bench
bench.main()
''' % (ROOT, ROOT)
for rep in range(2):
    for ab in ["default", "2304x768", "3072x768", "768x3072"]:
        out = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, AB=ab), capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(f"{ab:10s}", round(json.loads(line[-1])["ms_per_step"], 4) if line else out.stderr[-300:], flush=True)

"""A/B on ONE box: the headline window (LLaMA-2-7B-shaped, E4M3) with single Linear shapes moved to the other route."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "quantized-training_amd"))
sys.argv = ["bench.py", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-roofline", "--no-secondary"]
from quantized_training import fused
ab = os.environ.get("AB", "default")
if ab == "mlp_two_gemms":
    fused._MLP_TABLE[(1024, 11008, 4096)] = False
elif ab != "default":
    n, k = (int(v) for v in ab.split("x"))
    fused._FQ8_TABLE[(1024, n, k)] = not fused._FQ8_TABLE.get((1024, n, k), True)
import bench
bench.main()
''' % (ROOT, ROOT)
for rep in range(2):
    for ab in sys.argv[1:] or ["default", "4096x11008", "4096x4096", "12288x4096", "mlp_two_gemms"]:
        out = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, AB=ab), capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(f"{ab:14s}", round(json.loads(line[-1])["ms_per_step"], 4) if line else out.stderr[-300:], flush=True)

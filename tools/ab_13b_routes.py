"""A/B on ONE box: the configs[3] window (LLaMA-2-13B-shaped, posit(8,2)) with different Linear routes -- each variant runs bench.py's
workload in a child process of its own, alternating, so box-to-box spread (+-5 %) does not enter the comparison.

    python tools/ab_13b_routes.py            # default rule | gate / up as two launches | everything fused | split-K route off
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "quantized-training_amd"))
sys.argv = ["bench.py", "--workload", "llama-13b-posit8_2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-roofline"]
from quantized_training import fused
rule = fused.fqt_route_is_fused
ab = os.environ.get("AB", "default")
if ab == "gate_up_separate":
    os.environ["QT_GATE_UP_GROUP"] = "0"
elif ab == "all_fused":
    fused.fqt_route_is_fused = lambda M, ns, K, dev: M > 256
elif ab == "no_split":
    fused.fqt_route_is_fused = lambda M, ns, K, dev: False if fused.fqt_plan(M, sum(ns), K)[0] > 1 else rule(M, ns, K, dev)
import bench
bench.main()
''' % (ROOT, ROOT)

variants = sys.argv[1:] or ["default", "gate_up_separate", "all_fused", "no_split"]
for rep in range(2):
    for ab in variants:
        out = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, AB=ab), capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if line:
            j = json.loads(line[-1])
            lib = sum(1 for v in j["config"]["routes"].values() if "library" in v)
            print(f"{ab:14s} {j['ms_per_step']:7.3f} ms   mean NLL {j['mean_window_nll']:.6f}   shapes on the library route: {lib}", flush=True)
        else:
            print(ab, "FAILED", out.stderr[-400:], flush=True)

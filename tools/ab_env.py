"""A/B on ONE box, alternating child processes: a bench workload under sets of environment switches.

    python tools/ab_env.py [--workload llama-7b-e4m3] [--reps 2] [--steps 5] "" QT_FQ8_SPLITK=0 "QT_A=1 QT_B=0"
"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="llama-7b-e4m3")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("variants", nargs="+")
args = ap.parse_args()
for rep in range(args.reps):
    for v in args.variants:
        env = dict(os.environ)
        env.update(dict(kv.split("=", 1) for kv in v.split()))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", args.workload, "--steps", str(args.steps), "--warmup", "2",
                              "--no-cpu-baseline", "--no-roofline", "--no-secondary"], env=env, capture_output=True, text=True, cwd=ROOT)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(f"{args.workload} [{v or 'default'}]", round(json.loads(line[-1])["ms_per_step"], 4) if line else out.stderr[-400:], flush=True)

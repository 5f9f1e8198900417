"""Ordered kernel sequence of ONE replayed step from a rocprofv3 kernel trace (the last full step between two launches of `--anchor`):
index, duration, gap to the previous kernel, name.  For reading what a training step actually launches, in order.

    python tools/step_sequence.py gpurun_out/prof_train_stats --anchor scale_update_multi_kernel > gpurun_out/train_step_sequence.txt
"""
import argparse, csv, glob, os, re
ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--anchor", default="scale_update_multi_kernel")
args = ap.parse_args()
path = max(glob.glob(os.path.join(args.dir, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
idx = [i for i, r in enumerate(rows) if args.anchor in r[2]]
assert len(idx) >= 2, "fewer than two anchor launches in the trace"
lo, hi = idx[-2], idx[-1]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"at::native::", "", n)
    return n[:110]


print(f"# {hi - lo} launches, {(rows[hi][0] - rows[lo][0]) / 1e3:.1f} us")
prev = rows[lo][0]
for k, (s, e, n) in enumerate(rows[lo:hi]):
    print(f"{k:5d} {(e - s) / 1e3:8.2f} us  gap {(s - prev) / 1e3:6.2f}  {short(n)}")
    prev = e

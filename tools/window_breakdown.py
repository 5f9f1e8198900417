"""Per-window kernel time of the headline bench from a rocprofv3 kernel trace: takes the last `windows` windows of the run
(the timed hipGraph replays; a window holds `layers` softmax_fq launches) and prints, per kernel, launches per window and
microseconds per window, sorted by time.

    python tools/window_breakdown.py gpurun_out/prof_bench [--windows 5] [--layers 32] > profiles/r02_window_breakdown.txt
"""
import argparse
import collections
import csv
import glob
import os
import re

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--windows", type=int, default=5)
ap.add_argument("--layers", type=int, default=32)
ap.add_argument("--anchor", default="attention_fp8_split_kernel,attention_fp8_kernel,softmax_fq_kernel",
                help="kernel(s) launched once per layer, comma separated: the first one present in the trace marks the windows")
args = ap.parse_args()

path = max(glob.glob(os.path.join(args.dir, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
anchors = []
for name in args.anchor.split(","):
    anchors = [i for i, r in enumerate(rows) if name in r[2]]
    if anchors:
        break
need = args.windows * args.layers
assert len(anchors) >= need + args.layers, "trace holds fewer windows than asked for"
# a window starts at the first kernel after the previous window's last anchor launch + everything behind it in that layer:
# cut at the anchor launches themselves (the same phase in every window), which makes whole windows of the slice.
lo, hi = anchors[-need - 1], anchors[-1]
sel = rows[lo:hi]
span = (rows[hi][0] - rows[lo][0]) / 1e3 / args.windows


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"(Custom_)?Cijk_[A-Za-z]+_[A-Za-z]+_([A-Z0-9]+)_\w*?(MT\d+x\d+x\d+)", n)
    if m:                                               # Tensile name: ..._<types>_..., F8.. = FP8 inputs, BBS / BSS = bf16
        kind = "FP8" if m.group(2).startswith("F8") else ("bf16" if m.group(2).startswith("B") else m.group(2))
        return f"hipBLASLt {kind} GEMM {m.group(3)}" + (" (custom)" if m.group(1) else "")
    return re.sub(r"\(.*", "", n)[:72]


t = collections.defaultdict(lambda: [0, 0])
for s, e, n in sel:
    k = short(n)
    t[k][0] += 1
    t[k][1] += e - s
busy = sum(v[1] for v in t.values()) / 1e3 / args.windows
print(f"trace: {os.path.relpath(path)}")
print(f"{args.windows} windows, {span:.1f} us per window wall, {busy:.1f} us per window inside kernels ({len(sel) / args.windows:.0f} launches)")
print(f"{'kernel':74s} {'launches':>8s} {'us/window':>10s} {'us/launch':>10s} {'share':>6s}")
for k, (c, ns) in sorted(t.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:74s} {c / args.windows:8.1f} {ns / 1e3 / args.windows:10.1f} {ns / 1e3 / c:10.2f} {ns / 1e3 / args.windows / busy:6.1%}")

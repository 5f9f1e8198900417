"""Copies the rocprofv3 summaries of the last tools/gpu_session_bench.sh run from gpurun_out/ into
profiles/ (tracked) and derives the per-launch HBM traffic of the dominant kernel from the PMC passes
(FETCH_SIZE x2 on gfx950 for 16-B/lane streaming reads, MI355X_MICROARCH.md HBM section)."""
import csv
import glob
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
FRESH_S = float(os.environ.get("FRESH_HOURS", "4")) * 3600.0
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)


def one(pattern):
    f = glob.glob(os.path.join(ROOT, "gpurun_out", pattern))
    return max(f, key=os.path.getmtime) if f else None          # the latest session's file, whatever its numeric prefix


def pmc(path, name, kernel="fq_kernel"):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(vals) / len(vals), len(vals)


shutil.copy(one("prof_bench/*/*kernel_stats.csv"), os.path.join(out, f"{tag}_bench_kernel_stats.csv"))
shutil.copy(one("prof_roofline/*/*kernel_stats.csv"), os.path.join(out, f"{tag}_roofline_kernel_stats.csv"))
shutil.copy(os.path.join(ROOT, "gpurun_out", "bench_n1.json"), os.path.join(out, f"{tag}_bench_n1.json"))
rows = list(csv.DictReader(open(os.path.join(out, f"{tag}_roofline_kernel_stats.csv"))))
bench = json.load(open(os.path.join(out, f"{tag}_bench_n1.json")))
n = 4096 * 11008


def traffic_of(kernel, bytes_per_elem, bench_ms):
    fetch, nf = pmc(one("pmc_fetch/*/*counter_collection.csv"), "FETCH_SIZE", kernel)
    write, nw = pmc(one("pmc_write/*/*counter_collection.csv"), "WRITE_SIZE", kernel)
    k = [r for r in rows if kernel in r["Name"]][0]
    return {
        "kernel": k["Name"][:120],
        "launches_sampled": [nf, nw],
        "FETCH_SIZE_KB_raw_per_launch": fetch,
        "fetch_bytes_per_launch": int(fetch * 1024 * 2),
        "WRITE_SIZE_KB_per_launch": write,
        "write_bytes_per_launch": int(write * 1024),
        "hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),
        "algorithmic_bytes_per_launch": n * bytes_per_elem,
        "traffic_over_algorithmic": (fetch * 1024 * 2 + write * 1024) / (n * bytes_per_elem),
        "kernel_avg_duration_us_rocprof": float(k["AverageNs"]) / 1e3,
        "kernel_avg_duration_us_hip_events_in_bench": bench_ms * 1e3,
    }


roof = bench["roofline"].get("elementwise_pass", bench["roofline"])      # round 2: the pass sits beside the fused GEMM
res = traffic_of("fq_kernel", 4, roof["bf16_out"]["ms_per_launch"])
res.update({
    "tensor": "bf16[4096,11008], 8-tensor rotating pool (1.44 GB in + out)",
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/roofline_only.py",
    "FETCH_SIZE_correction": "x2 on gfx950 (16 B/lane coalesced streaming reads are tallied at half)",
    "fp8_only": traffic_of("fq8_kernel", 3, roof["ms_per_launch"]),
})
try:                                                     # the fused FP8 GEMM leg (round 2): same passes, its own kernel rows
    fetch, nf = pmc(one("pmc_fetch/*/*counter_collection.csv"), "FETCH_SIZE", "linear_fq8r_kernel")
    write, nw = pmc(one("pmc_write/*/*counter_collection.csv"), "WRITE_SIZE", "linear_fq8r_kernel")
    k = [r for r in rows if "linear_fq8r_kernel" in r["Name"]][0]
    M_, N_, K_ = 1024, 11008, 4096
    alg = N_ * K_ * 2 + M_ * K_ + M_ * N_ * 2
    res["fused_gemm"] = {
        "kernel": k["Name"][:120], "launches_sampled": [nf, nw], "FETCH_SIZE_KB_raw_per_launch": fetch,
        "fetch_bytes_per_launch": int(fetch * 1024 * 2), "WRITE_SIZE_KB_per_launch": write, "write_bytes_per_launch": int(write * 1024),
        "hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024), "algorithmic_bytes_per_launch": alg,
        "traffic_over_algorithmic": (fetch * 1024 * 2 + write * 1024) / alg,
        "kernel_avg_duration_us_rocprof": float(k["AverageNs"]) / 1e3,
        "note": "1024 x 11008 x 4096: weights 2 B/element once + FP8 activations once + bf16 output once; fetches beyond that are the "
                "activation tile re-read by the 64 column tiles through L2 misses",
    }
except Exception as e:  # noqa: BLE001
    print("no fused GEMM rows in the PMC passes:", e)
json.dump(res, open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))


def clean(src, dst):
    """Experiment logs without the profiler's own chatter."""
    if not os.path.exists(src) or time.time() - os.path.getmtime(src) > FRESH_S:       # an older session's file: not this round's evidence
        return
    keep = [ln for ln in open(src, errors="replace") if not any(t in ln for t in ("amdgpu.ids", "rocprofv3", "output_stream.cpp", "simple_timer.cpp", "tool.cpp"))]
    open(dst, "w").writelines(keep)


# per-window breakdown of the headline run, the LLaMA-2-13B posit(8,2) leg, block-scaled path, kernel experiments
G = os.path.join(ROOT, "gpurun_out")
for src, dst in (("window_breakdown.txt", "window_breakdown.txt"), ("window_breakdown_13b_posit.txt", "13b_posit8_2_window_breakdown.txt"),
                 ("bench_13b_posit8_2.json", "13b_posit8_2_bench.json"), ("mx_gemm.log", "mx_gemm.txt"), ("mx_linear.log", "mx_linear.txt"),
                 ("mx_quant.log", "mx_quant.txt"), ("mx_wide.log", "mx_wide_ablation.txt"), ("tile_fetch.log", "tile_fetch_probe.txt"),
                 ("fq8_session.txt", "linear_fq8_gemm.txt"), ("mlp_fq8.txt", "mlp_fq8.txt"), ("attention_fp8.txt", "attention_fp8.txt"), ("table_formats.txt", "table_formats.txt"), ("oracle_attention.txt", "oracle_attention.txt"),
                 ("bert_batch.txt", "bert_batch.txt"), ("train_step.txt", "train_step.txt"),
                 # round 3
                 ("bench_llama-13b-posit8_2.json", "13b_posit8_2_bench.json"), ("bench_bert-base-squad-e4m3.json", "bert_base_squad_bench.json"),
                 ("bench_roberta-mrpc-int8-e5m2-train.json", "roberta_mrpc_train_bench.json"), ("fqt_gemm.txt", "linear_fqt_gemm.txt"),
                 ("fqt_ablate.txt", "linear_fqt_ablations.txt"), ("fqt_stamps.txt", "linear_fqt_step_stamps.txt"), ("fq8_routes.txt", "fq8_routes.txt"),
                 ("pt2e_bench.json", "pt2e_route_bench.json"), ("fq8_ablate.txt", "linear_fq8_ablations.txt"),
                 # round 4
                 ("train_step_breakdown.txt", "train_step_breakdown.txt"), ("bert_batch_breakdown.txt", "bert_batch_breakdown.txt"),
                 ("ab_13b_routes.txt", "13b_route_ab.txt"), ("ab_7b_routes.txt", "7b_route_ab.txt"), ("ab_bert_routes.txt", "bert_route_ab.txt"),
                 ("small_fq.txt", "small_fq.txt"), ("attention_rows.txt", "attention_rows.txt"), ("attention_rows_stamps.txt", "attention_rows_stamps.txt"),
                 # round 5
                 ("train_step_sequence.txt", "train_step_sequence.txt"), ("train_step_ab.txt", "train_step_ab.txt"), ("chain_microbench.txt", "chain_microbench.txt"),
                 ("bench_roberta-mrpc-int8-e5m2-train-dropout.json", "roberta_mrpc_train_dropout_bench.json"),
                 ("attention_train_stamps.txt", "attention_train_stamps.txt"),
                 # round 6
                 ("train_stamps.txt", "train_stamps.txt"), ("graph_branches.txt", "graph_branches.txt")):
    clean(os.path.join(G, src), os.path.join(out, f"{tag}_{dst}"))
for pattern, dst in (("prof_13b_posit/*/*kernel_stats.csv", "13b_posit8_2_kernel_stats.csv"), ("prof_mx_gemm/*/*kernel_stats.csv", "mx_gemm_kernel_stats.csv"),
                     ("prof_mx_layer/*/*kernel_stats.csv", "mx_layer_kernel_stats.csv"),
                     ("prof_bert_stats/*/*kernel_stats.csv", "bert_kernel_stats.csv"), ("prof_train_stats/*/*kernel_stats.csv", "train_kernel_stats.csv"),
                     ("prof_fqt/*/*kernel_stats.csv", "linear_fqt_kernel_stats.csv")):
    f = one(pattern)
    if f and time.time() - os.path.getmtime(f) <= FRESH_S:
        shutil.copy(f, os.path.join(out, f"{tag}_{dst}"))

# round 3: HBM traffic of the value-map GEMM (qt_linear_fqt_bf16, 1024 x 15360 x 5120) from its own PMC passes
try:
    fetch, nf = pmc(one("pmc_fetch_fqt/*/*counter_collection.csv"), "FETCH_SIZE", "linear_fqt_kernel")
    write, nw = pmc(one("pmc_write_fqt/*/*counter_collection.csv"), "WRITE_SIZE", "linear_fqt_kernel")
    krows = list(csv.DictReader(open(one("prof_fqt/*/*kernel_stats.csv"))))
    k = [r for r in krows if "linear_fqt_kernel" in r["Name"]][0]
    M_, N_, K_ = 1024, 15360, 5120
    alg = N_ * K_ * 2 + M_ * K_ * 2 + M_ * N_ * 2
    fq = {"kernel": k["Name"][:120], "launches_sampled": [nf, nw], "FETCH_SIZE_KB_raw_per_launch": fetch, "fetch_bytes_per_launch": int(fetch * 1024 * 2),
          "WRITE_SIZE_KB_per_launch": write, "write_bytes_per_launch": int(write * 1024), "hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),
          "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (fetch * 1024 * 2 + write * 1024) / alg,
          "kernel_avg_duration_us_rocprof": float(k["AverageNs"]) / 1e3,
          "note": "1024 x 15360 x 5120 (q / k / v of LLaMA-2-13B as one launch): bf16 weights once + bf16 activations once + bf16 output once"}
    path = os.path.join(out, f"{tag}_pmc_traffic.json")
    res = json.load(open(path)) if os.path.exists(path) else {}
    res["value_map_gemm"] = fq
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps(fq, indent=1))
except Exception as e:  # noqa: BLE001
    print("no value-map GEMM rows in the PMC passes:", e)

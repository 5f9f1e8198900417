#!/bin/bash
# PMC passes over the fused FP8 GEMM (qt_linear_fq8_bf16, 1024 x 11008 x 4096): translation, L1 / L2 request path, issue counters.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for set in "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_REQ_sum" \
           "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY GRBM_UTCL2_BUSY GRBM_EA_BUSY"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_fq8_$i
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_fq8_$i -- python3 tools/exp_linear_fq8.py --skip-checks --iters 30 --shapes 1024x11008x4096 > gpurun_out/pmc_fq8_$i.log 2>&1
done
python3 - <<'PY' > gpurun_out/fq8_pmc_summary.txt
import csv, glob, collections
out = collections.OrderedDict()
for d in sorted(glob.glob("gpurun_out/pmc_fq8_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        acc = collections.defaultdict(list)
        for r in rows:
            if "linear_fq8r_kernel" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[k] = (sum(v) / len(v), len(v))
for k, (v, n) in out.items():
    print(f"{k:48s} {v:18.1f}   per launch, mean of {n}")
PY
cat gpurun_out/fq8_pmc_summary.txt
rm -rf gpurun_out/pmc_fq8_[0-9]*/

#!/bin/bash
# PMC passes over qt_linear_fqt_bf16 at 1024 x 13824 x 5120 (tools/exp_linear_fqt.py --shapes one): where do the waves' cycles go?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/fqt_pmc
mkdir -p $OUT
run() {   # name, counters...
  n=$1; shift
  rocprofv3 --pmc "$@" -d $OUT/$n -o $n --output-format csv -- python3 $R/tools/exp_linear_fqt.py --iters 3 --shapes one --skip-checks > $OUT/$n.log 2>&1
}
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run p2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU
run p3 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
python3 - <<PY
import csv,glob,collections
for n in ("p1","p2","p3"):
    files=glob.glob("$OUT/%s/**/*counter_collection.csv"%n,recursive=True)
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            k=row.get("Kernel_Name","")
            if "linear_fqt" not in k: continue
            agg[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,d in agg.items():
        print(n,k)
        for c,v in d.items(): print("   %-28s n=%d mean=%.4g"%(c,len(v),sum(v)/len(v)))
PY

#!/bin/bash
# Round 6 GPU session 1: hardware counters behind the fused FP8 GEMM's MFMA fraction (north_star: "rocprof showing ... MFMA utilisation
# for the GEMM against peak"): matrix-core busy cycles, LDS activity / conflicts / stalls, vector issue, wave-cycle split -- for the wide
# kernel (gate / up: 1024 x 11008 x 4096, linear_fq8r_kernel) and the narrow one (down / o: linear_fq8r2_kernel); then the clock the chip
# holds inside those tiles (tuning build stamps), whole kernel and ablations.  Summary -> gpurun_out/r06_linear_fq8_pmc.txt.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
export FUSED_ONLY=1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_INSTS_VALU_MFMA_MOPS_F6F4 SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_CYCLES" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_INSTS_VALU_MFMA_F8" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc6_$i
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc6_$i -- python3 tools/exp_linear_fq8.py --skip-checks --iters 30 --shapes 1024x11008x4096,1024x4096x11008,1024x4096x4096 > gpurun_out/pmc6_$i.log 2>&1
  tail -2 gpurun_out/pmc6_$i.log
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc6_trace -- python3 tools/exp_linear_fq8.py --skip-checks --iters 30 --shapes 1024x11008x4096,1024x4096x11008,1024x4096x4096 > gpurun_out/pmc6_trace.log 2>&1
python3 - <<'PY' > gpurun_out/r06_linear_fq8_pmc.txt
import csv, glob, collections
out = {"linear_fq8r_kernel": collections.OrderedDict(), "linear_fq8r2_kernel": collections.OrderedDict()}
for d in sorted(glob.glob("gpurun_out/pmc6_[0-9]*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            for k in out:
                if k in r.get("Kernel_Name", ""):
                    acc[(k, r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
        for (k, c, g), v in acc.items():
            out[k][(c, g)] = (sum(v) / len(v), len(v))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc6_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in out:
            if k in r.get("Kernel_Name", ""):
                dur[(k, r.get("Grid_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# rocprofv3 --pmc, per launch, summed over the chip (256 CUs x 4 SIMDs); grid = threads (workgroups x 512)")
for k in out:
    print(f"== {k}")
    for (k2, g), v in sorted(dur.items()):
        if k2 == k:
            v = sorted(v)
            print(f"   kernel-trace duration grid {g}: median {v[len(v)//2]:.2f} us  min {v[0]:.2f}  n {len(v)}")
    for (c, g), (v, n) in out[k].items():
        print(f"   {c:36s} grid {g:>8s} {v:18.1f}   mean of {n}")
PY
cat gpurun_out/r06_linear_fq8_pmc.txt
find gpurun_out/pmc6_trace -name "*kernel_trace.csv" -delete
rm -rf gpurun_out/pmc6_[0-9]*/
unset FUSED_ONLY
QT_HIP_LIB=tools/build/libqt_hip_tuning.so timeout 600 python3 tools/exp_fq8_clock.py 2>&1 | grep -v Warn > gpurun_out/r06_fq8_clock.txt; cat gpurun_out/r06_fq8_clock.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-secondary > gpurun_out/r06_bench_base.json 2> gpurun_out/r06_bench_base.err; head -c 2500 gpurun_out/r06_bench_base.json

#!/bin/bash
# One GPU session: headline bench, kernel-trace profile of the same command, PMC traffic passes for the
# dominant elementwise kernel.  Outputs land in gpurun_out/ and are copied into profiles/ by tools/collect_profiles.py.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
cat gpurun_out/bench_n1.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/prof_bench.log 2>&1
python tools/window_breakdown.py gpurun_out/prof_bench --windows 5 > gpurun_out/window_breakdown.txt 2>&1
cat gpurun_out/window_breakdown.txt | head -40
if [ "$1" != "quick" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_roofline -- python3 tools/roofline_only.py > gpurun_out/prof_roofline.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/roofline_only.py > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/roofline_only.py > gpurun_out/pmc_write.log 2>&1
fi
# the raw traces are large: keep the summaries only
find gpurun_out/prof_bench -name "*kernel_trace.csv" -delete
find gpurun_out -name "*stats.csv" -newer gpurun_out/bench_n1.json | head

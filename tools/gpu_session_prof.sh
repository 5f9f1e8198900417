#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench3 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/prof_bench3.log 2>&1
tail -1 gpurun_out/prof_bench3.log | cut -c1-300

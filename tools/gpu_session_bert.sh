#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bert -- python3 tools/exp_bert.py > gpurun_out/prof_bert.log 2>&1
grep -E "eager|hipGraph" gpurun_out/prof_bert.log

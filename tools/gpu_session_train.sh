#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_train -- python3 tools/exp_roberta_train.py --bf16 --fused > gpurun_out/prof_train.log 2>&1
tail -2 gpurun_out/prof_train.log

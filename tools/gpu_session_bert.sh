#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_bert -- python3 tools/exp_bert_graph.py > gpurun_out/prof_bert.log 2>&1
grep "replay ms" gpurun_out/prof_bert.log

#!/bin/bash
# Kernel-trace summaries of the two secondary configurations: BERT-base QA batch (replayed graph) and the RoBERTa-base
# training step (replayed graph).  Copied into profiles/ as r01_bert_kernel_stats.csv / r01_train_kernel_stats.csv.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bert_stats -- python3 tools/exp_bert_graph.py > gpurun_out/prof_bert_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_stats -- python3 tools/exp_roberta_train.py --bf16 --fused > gpurun_out/prof_train_stats.log 2>&1
grep "replay ms" gpurun_out/prof_bert_stats.log; tail -2 gpurun_out/prof_train_stats.log

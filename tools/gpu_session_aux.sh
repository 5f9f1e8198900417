#!/bin/bash
# Kernel-trace summaries of the two secondary configurations: BERT-base QA batch (replayed graph) and the RoBERTa-base
# training step (replayed graph).  Copied into profiles/ as rNN_bert_kernel_stats.csv / rNN_train_kernel_stats.csv (tools/collect_profiles.py); the un-profiled timings
# of the same two scripts as rNN_bert_batch.txt / rNN_train_step.txt.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/exp_bert_graph.py > gpurun_out/bert_batch.txt 2>&1
QT_FP8_ATTENTION_KERNEL=0 python tools/exp_bert_graph.py 2>&1 | grep "replay ms" | sed 's/^/bf16 attention kernel (QT_FP8_ATTENTION_KERNEL=0): /' >> gpurun_out/bert_batch.txt
python tools/exp_roberta_train.py --bf16 --fused > gpurun_out/train_step.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bert_stats -- python3 tools/exp_bert_graph.py > gpurun_out/prof_bert_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_stats -- python3 tools/exp_roberta_train.py --bf16 --fused > gpurun_out/prof_train_stats.log 2>&1
grep "replay ms" gpurun_out/prof_bert_stats.log; tail -2 gpurun_out/prof_train_stats.log
find gpurun_out/prof_bert_stats gpurun_out/prof_train_stats -name "*kernel_trace.csv" -delete
cat gpurun_out/bert_batch.txt | tail -3

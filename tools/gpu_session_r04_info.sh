#!/bin/bash
# Round 4, first session: per-step launch lists of the two secondary workloads the round works on (configs[4] training step,
# configs[1] BERT batch), taken from rocprofv3 kernel traces of the replayed graphs.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_train4 -- python3 bench.py --workload roberta-mrpc-int8-e5m2-train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_train4.log 2>&1
tail -1 gpurun_out/prof_train4.log | head -c 300; echo
python tools/window_breakdown.py gpurun_out/prof_train4 --windows 3 --layers 1 --anchor scale_update_multi_kernel > gpurun_out/train_step_breakdown.txt 2>&1
head -70 gpurun_out/train_step_breakdown.txt
find gpurun_out/prof_train4 -name "*kernel_trace.csv" -delete
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_bert4 -- python3 bench.py --workload bert-base-squad-e4m3 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_bert4.log 2>&1
tail -1 gpurun_out/prof_bert4.log | head -c 300; echo
python tools/window_breakdown.py gpurun_out/prof_bert4 --windows 3 --layers 12 --anchor attention_fp8_split_kernel > gpurun_out/bert_batch_breakdown.txt 2>&1
head -40 gpurun_out/bert_batch_breakdown.txt
find gpurun_out/prof_bert4 -name "*kernel_trace.csv" -delete

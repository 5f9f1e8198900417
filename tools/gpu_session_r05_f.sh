#!/bin/bash
mkdir -p gpurun_out
OUT=gpurun_out/r05_session_f.txt
{
echo "== parity"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chain or train_kernels or softmax" 2>&1 | tail -15
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_blocks_golden.py -x -q -m gpu -s -k "full_size_roberta or training_chains or graphed_training or toy or bert" 2>&1 | grep -E "one step|3 steps|passed|failed|Error|assert|^E " | head -60
echo "== training step A/B"
timeout 900 python tools/ab_env.py --workload roberta-mrpc-int8-e5m2-train --reps 1 --steps 10 "" QT_TRAIN_CHAINS=0
echo "== trace"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_stats -- python3 bench.py --workload roberta-mrpc-int8-e5m2-train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_train4.log 2>&1
python tools/window_breakdown.py gpurun_out/prof_train_stats --windows 3 --layers 1 --anchor scale_update_multi_kernel > gpurun_out/r05_train_step_breakdown.txt 2>&1
python tools/step_sequence.py gpurun_out/prof_train_stats > gpurun_out/r05_train_step_sequence.txt 2>&1
find gpurun_out/prof_train_stats -name "*kernel_trace.csv" -delete
head -60 gpurun_out/r05_train_step_breakdown.txt
} > $OUT 2>&1
cat $OUT | cut -c1-220

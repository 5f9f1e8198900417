#!/bin/bash
# GPU session (round 5, third): training-step chains -- parity, then the step's time with / without them, then a kernel trace
mkdir -p gpurun_out
OUT=gpurun_out/r05_session_c.txt
TUNE=$PWD/tools/build/libqt_hip_tuning.so
{
echo "== parity"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chain or splitk_scratch or lt_fp8" 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_blocks_golden.py -x -q -m gpu -s -k "training or train or roberta or quantize_toy" 2>&1 | tail -12
echo "== training step A/B"
timeout 900 python tools/ab_env.py --workload roberta-mrpc-int8-e5m2-train --reps 2 --steps 10 "" QT_TRAIN_CHAINS=0 QT_TRAIN_COLSUM=0 "QT_HIP_LIB=$TUNE QT_OBS_UNR=1" "QT_HIP_LIB=$TUNE QT_OBS_UNR=1 QT_TRAIN_CHAINS=0"
echo "== trace"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_stats -- python3 bench.py --workload roberta-mrpc-int8-e5m2-train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_train4.log 2>&1
python tools/window_breakdown.py gpurun_out/prof_train_stats --windows 3 --layers 1 --anchor scale_update_multi_kernel > gpurun_out/r05_train_step_breakdown.txt 2>&1
python tools/step_sequence.py gpurun_out/prof_train_stats > gpurun_out/r05_train_step_sequence.txt 2>&1
find gpurun_out/prof_train_stats -name "*kernel_trace.csv" -delete
head -70 gpurun_out/r05_train_step_breakdown.txt
} > $OUT 2>&1
grep -E "^==|passed|failed|rror|roberta|^\[|launches" $OUT | cut -c1-250

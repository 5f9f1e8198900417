#!/bin/bash
mkdir -p gpurun_out
W="--workload roberta-mrpc-int8-e5m2-train --steps 5 --warmup 2 --no-roofline --no-cpu-baseline"
ms() { tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
{
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "attention_train or fanin or embedding" 2>&1 | tail -25
timeout 1500 python -m pytest tests/test_gpu_models.py -q -m gpu -k "training or train" 2>&1 | tail -40
for r in 1 2; do
for v in "QT_TRAIN_EMBEDDING=1" "QT_TRAIN_EMBEDDING=0"; do
echo "== $v"; env $v python bench.py $W 2>&1 | ms
done
done
rm -rf gpurun_out/prof_train_stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_stats -- python3 bench.py --workload roberta-mrpc-int8-e5m2-train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_train4.log 2>&1
python tools/window_breakdown.py gpurun_out/prof_train_stats --windows 3 --layers 1 --anchor scale_update_multi_kernel > gpurun_out/train_j_breakdown.txt 2>&1
python tools/step_sequence.py gpurun_out/prof_train_stats > gpurun_out/train_j_sequence.txt 2>&1
find gpurun_out/prof_train_stats -name "*kernel_trace.csv" -delete
head -34 gpurun_out/train_j_breakdown.txt
} > gpurun_out/r05_session_j.txt 2>&1
cut -c1-300 gpurun_out/r05_session_j.txt

#!/bin/bash
mkdir -p gpurun_out
W="--workload roberta-mrpc-int8-e5m2-train --steps 5 --warmup 2 --no-roofline --no-cpu-baseline"
ms() { tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
{
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "embedding or layernorm_train or fanin" 2>&1 | tail -15
timeout 1500 python -m pytest tests/test_gpu_models.py -q -m gpu -k "training or train" 2>&1 | tail -30
for r in 1 2 3; do
echo "== default"; python bench.py $W 2>&1 | ms
done
rm -rf gpurun_out/prof_train_stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_stats -- python3 bench.py --workload roberta-mrpc-int8-e5m2-train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_train4.log 2>&1
python tools/window_breakdown.py gpurun_out/prof_train_stats --windows 3 --layers 1 --anchor scale_update_multi_kernel > gpurun_out/train_k_breakdown.txt 2>&1
find gpurun_out/prof_train_stats -name "*kernel_trace.csv" -delete
head -24 gpurun_out/train_k_breakdown.txt
} > gpurun_out/r05_session_k.txt 2>&1
cut -c1-300 gpurun_out/r05_session_k.txt

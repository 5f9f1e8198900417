#!/bin/bash
mkdir -p gpurun_out
W="--workload roberta-mrpc-int8-e5m2-train --steps 5 --warmup 2 --no-roofline --no-cpu-baseline"
ms() { tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['workload'][-60:])"; }
{
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "layernorm_train" 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_gpu_models.py -q -m gpu -k "training or train" 2>&1 | tail -20
for r in 1 2; do
for v in "QT_TRAIN_ADDLN=1" "QT_TRAIN_ADDLN=0"; do
echo "== $v"; env $v python bench.py $W 2>&1 | ms
done
done
echo "== dropout 0.1"; QT_BENCH_DROPOUT=0.1 python bench.py $W 2>&1 | ms
} > gpurun_out/r05_session_k.txt 2>&1
cut -c1-300 gpurun_out/r05_session_k.txt

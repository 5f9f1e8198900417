#!/bin/bash
# TunableOp experiment on the configs[4] training step: tune torch's bf16 GEMM choice per shape once, replay with the file.
mkdir -p gpurun_out
W="--workload roberta-mrpc-int8-e5m2-train --steps 5 --warmup 2 --no-roofline --no-cpu-baseline"
F=gpurun_out/tunableop_train.csv
{
echo "== tuning run"
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$F PYTORCH_TUNABLEOP_ROCBLAS_ENABLED=0 \
PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=10 PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=30 PYTORCH_TUNABLEOP_VERBOSE=0 \
  timeout 2400 python bench.py $W 2>&1 | tail -3 | cut -c1-400
ls -la gpurun_out/tunableop_train*
for r in 1 2; do
echo "== default"
python bench.py $W 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
echo "== tuned file, tuning off"
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_FILENAME=$F \
  python bench.py $W 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
} > gpurun_out/r05_session_i.txt 2>&1
cut -c1-400 gpurun_out/r05_session_i.txt

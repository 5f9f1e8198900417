#!/bin/bash
mkdir -p gpurun_out
W="--workload roberta-mrpc-int8-e5m2-train --steps 5 --warmup 2 --no-roofline --no-cpu-baseline"
ms() { tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['workload'][-60:])"; }
{
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "attention_train" 2>&1 | tail -5
QT_HIP_LIB=tools/build/libqt_hip_tuning.so timeout 600 python tools/exp_attention_train.py 2>&1 | grep -v Warn | cut -c1-260
for r in 1 2; do
echo "== default"; python bench.py $W 2>&1 | ms
done
timeout 1500 python -m pytest tests/test_gpu_models.py -q -m gpu -k "training or train" 2>&1 | tail -5
} > gpurun_out/r05_session_k.txt 2>&1
cut -c1-300 gpurun_out/r05_session_k.txt

#!/bin/bash
mkdir -p gpurun_out
W="--workload roberta-mrpc-int8-e5m2-train --steps 5 --warmup 2 --no-roofline --no-cpu-baseline"
ms() { tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
{
for r in 1 2; do
for g in 192 288 384 768 1536; do
echo "== tuning build QT_CHAIN_WGS_PRE=$g"; QT_HIP_LIB=tools/build/libqt_hip_tuning.so QT_CHAIN_WGS_PRE=$g python bench.py $W 2>&1 | ms
done
done
} > gpurun_out/r05_session_k.txt 2>&1
cut -c1-300 gpurun_out/r05_session_k.txt

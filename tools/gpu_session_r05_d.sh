#!/bin/bash
mkdir -p gpurun_out
OUT=gpurun_out/r05_session_d.txt
{
echo "== chain microbenchmark"
timeout 600 python tools/exp_chain.py
echo "== parity"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chain" 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_models.py -x -q -m gpu -s -k "training_chains or full_size_roberta" 2>&1 | tail -80
echo "== training step A/B"
timeout 900 python tools/ab_env.py --workload roberta-mrpc-int8-e5m2-train --reps 2 --steps 10 "" QT_TRAIN_CHAINS=0 QT_TRAIN_COLSUM=0
} > $OUT 2>&1
cat $OUT | cut -c1-250

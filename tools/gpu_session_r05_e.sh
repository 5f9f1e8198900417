#!/bin/bash
mkdir -p gpurun_out
OUT=gpurun_out/r05_session_e.txt
{
echo "== parity"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chain or linear_fqt" 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_models.py -x -q -m gpu -s -k "full_size_roberta" 2>&1 | grep -E "one step|3 steps|scale cpu|passed|failed|Error"
echo "== chain microbenchmark"
timeout 600 python tools/exp_chain.py
} > $OUT 2>&1
cat $OUT | cut -c1-250

#!/bin/bash
mkdir -p gpurun_out
OUT=gpurun_out/r05_session_e.txt
{
echo "== parity"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chain" 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_blocks_golden.py -x -q -m gpu -s -k "full_size_roberta or training_chains or graphed_training or toy" 2>&1 | grep -E "one step|3 steps|passed|failed|Error|assert"


} > $OUT 2>&1
cat $OUT | cut -c1-250

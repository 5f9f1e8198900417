#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_blocks_golden.py -x -q -m gpu -k "training or train or roberta or toy" 2>&1 | grep -E "passed|failed|Error|assert|^E " | head -12
} > gpurun_out/r05_session_g.txt 2>&1
cut -c1-900 gpurun_out/r05_session_g.txt

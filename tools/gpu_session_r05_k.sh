#!/bin/bash
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_parity.py -q -m gpu -k "fanin" 2>&1 | tail -30
} > gpurun_out/r05_session_k.txt 2>&1
cut -c1-300 gpurun_out/r05_session_k.txt

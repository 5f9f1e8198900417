#!/bin/bash
mkdir -p gpurun_out
{
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
} > gpurun_out/gpu_tests.txt 2>&1
cut -c1-300 gpurun_out/gpu_tests.txt

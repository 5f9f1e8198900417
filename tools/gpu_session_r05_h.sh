#!/bin/bash
mkdir -p gpurun_out
{
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12
} > gpurun_out/r05_session_h.txt 2>&1
cut -c1-300 gpurun_out/r05_session_h.txt

#!/bin/bash
mkdir -p gpurun_out
TUNE=$PWD/tools/build/libqt_hip_tuning.so
{
for w in 96 192 384; do for l in 0 1; do echo "== QT_CHAIN_WGS=$w QT_CHAIN_LDS=$l"; QT_HIP_LIB=$TUNE QT_CHAIN_WGS=$w QT_CHAIN_LDS=$l timeout 300 python tools/exp_chain.py 2>&1 | grep -v amdgpu | cut -c1-110; done; done
} > gpurun_out/r05_chain_geometry.txt 2>&1
cat gpurun_out/r05_chain_geometry.txt

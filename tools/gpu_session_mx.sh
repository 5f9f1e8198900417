#!/bin/bash
# Kernel-trace profile of the block-scaled path (MX GEMM + fused quantize_mx); summaries land in gpurun_out/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mx_gemm -- python3 tools/exp_mx_gemm.py > gpurun_out/mx_gemm.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mx_layer -- python3 tools/exp_mx_linear.py > gpurun_out/mx_linear.log 2>&1
python3 tools/exp_mx_quant.py > gpurun_out/mx_quant.log 2>&1
python3 tools/exp_mx_wide.py > gpurun_out/mx_wide.log 2>&1
tools/build/probe_tile_fetch > gpurun_out/tile_fetch.log 2>&1
find gpurun_out/prof_mx_gemm gpurun_out/prof_mx_layer -name "*kernel_trace.csv" -delete
grep -v amdgpu gpurun_out/mx_gemm.log | tail -6
grep -v amdgpu gpurun_out/mx_linear.log | tail -8

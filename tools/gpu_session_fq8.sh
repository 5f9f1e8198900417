#!/bin/bash
# GPU session: fused FP8 Linear GEMM experiments (tools/exp_linear_fq8.py) under the tuning switches of qt_linear_fq8.hip
mkdir -p gpurun_out
run() { echo "== $1"; shift; env "$@" timeout 300 python tools/exp_linear_fq8.py --skip-checks --shapes ${SHAPES:-llama}; }
{
echo "== checks"; timeout 600 python -u tools/exp_linear_fq8.py --shapes probe
run "default"
run "no stagger" QT_FQ8_DEBUG=128
run "no compute" QT_FQ8_DEBUG=2
} > gpurun_out/fq8_session.txt 2>&1
grep -E "^==|bench|exact|accuracy|CHECKS|rror|fault" gpurun_out/fq8_session.txt | cut -c1-118

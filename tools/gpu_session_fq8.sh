#!/bin/bash
# GPU session: fused FP8 Linear GEMM experiments (tools/exp_linear_fq8.py) under the tuning switches of qt_linear_fq8.hip
mkdir -p gpurun_out
run() { echo "== $1"; shift; env "$@" timeout 300 python tools/exp_linear_fq8.py --skip-checks --shapes ${SHAPES:-llama}; }
{
echo "== checks"; timeout 600 python tools/exp_linear_fq8.py --shapes probe
run "default (11 groups at most)" QT_FQ8_MAX_NT=11
run "no warm-up loads" QT_FQ8_DEBUG=64 QT_FQ8_MAX_NT=11
run "no stagger" QT_FQ8_DEBUG=128 QT_FQ8_MAX_NT=11
run "no compute" QT_FQ8_DEBUG=2 QT_FQ8_MAX_NT=11
run "no compute, no warm-up" QT_FQ8_DEBUG=66 QT_FQ8_MAX_NT=11
} > gpurun_out/fq8_session.txt 2>&1
grep -E "^==|bench|exact|accuracy|CHECKS|Error|error" gpurun_out/fq8_session.txt | cut -c1-118

#!/bin/bash
# GPU session: fused FP8 Linear GEMM experiments (tools/exp_linear_fq8.py) under the tuning switches of qt_linear_fq8.hip
mkdir -p gpurun_out
run() { echo "== $1"; shift; env "$@" timeout 300 python tools/exp_linear_fq8.py --skip-checks --shapes ${SHAPES:-llama}; }
{
echo "== checks (variant 2: weights converted in registers)"; QT_FQ8_VARIANT=2 timeout 600 python -u tools/exp_linear_fq8.py --shapes probe
run "variant 2" QT_FQ8_VARIANT=2
run "variant 1 (raw bf16 weight tiles by LDS-DMA)" QT_FQ8_VARIANT=1
} > gpurun_out/fq8_session.txt 2>&1
grep -E "^==|bench|exact|accuracy|CHECKS|rror|fault" gpurun_out/fq8_session.txt | cut -c1-118

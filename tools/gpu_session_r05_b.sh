#!/bin/bash
# GPU session (round 5, second): fixed test, L2-resident-weights probe, library algorithm table, window A/B of the split-K down projection
mkdir -p gpurun_out
OUT=gpurun_out/r05_session_b.txt
TUNE=$PWD/tools/build/libqt_hip_tuning.so
{
echo "== parity (product library)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "linear_fq8 or mlp_fq8 or lt_fp8" 2>&1 | tail -5
b() { echo "== $1"; shift; env FUSED_ONLY=1 QT_HIP_LIB=$TUNE "$@" timeout 300 python tools/exp_linear_fq8.py --skip-checks --iters 100 --shapes ${SHAPES:-1024x11008x4096,1024x4096x11008}; }
b "tuning build defaults"
b "weights of every column tile = the first tile's (L2 hits)" QT_FQ8_DEBUG=256
b "the same, pool of one weight (Infinity Cache)" POOL=1
echo "== library algorithms"
timeout 900 python tools/tune_lt_algos.py > gpurun_out/r05_lt_algos.txt 2>&1; tail -30 gpurun_out/r05_lt_algos.txt
echo "== window A/B: split-K down projection"
timeout 900 python tools/ab_env.py --reps 2 "" QT_FQ8_SPLITK=0
} > $OUT 2>&1
grep -E "^==|bench|passed|failed|rror|fault|CHECKS|llama|_LT|^    \(" $OUT | cut -c1-220

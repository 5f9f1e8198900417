#!/bin/bash
# Round-end GPU session: whole -m gpu suite, headline bench + profiles (tools/gpu_session_bench.sh), and the
# LLaMA-2-13B posit8_2 leg of BASELINE.json's configs[1] with its kernel stats.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; tail -3 gpurun_out/pytest_gpu.txt
bash tools/gpu_session_bench.sh full
timeout 900 python bench.py --model llama-2-13b --activation posit8_2 --weight posit8_2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_13b_posit8_2.json 2> gpurun_out/bench_13b_posit8_2.err
cut -c1-700 gpurun_out/bench_13b_posit8_2.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_13b_posit -- python3 bench.py --model llama-2-13b --activation posit8_2 --weight posit8_2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_13b_posit.log 2>&1
python tools/window_breakdown.py gpurun_out/prof_13b_posit --windows 3 --layers 40 --anchor softmax > gpurun_out/window_breakdown_13b_posit.txt 2>&1
head -25 gpurun_out/window_breakdown_13b_posit.txt
find gpurun_out/prof_13b_posit -name "*kernel_trace.csv" -delete
# the fused FP8 GEMM against the pair it replaces (both variants) and the one-launch MLP front half
bash tools/gpu_session_fq8.sh > /dev/null 2>&1
timeout 600 python tools/exp_mlp_fq8.py > gpurun_out/mlp_fq8.txt 2>&1
grep -E "^bench" gpurun_out/fq8_session.txt | head -8 | cut -c1-120
timeout 600 python tools/exp_attention_fp8.py > gpurun_out/attention_fp8.txt 2>&1
# the two secondary configurations: BERT-base QA batch and the RoBERTa-base training step
bash tools/gpu_session_aux.sh

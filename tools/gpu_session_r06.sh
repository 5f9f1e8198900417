#!/bin/bash
# Round 6 GPU session (final profiles): headline bench (with the secondary legs) + kernel-trace profile + PMC passes, the three secondary
# workloads with per-step launch lists, the training step's ordered launch sequence, the chain microbenchmark, same-box A/B runs.
# tools/collect_profiles.py r06 copies the summaries into profiles/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_session_bench.sh full > gpurun_out/session_bench.txt 2>&1
head -c 1500 gpurun_out/bench_n1.json; echo
for w in llama-13b-posit8_2 bert-base-squad-e4m3 roberta-mrpc-int8-e5m2-train roberta-mrpc-int8-e5m2-train-dropout; do
  timeout 900 python bench.py --workload $w --steps 5 --warmup 2 > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  head -c 300 gpurun_out/bench_$w.json; echo
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_13b_posit -- python3 bench.py --workload llama-13b-posit8_2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_13b_posit.log 2>&1
python tools/window_breakdown.py gpurun_out/prof_13b_posit --windows 3 --layers 40 --anchor attention_rows > gpurun_out/window_breakdown_13b_posit.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_stats -- python3 bench.py --workload roberta-mrpc-int8-e5m2-train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_train4.log 2>&1
python tools/window_breakdown.py gpurun_out/prof_train_stats --windows 3 --layers 1 --anchor scale_update_multi_kernel > gpurun_out/train_step_breakdown.txt 2>&1
python tools/step_sequence.py gpurun_out/prof_train_stats > gpurun_out/train_step_sequence.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bert_stats -- python3 bench.py --workload bert-base-squad-e4m3 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_bert4.log 2>&1
python tools/window_breakdown.py gpurun_out/prof_bert_stats --windows 3 --layers 12 --anchor attention_fp8_split_kernel > gpurun_out/bert_batch_breakdown.txt 2>&1
find gpurun_out/prof_13b_posit gpurun_out/prof_train_stats gpurun_out/prof_bert_stats -name "*kernel_trace.csv" -delete
head -12 gpurun_out/window_breakdown_13b_posit.txt; head -8 gpurun_out/train_step_breakdown.txt; head -8 gpurun_out/bert_batch_breakdown.txt
# training step A/B on this box, the chain microbenchmark
# (QT_TRAIN_DEBUG 1920 = 128 + 256 + 512 + 1024: the launch structure of the round's first half -- single GEMM launches, torch's clip + optimizer)
timeout 2400 python tools/ab_env.py --workload roberta-mrpc-int8-e5m2-train --reps 2 --steps 10 "" QT_TRAIN_DEBUG=256 QT_TRAIN_DEBUG=512 QT_TRAIN_DEBUG=1024 QT_TRAIN_DEBUG=128 QT_TRAIN_DEBUG=1920 "QT_TRAIN_DEBUG=1920 QT_TRAIN_GEMM=0" QT_TRAIN_DEBUG=8 QT_TRAIN_DEBUG=4 QT_TRAIN_DEBUG=1 > gpurun_out/train_step_ab.txt 2>&1; cat gpurun_out/train_step_ab.txt
timeout 600 python tools/exp_train_stamps.py 12 > gpurun_out/train_stamps.txt 2>&1; tail -5 gpurun_out/train_stamps.txt
timeout 300 python tools/exp_graph_branches.py > gpurun_out/graph_branches.txt 2>&1; cat gpurun_out/graph_branches.txt
# value-map GEMM PMC passes (unchanged kernel: the traffic entry of the bench line's secondary roofline)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fqt -- python3 tools/roofline_fqt.py > gpurun_out/prof_fqt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_fqt -- python3 tools/roofline_fqt.py > gpurun_out/pmc_fetch_fqt.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_fqt -- python3 tools/roofline_fqt.py > gpurun_out/pmc_write_fqt.log 2>&1
find gpurun_out/prof_fqt -name "*kernel_trace.csv" -delete
# the reference's current flow (PT2E prepared graph, fused)
timeout 900 python bench.py --route pt2e --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/pt2e_bench.json 2> gpurun_out/pt2e_bench.err
head -c 300 gpurun_out/pt2e_bench.json; echo

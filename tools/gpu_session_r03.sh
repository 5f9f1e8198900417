#!/bin/bash
# Round 3 GPU session: headline bench + profiles, the other three bench workloads, the value-map GEMM (qt_linear_fqt_bf16) tables,
# ablations, in-kernel stamps and PMC passes, the row-form elementwise passes, the FP8 route table.  tools/collect_profiles.py r03
# copies the summaries into profiles/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_session_bench.sh full > gpurun_out/session_bench.txt 2>&1
head -c 600 gpurun_out/bench_n1.json; echo
for w in llama-13b-posit8_2 bert-base-squad-e4m3 roberta-mrpc-int8-e5m2-train; do
  timeout 900 python bench.py --workload $w --steps 5 --warmup 2 > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  head -c 300 gpurun_out/bench_$w.json; echo
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_13b_posit -- python3 bench.py --workload llama-13b-posit8_2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_13b_posit.log 2>&1
python tools/window_breakdown.py gpurun_out/prof_13b_posit --windows 3 --layers 40 --anchor attention_fq > gpurun_out/window_breakdown_13b_posit.txt 2>&1
head -25 gpurun_out/window_breakdown_13b_posit.txt
find gpurun_out/prof_13b_posit -name "*kernel_trace.csv" -delete
# the value-map GEMM: parity checks + timing against the pair at the 13B / 7B / BERT shapes, ablations, stamps, profiler passes
timeout 900 python tools/exp_linear_fqt.py --iters 30 --shapes all > gpurun_out/fqt_gemm.txt 2>&1
grep -E "^bench|CHECKS" gpurun_out/fqt_gemm.txt | cut -c1-150
(for a in 0 1 2 5 6 7 10; do echo "QT_FQT_ABLATE=$a"; QT_FQT_ABLATE=$a timeout 200 python tools/exp_linear_fqt.py --iters 20 --shapes one --skip-checks 2>&1 | grep bench; done) > gpurun_out/fqt_ablate.txt 2>&1
timeout 300 python tools/exp_fqt_stamps.py > gpurun_out/fqt_stamps.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fqt -- python3 tools/roofline_fqt.py > gpurun_out/prof_fqt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_fqt -- python3 tools/roofline_fqt.py > gpurun_out/pmc_fetch_fqt.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_fqt -- python3 tools/roofline_fqt.py > gpurun_out/pmc_write_fqt.log 2>&1
find gpurun_out/prof_fqt -name "*kernel_trace.csv" -delete
# table formats: the row form against the LDS table
timeout 600 python tools/exp_table_formats.py > gpurun_out/table_formats.txt 2>&1
grep -E "posit8_2|per-channel" gpurun_out/table_formats.txt | head -8
# the FP8 route table
(python tools/exp_linear_fq8.py --skip-checks --shapes all --iters 30; python tools/exp_linear_fq8.py --skip-checks --shapes sweep --iters 30) 2>&1 | grep bench > gpurun_out/fq8_routes.txt
# the reference's current flow: PT2E prepared graph on the fused kernels (bench --route pt2e)
timeout 900 python bench.py --route pt2e --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/pt2e_bench.json 2> gpurun_out/pt2e_bench.err
head -c 400 gpurun_out/pt2e_bench.json; echo
# what bounds the fused FP8 GEMM: compile-time ablations of variant R
(for a in 0 1 2 3 4 5 6 7 8 0; do echo "QT_FQ8_ABLATE=$a"; QT_FQ8_ABLATE=$a timeout 200 python tools/exp_linear_fq8.py --iters 60 --shapes probe --skip-checks 2>&1 | grep bench | head -1; done) > gpurun_out/fq8_ablate.txt 2>&1
cat gpurun_out/fq8_ablate.txt | cut -c1-100

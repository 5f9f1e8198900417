mkdir -p gpurun_out
(
python -m pytest tests/test_gpu_parity.py -x -q -k "table_format_producers" 2>&1 | tail -15
python -m pytest tests/test_gpu_models.py -x -q -k "table_format_window" 2>&1 | tail -15
echo "== 13B posit8_2"; python bench.py --workload llama-13b-posit8_2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-220
echo "== 13B posit8_2 QT_FUSED_PRODUCER_MAP=0"; QT_FUSED_PRODUCER_MAP=0 python bench.py --workload llama-13b-posit8_2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-220
) > gpurun_out/map_producers.log 2>&1
cat gpurun_out/map_producers.log

mkdir -p gpurun_out
(
python -m pytest tests/test_gpu_models.py -x -q -k "pt2e_prepared_route_table or table_format_window" 2>&1 | grep -E "^E |passed|failed|Error" | head -20
python -m pytest tests/test_gpu_parity.py -x -q -k "table_format_producers" 2>&1 | tail -2
echo "== 13B posit8_2 pt2e"; python bench.py --workload llama-13b-posit8_2 --route pt2e --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/pt2e13.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['mean_window_nll'], d['config'].get('pt2e_fusions'), d['config']['routes'])"
grep -v "Warn\|warn" gpurun_out/pt2e13.err | grep -i "error\|Traceback" | head
echo "== 13B posit8_2 eager"; python bench.py --workload llama-13b-posit8_2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['mean_window_nll'])"
) > gpurun_out/map_producers.log 2>&1
cat gpurun_out/map_producers.log

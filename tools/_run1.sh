mkdir -p gpurun_out
(
python -m pytest tests/test_gpu_models.py -x -q -k "table_format or pt2e_prepared_route_table" 2>&1 | grep -E "^E |passed|failed|Error" | head
python -m pytest tests/test_gpu_parity.py -x -q -k "attention" 2>&1 | tail -2
echo "== 13B posit8_2 eager"; python bench.py --workload llama-13b-posit8_2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['mean_window_nll'], d['config']['fake_quant_calls_per_step'], d['config']['elements_per_step'])"
echo "== 13B posit8_2 eager, QT_FUSED_PRODUCER_MAP=0"; QT_FUSED_PRODUCER_MAP=0 python bench.py --workload llama-13b-posit8_2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['mean_window_nll'], d['config']['fake_quant_calls_per_step'], d['config']['elements_per_step'])"
) > gpurun_out/attn_out.log 2>&1
cat gpurun_out/attn_out.log

mkdir -p gpurun_out
(
python -m pytest tests/test_gpu_models.py -x -q -s -k "pt2e_prepared" 2>&1 | tail -40
) > gpurun_out/pt2e_route.log 2>&1
grep -v "Warning\|warn" gpurun_out/pt2e_route.log | tail -50

mkdir -p gpurun_out
(
python -m pytest tests/test_gpu_models.py -x -q 2>&1 | grep -E "^E |Error|error" | head -30
) > gpurun_out/attn13b.log 2>&1
cat gpurun_out/attn13b.log

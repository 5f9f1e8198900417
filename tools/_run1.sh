mkdir -p gpurun_out
(
python -m pytest tests/test_gpu_parity.py -q -x -k "row_form_pass or fake_quant_rows or strided" 2>&1 | tail -5
python tools/exp_linear_fqt.py --skip-checks --iters 40 --shapes 1024x12288x5120,1024x13824x5120,1024x14336x5120,1024x15360x5120,1024x16384x5120,1024x13824x5120,1024x14336x5120
) > gpurun_out/ntw_probe.log 2>&1
tail -30 gpurun_out/ntw_probe.log

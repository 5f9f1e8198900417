mkdir -p gpurun_out
(
echo "== RX checks"; QT_FQ8_R2_ABLATE=30 timeout 300 python tools/exp_linear_fq8.py --iters 40 --shapes 1024x4096x11008,1024x4096x4096,1024x4096x4224 2>&1 | grep -E "exact x=e4m3 w=e4m3|accuracy|CHECKS|bench" | cut -c1-120
echo "== default"; python tools/exp_linear_fq8.py --skip-checks --iters 40 --shapes 1024x4096x11008,1024x4096x4096,1024x4096x4224 2>&1 | grep bench | cut -c1-100
echo "== RX again"; QT_FQ8_R2_ABLATE=30 timeout 300 python tools/exp_linear_fq8.py --skip-checks --iters 40 --shapes 1024x4096x11008,1024x4096x4096,1024x4096x4224 2>&1 | grep bench | cut -c1-100
) > gpurun_out/fq8_rx.log 2>&1
cat gpurun_out/fq8_rx.log

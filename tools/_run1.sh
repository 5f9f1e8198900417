mkdir -p gpurun_out
(
for abl in 0 7 8 3 2 0; do echo "== ABLATE $abl"; QT_FQ8_ABLATE=$abl python tools/exp_linear_fq8.py --skip-checks --iters 60 --shapes probe 2>&1 | grep bench | head -2 | cut -c1-90; done
) > gpurun_out/fq8_abl.log 2>&1
cat gpurun_out/fq8_abl.log

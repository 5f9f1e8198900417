mkdir -p gpurun_out
(
for abl in 0 5 6 1 2 3 4 0; do echo "== ABLATE $abl"; QT_FQ8_ABLATE=$abl python tools/exp_linear_fq8.py --skip-checks --iters 60 --shapes probe 2>&1 | grep bench | cut -c1-90; done
) > gpurun_out/fq8_abl.log 2>&1
cat gpurun_out/fq8_abl.log

mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/suite.log 2>&1
tail -15 gpurun_out/suite.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>gpurun_out/b1.err | tail -1 | cut -c1-330; tail -3 gpurun_out/b1.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-graph 2>/dev/null | tail -1 | cut -c1-330
QT_FUSED_SOFTMAX=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-330

#!/bin/bash
# GPU session: table-format kernels before / after the half-table and per-channel LDS-table variants
mkdir -p gpurun_out
{
timeout 300 python tools/exp_table_formats.py
QT_LUT_HALF=0 QT_PC_LDS=0 timeout 300 python tools/exp_table_formats.py
} > gpurun_out/table_formats.txt 2>&1
cat gpurun_out/table_formats.txt | cut -c1-150

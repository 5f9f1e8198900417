#!/bin/bash
mkdir -p gpurun_out
{
QT_HIP_LIB=tools/build/libqt_hip_tuning.so timeout 600 python tools/exp_attention_train.py 2>&1 | grep -v Warn
} > gpurun_out/r05_session_k.txt 2>&1
cut -c1-600 gpurun_out/r05_session_k.txt

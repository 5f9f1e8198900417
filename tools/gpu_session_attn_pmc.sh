#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_attn1 -- python3 tools/exp_attention.py > gpurun_out/pmc_attn1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmc_attn2 -- python3 tools/exp_attention.py > gpurun_out/pmc_attn2.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_attn3 -- python3 tools/exp_attention.py > gpurun_out/pmc_attn3.log 2>&1
ls gpurun_out/pmc_attn1/*/ | head -3

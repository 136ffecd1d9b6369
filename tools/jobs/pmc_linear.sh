#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_lin -- python3 $R/tools/conv_one.py 256 256 188800 1 1 1 1 > $R/gpurun_out/pmc_lin.log 2>&1
echo rc $?
timeout 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc_lin2 -- python3 $R/tools/conv_one.py 256 256 188800 1 1 1 1 > $R/gpurun_out/pmc_lin2.log 2>&1
echo rc $?
timeout 150 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $R/gpurun_out/pmc_lin3 -- python3 $R/tools/conv_one.py 256 256 188800 1 1 1 1 > $R/gpurun_out/pmc_lin3.log 2>&1
echo rc $?

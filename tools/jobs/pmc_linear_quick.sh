#!/bin/bash
# conflicts + time of the K = 256 Linear after a change, then conv tests and the bench line
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_conv3d.py -x -q > gpurun_out/conv_tests.log 2>&1; echo tests rc $?
timeout 120 python tools/linear_bench.py > gpurun_out/linear_bench.log 2>&1; echo lin rc $?
timeout 120 python tools/conv_bench.py > gpurun_out/conv_bench.log 2>&1; echo conv rc $?
timeout 300 python bench.py > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; echo bench rc $?
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_lin -- python3 $R/tools/conv_one.py 256 256 188800 1 1 1 1 > $R/gpurun_out/pmc_lin.log 2>&1
echo rc $?

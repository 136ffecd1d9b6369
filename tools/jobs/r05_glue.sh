#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python tools/train_step_bench.py --steps 5 --glue 2>&1 | grep -v "amdgpu.ids\|warn\|Warning" | cut -c1-220 > gpurun_out/r05_train_step_glue.txt
timeout 900 python tools/train_step_bench.py --steps 10 --profile 2>&1 | grep -v "amdgpu.ids\|warn\|Warning" | cut -c1-200 > gpurun_out/r05_train_step_kernels2.txt
head -70 gpurun_out/r05_train_step_glue.txt; head -30 gpurun_out/r05_train_step_kernels2.txt

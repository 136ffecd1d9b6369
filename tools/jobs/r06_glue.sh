#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 600 python tools/train_step_bench.py --steps 10 --glue 2>&1 | grep -v "amdgpu.ids\|warn\|Warning" | cut -c1-260 > gpurun_out/r06_train_step_glue_a.txt
head -90 gpurun_out/r06_train_step_glue_a.txt

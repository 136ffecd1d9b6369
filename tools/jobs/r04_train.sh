#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/train_step_bench.py --steps 10 --profile > gpurun_out/r04_train_step.json 2> gpurun_out/r04_train_step_kernels.raw; echo rc $?
grep -v "amdgpu.ids\|warn\|Warning" gpurun_out/r04_train_step_kernels.raw | cut -c1-200 > gpurun_out/r04_train_step_kernels.txt
cat gpurun_out/r04_train_step.json; head -32 gpurun_out/r04_train_step_kernels.txt
timeout 600 python tools/train_step_bench.py --steps 5 --glue 2>&1 | grep -v "amdgpu.ids\|warn\|Warning" | head -45 | cut -c1-200 > gpurun_out/r04_train_step_glue.txt; head -30 gpurun_out/r04_train_step_glue.txt

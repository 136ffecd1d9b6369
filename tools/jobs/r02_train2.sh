#!/bin/bash
python -m pytest tests/test_gpu_modules.py -x -q -k "autograd_on_hip or training" 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_train -o train -- python3 /root/repo/tools/train_step_bench.py --steps 3 > /dev/null 2>&1
cd /root/repo; f=$(ls gpurun_out/prof_train/*/train_kernel_stats.csv 2>/dev/null | head -1); head -25 $f | cut -c1-170

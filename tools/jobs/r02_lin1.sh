#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_conv3d.py -x -q -k "linear_rows_function" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -k "training or train" 2>&1 | tail -5
timeout 300 python tools/train_step_bench.py --steps 5 --profile 2>&1 | grep -v "amdgpu.ids\|warn\|Warning" | tail -16 | cut -c1-130

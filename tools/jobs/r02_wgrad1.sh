#!/bin/bash
timeout 300 python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_conv3d.py -x -q -k "wgrad or channels_last or linear_rows_function" 2>&1 | tail -3
timeout 300 python tools/train_step_bench.py --steps 5 2>&1 | tail -1

#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "view_attend" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -k "training or train" 2>&1 | tail -5
timeout 300 python tools/train_step_bench.py --steps 5 2>&1 | tail -1

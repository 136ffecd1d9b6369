#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -k "training or train" 2>&1 | tail -3
timeout 300 python tools/train_step_bench.py --steps 5 --profile 2>&1 | grep -v "amdgpu.ids\|warn\|Warning" | head -30 | cut -c1-150

#!/bin/bash
timeout 300 python tools/train_step_bench.py --steps 5 --profile 2>&1 | grep -v "amdgpu.ids\|warn\|Warning" | head -14 | cut -c1-190

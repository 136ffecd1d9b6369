#!/bin/bash
timeout 600 python bench.py --workload cfg4_scannet200_large --steps 8 --warmup 3 --no-cpu-baseline --no-strict-fp32 --sustain 0 --breakdown 2>&1 | grep -v amdgpu.ids | grep -v "^{" | cut -c1-150 | head -90

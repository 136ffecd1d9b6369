#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 300 python bench.py --steps 30 --warmup 5 2>gpurun_out/bench.err | tee gpurun_out/bench.json
timeout 300 python tools/host_profile.py 2>&1 | grep -E "^issue|^voxel_head|^neck"

#!/bin/bash
# round 4, item 1: sample coordinates with the reference's two roundings (no border mask in the stress test), and what the
# IEEE forms of offset / size, exp and the softmax division are worth in near-tie flips and in gather time
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -k "stress or tiled_gather or pairs or dfa3d or depth_score or wms" 2>&1 | tail -4
timeout 1200 python tools/tie_flips.py 6 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_tie_flips.jsonl
SGC_TILE_CONFIGS="16,22,3,3,0,0,1,0,0,1" timeout 300 python tools/tile_bench.py cfg2 64x80 ring 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_ieee_ab_cfg2.log
SGC_TILE_CONFIGS="27,30,3,3,1,0,1,0,0,0" timeout 300 python tools/tile_bench.py cfg4 64x80 ring 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_ieee_ab_cfg4.log

#!/bin/bash
mkdir -p gpurun_out
export SGC_TILE_DIAG=1
SGC_TILE_CONFIGS="27,32,3,3,0,16,1;27,32,3,3,0,8,1;16,22,3,3,0,8,1;16,22,3,3,0,16,1;16,22,3,3,1,16,1;40,22,2,2,0,16,1;16,11,3,3,1,8,1;27,16,3,3,1,16,1" timeout 600 python tools/tile_bench.py cfg2 64x80 ring > gpurun_out/r02_tile2_cfg2.log 2>&1; cat gpurun_out/r02_tile2_cfg2.log | grep -v "^  "
SGC_TILE_CONFIGS="27,11,3,3,1,8,1;27,11,3,3,1,16,1;16,59,3,3,1,16,1;27,22,3,3,1,16,1;40,30,3,3,1,16,1;80,16,3,3,1,16,1" timeout 600 python tools/tile_bench.py cfg4 59x80 ring > gpurun_out/r02_tile2_cfg4.log 2>&1; cat gpurun_out/r02_tile2_cfg4.log | grep -v "^  "
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-1500

#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "tiled or headmajor" 2>&1 | tail -3
export SGC_TILE_DIAG=1
C2="27,32,3,3,0,16,1,1,1;27,32,3,3,0,16,1,1,2;27,22,3,3,1,16,1,1,1;27,22,3,3,1,16,1,1,2;20,16,3,3,0,8,1,1,1;20,16,3,3,0,8,1,1,2;16,22,3,3,0,8,1,1,1;27,11,3,3,0,8,1,1,1;16,11,3,3,1,8,1,1,1;16,11,3,3,1,8,1,1,2;40,16,3,3,0,16,1,1,1;27,32,2,2,0,16,1,1,1;80,8,3,3,0,16,1,1,1;20,16,3,3,0,16,1,1,1;27,16,3,3,1,16,1,1,1;20,22,3,3,1,16,1,1,1"
SGC_TILE_CONFIGS="$C2" timeout 600 python tools/tile_bench.py cfg2 64x80 ring > gpurun_out/r02_tile4_cfg2.log 2>&1; grep -v "^  \|amdgpu.ids" gpurun_out/r02_tile4_cfg2.log | cut -c1-230
C4="27,30,3,3,1,16,1,1,2;27,30,3,3,1,16,1,1,1;27,30,3,3,1,16,1,1,4;40,30,3,3,1,16,1,1,2;27,59,3,3,1,16,1,1,2;27,11,3,3,1,8,1,1,1;27,11,3,3,1,8,1,1,2;20,15,3,3,1,8,1,1,2;27,20,3,3,1,16,1,1,2;20,30,3,3,1,16,1,1,2;27,15,3,3,1,16,1,1,2"
SGC_TILE_CONFIGS="$C4" timeout 600 python tools/tile_bench.py cfg4 59x80 ring > gpurun_out/r02_tile4_cfg4.log 2>&1; grep -v "^  \|amdgpu.ids" gpurun_out/r02_tile4_cfg4.log | cut -c1-230

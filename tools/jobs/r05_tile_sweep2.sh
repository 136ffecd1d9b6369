#!/bin/bash
# round 5: three workgroups per CU for the Cm = 32 tiled gather (windows <= 53 KB, 4 waves each)?
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
CFG="16,22,3,3,0,8,1,1,1,1;16,22,3,3,0,4,1,1,1,1;12,16,3,3,0,4,1,1,1,1;12,16,3,3,0,8,1,1,1,1;13,13,3,3,0,4,1,1,1,1;10,16,3,3,0,4,1,1,1,1;16,11,3,3,0,4,1,1,1,1;16,11,3,3,0,8,1,1,1,1;20,11,3,3,0,4,1,1,1,1;8,22,3,3,0,4,1,1,1,1;16,16,3,3,0,4,1,1,1,1"
SGC_DIAG=1 SGC_TILE_DIAG=1 SGC_TILE_CONFIGS="$CFG" timeout 600 python tools/tile_bench.py cfg2 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r05_tile_sweep2_cfg2.txt
cut -c1-250 gpurun_out/r05_tile_sweep2_cfg2.txt

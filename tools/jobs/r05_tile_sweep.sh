#!/bin/bash
# round 5: the tiled gather at config 2 -- does a (camera, bin) workgroup that walks several heads with two window buffers
# (fill of head h + 1 under the compute of head h) beat one head per workgroup with two workgroups per CU?
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
CFG="16,22,3,3,0,8,1,1,1,1;16,22,3,3,0,16,1,2,8,1;16,22,3,3,0,8,1,2,8,1;16,22,3,3,0,16,1,2,4,1;16,22,3,3,0,16,1,2,2,1;16,22,3,3,0,8,1,2,2,1;16,22,3,3,0,16,1,1,1,1;16,16,3,3,0,8,1,1,1,1;16,16,3,3,0,8,1,2,8,1;20,22,3,3,0,16,1,2,8,1;27,30,3,3,0,16,1,1,1,1"
SGC_DIAG=1 SGC_TILE_DIAG=1 SGC_TILE_CONFIGS="$CFG" timeout 600 python tools/tile_bench.py cfg2 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r05_tile_sweep_cfg2.txt
cat gpurun_out/r05_tile_sweep_cfg2.txt

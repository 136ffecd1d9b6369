#!/bin/bash
# Cm = 16 tiled gather with the per-head depth window: bins that let two workgroups share a CU (<= 80 KB each)
export SGC_TILE_DIAG=1 SGC_DIAG=1
SGC_TILE_CONFIGS="27,30,3,3,1,0,1,0,0,0;16,22,3,3,1,0,1,0,0,0;20,20,3,3,1,0,1,0,0,0;18,24,3,3,1,0,1,0,0,0;16,16,3,3,1,0,1,0,0,0;27,30,3,3,0,0,1,0,0,0;16,22,3,3,0,0,1,0,0,0;20,20,2,2,1,0,1,0,0,0" timeout 600 python tools/tile_bench.py cfg4 64x80 ring 2>&1 | grep -v amdgpu.ids

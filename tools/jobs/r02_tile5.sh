#!/bin/bash
SGC_TILE_DIAG=1 SGC_TILE_CONFIGS="16,22,3,3,0,0,1,1,0;16,22,3,3,0,0,1,2,0;16,22,3,3,0,8,1,1,0;16,22,3,3,0,16,1,1,0;16,22,3,3,0,0,1,1,2;16,11,3,3,0,0,1,1,0;16,11,3,3,0,0,1,2,0;27,22,3,3,0,0,1,1,0" python tools/tile_bench.py cfg2 2>&1 | grep -v amdgpu.ids | grep "tile bin\|wave kernel"

#!/bin/bash
for x in 0 1 0 1; do
echo "== tile_xcd $x cfg2"; SGC_TILE_XCD=$x SGC_TILE_CONFIGS="16,22,3,3,0,0,1,1,0;16,22,3,3,1,0,1,1,0;20,22,3,3,0,0,1,1,0" python tools/tile_bench.py cfg2 2>&1 | grep -v amdgpu.ids | grep "tile bin"
done
for x in 0 1; do
echo "== tile_xcd $x cfg4"; SGC_TILE_XCD=$x SGC_TILE_CONFIGS="27,30,3,3,1,0,1,1,0;27,30,3,3,0,0,1,1,0" python tools/tile_bench.py cfg4 2>&1 | grep -v amdgpu.ids | grep "tile bin"
done
python -m pytest tests/test_gpu_kernels.py -x -q -k "tile or tiled" 2>&1 | tail -3

#!/bin/bash
for sg in 0 1 2 3 4 0 2; do
echo "== stagger $sg"; SGC_TILE_STAGGER=$sg SGC_TILE_CONFIGS="16,22,3,3,0,0,1,1,0;16,22,3,3,0,0,1,1,2" python tools/tile_bench.py cfg2 2>&1 | grep -v amdgpu.ids | grep "tile bin" | cut -c1-150
done

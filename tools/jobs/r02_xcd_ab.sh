#!/bin/bash
C="16,22,3,3,0,0,1,1,0,0;16,22,3,3,0,0,1,1,0,1"
SGC_TILE_CONFIGS="$C;$C;$C;$C;$C" timeout 600 python tools/tile_bench.py cfg2 2>&1 | grep "tile bin" | cut -c60-140
D="27,30,3,3,1,0,1,1,0,0;27,30,3,3,1,0,1,1,0,1"
SGC_TILE_CONFIGS="$D;$D;$D;$D" timeout 600 python tools/tile_bench.py cfg4 80x80 2>&1 | grep "tile bin" | cut -c60-140

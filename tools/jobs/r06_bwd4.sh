#!/bin/bash
cd $GRAFT_REPO_ROOT
export SGC_DIAG=1
echo "--- shift on"; SGC_BWD_CONFIGS="8,11,2,2,4;8,11,2,2,4,17;8,16,2,2,4;8,16,2,2,4,17;12,11,2,2,4;8,16,3,2,4;8,22,2,2,4" timeout 900 python tools/bwd_tile_bench.py cfg2 2>&1 | grep "bins"
echo "--- shift off"; SGC_BWD_SHIFT=0 SGC_BWD_CONFIGS="8,11,2,2,4;8,11,2,2,4,17;8,16,2,2,4" timeout 900 python tools/bwd_tile_bench.py cfg2 2>&1 | grep "bins"

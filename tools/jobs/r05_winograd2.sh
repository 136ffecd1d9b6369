#!/bin/bash
# round 5: Winograd-z also on slices the 8 x 8 pixel bricks do not tile exactly (the 20 x 20 x 8 scale of config 2)?
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_conv3d.py -x -q -k "winograd" 2>&1 | tail -3
for rep in 1 2; do
for rg in 0 1; do
SGC_WINOGRAD_Z_RAGGED=$rg timeout 600 python bench.py --no-cpu-baseline --no-strict-fp32 --steps 60 --warmup 15 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cfg2 winograd ragged $rg:', d['value'], 'sustained', d['sustained']['value'], 'issued GF', d['path_roofline']['gemm_gflop_issued'], 'self_check', d['self_check']['mismatching'])"
done
done

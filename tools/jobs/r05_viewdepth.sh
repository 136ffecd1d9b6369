#!/bin/bash
# round 5: rows in flight per lane in the inter-view kernels (view_mean / view_attend / view_attend_pq): parity tests, then A/B per workload
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -x -q -k "view or attend or pq or hot_path or mean" 2>&1 | tail -3
for wl in cfg2_scannet cfg5_arkit_large cfg3_arkit cfg4_scannet200_large; do
n=${wl%%_*}
for rep in 1 2; do
for d in 1 4 8; do
SGC_TUNE=view_depth=$d timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-strict-fp32 --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n view_depth $d:', d['value'], 'sustained', d['sustained']['value'], 'self_check', d['self_check']['mismatching'])"
done
done
done

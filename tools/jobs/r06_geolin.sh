#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "geometry_sample_fused" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_modules.py -q -m gpu -x -k "test_hot_path_against_oracle" 2>&1 | tail -3
for rnd in 1 2; do for g in 0 1; do for wl in cfg2_scannet cfg3_arkit cfg2_scannet_100v; do
SGC_GEO_LINEAR=$g timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-strict-fp32 --steps 60 --warmup 15 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$wl geo_linear=$g', d['value'], 'sustained', d['sustained']['value'], 'gather', d['roofline']['frac'], 'self_check', d['self_check']['mismatching'])"
done; done; done 2>&1 | tee gpurun_out/r06_geo_linear_ab_c256.txt

#!/bin/bash
# round 5: reduction splits of the few-voxel layers with scenes in flight (throughput geometry): fewer splits = fewer epilogue launches and
# less workspace traffic, longer single workgroups.  alternated bench runs, explicit SGC_TUNE wins over the throughput defaults.
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for t in "256 96" "128 96" "256 48" "128 48" "64 32" "512 192"; do
set -- $t
SGC_TUNE="split_target=$1,halo_split_target=$2" timeout 600 python bench.py --no-cpu-baseline --no-strict-fp32 --steps 60 --warmup 15 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('split_target $1 halo_split_target $2:', d['value'], 'sustained', d['sustained']['value'], 'self_check', d['self_check']['mismatching'])"
done
done 2>&1 | tee gpurun_out/r05_split_sweep.txt

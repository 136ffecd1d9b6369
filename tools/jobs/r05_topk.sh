#!/bin/bash
# round 5: top-k of the 25 600-voxel level as ONE single-workgroup launch (keys in registers, topk_select_kernel<32>) against the 7-launch many-workgroup form
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for m in 16384 40000; do
SGC_TUNE=topk_multi_min=$m timeout 600 python bench.py --no-cpu-baseline --no-strict-fp32 --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cfg2 topk_multi_min $m:', d['value'], 'sustained', d['sustained']['value'], 'self_check', d['self_check']['mismatching'])"
done
done
for m in 16384 40000; do
SGC_TUNE=topk_multi_min=$m timeout 600 python bench.py --streams 1 --no-cpu-baseline --no-strict-fp32 --steps 60 --warmup 15 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cfg2 one stream topk_multi_min $m:', d['value'], 'self_check', d['self_check']['mismatching'])"
done

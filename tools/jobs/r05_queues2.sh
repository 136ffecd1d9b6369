#!/bin/bash
cd $GRAFT_REPO_ROOT
for q in 4 5 6 7 8 9 10; do
for n in 4; do
GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --streams $n --no-cpu-baseline --no-strict-fp32 --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cfg2 queues $q streams $n:', d['value'], 'sustained', d['sustained']['value'], 'self_check', d['self_check']['mismatching'])"
done
done

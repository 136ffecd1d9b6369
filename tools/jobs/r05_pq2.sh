#!/bin/bash
# round 5: projected-query attention at config 2 (40 views) again, now that its kernel keeps four rows in flight
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for pq in 0 1; do
SGC_PROJECTED_QUERY=$pq timeout 600 python bench.py --no-cpu-baseline --no-strict-fp32 --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cfg2 pq $pq:', d['value'], 'sustained', d['sustained']['value'], 'self_check', d['self_check']['mismatching'])"
done
done

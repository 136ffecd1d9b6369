#!/bin/bash
# round 5: scenes in flight re-swept on the final tree (the Winograd form changed the kernel mix)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for n in 3 4 5 6 8; do
timeout 600 python bench.py --streams $n --no-cpu-baseline --no-strict-fp32 --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cfg2 streams $n:', d['value'], 'sustained', d['sustained']['value'], 'self_check', d['self_check']['mismatching'])"
done
done

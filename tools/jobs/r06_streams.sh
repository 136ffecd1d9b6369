#!/bin/bash
cd $GRAFT_REPO_ROOT
for rnd in 1 2; do for s in 3 4 5 6 8; do
timeout 600 python bench.py --streams $s --no-cpu-baseline --no-strict-fp32 --steps 60 --warmup 15 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('streams $s', d['value'], 'sustained', d['sustained']['value'])"
done; done

#!/bin/bash
# break-even of the projected-query attention (V head by head) against the per-pair K | V form over the view count, config-2 shapes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_pq_views.txt
: > $out
run() {
  tag="$1"; v="$2"; shift 2
  env "$@" timeout 600 python bench.py --views $v --no-cpu-baseline --no-strict-fp32 --sustain 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('views %-4s %-34s' % ('$v', '$tag'), d['value'], 'sustained', d['sustained']['value'], 'self_check', d['self_check']['mismatching'])" | tee -a $out
}
for v in 8 12 16 20 30; do
  for rnd in 1 2; do
    run "per-pair K|V" $v SGC_PROJECTED_QUERY=0
    run "projected query" $v SGC_PROJECTED_QUERY=1
  done
done

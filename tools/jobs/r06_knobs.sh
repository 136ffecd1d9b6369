#!/bin/bash
# throughput-geometry knobs on the final tree (the mix of GEMMs changed with the projected-query form at 40 views), alternated
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_knobs_final.txt
: > $out
run() {
  SGC_TUNE="$1" timeout 600 python bench.py --no-cpu-baseline --no-strict-fp32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-40s' % '$1', d['value'], 'sustained', d['sustained']['value'], 'self_check', d['self_check']['mismatching'])" | tee -a $out
}
for rnd in 1 2; do
  run "rows_cu_pct=50"
  run "rows_cu_pct=75"
  run "rows_cu_pct=100"
  run "rows_depth=1"
  run "halo_split_target=192"
  run "halo_split_target=64"
done

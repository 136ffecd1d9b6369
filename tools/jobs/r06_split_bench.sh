#!/bin/bash
# headline run under the tile kernel's split policies, alternated (tools/split_free_ab.py has the layers alone)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_split_bench_ab.txt
: > $out
run() {
  SGC_TUNE="$1" timeout 600 python bench.py --no-cpu-baseline --no-strict-fp32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-40s' % '$1', d['value'], 'sustained', d['sustained']['value'], 'self_check', d['self_check']['mismatching'])" | tee -a $out
}
for rnd in 1 2 3; do
  run "split_free=0"
  run "split_free=1"
  run "split_free=1,split_target=384"
  run "split_free=1,split_target=512"
done

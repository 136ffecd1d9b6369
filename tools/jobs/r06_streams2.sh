#!/bin/bash
# scenes in flight (streams) on the final tree, alternated; and the fixed cost of a timed region (fill + drain) from its length
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_streams_final.txt
: > $out
run() {
  timeout 600 python bench.py --no-cpu-baseline --no-strict-fp32 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-36s' % '$*', d['value'], 'ms/step', d['ms_per_step'], 'steps', d['steps'], 'sustained', d['sustained']['value'])" | tee -a $out
}
for rnd in 1 2; do
  run --streams 3
  run --streams 4
  run --streams 5
  run --streams 6
done
run --steps 10 --warmup 5
run --steps 20 --warmup 5
run --steps 40 --warmup 5
run --steps 80 --warmup 5
run --steps 20 --warmup 5 --scenes-per-step 8

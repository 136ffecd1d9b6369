#!/bin/bash
# projected-query attention with its V projection head by head (SGC_PQ_BLOCKDIAG=1, round 6) against the dense 8C -> C GEMM (0), and --
# at config 2's 40 views -- against the per-pair K | V form (SGC_PROJECTED_QUERY=0): alternated bench runs on one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_pq_ab.txt
: > $out
run() {
  tag="$1"; wl="$2"; shift 2
  env "$@" timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-strict-fp32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-20s %-44s' % ('$wl', '$tag'), d['value'], 'sustained', d['sustained']['value'], 'self_check', d['self_check']['mismatching'])" | tee -a $out
}
for rnd in 1 2; do
  run "per-pair K|V (auto at 40 views)" cfg2_scannet SGC_PROJECTED_QUERY=0
  run "projected query, dense V" cfg2_scannet SGC_PROJECTED_QUERY=1 SGC_PQ_BLOCKDIAG=0
  run "projected query, V head by head" cfg2_scannet SGC_PROJECTED_QUERY=1 SGC_PQ_BLOCKDIAG=1
done
for wl in cfg2_scannet_100v cfg3_arkit cfg4_scannet200_large cfg5_arkit_large; do
  for rnd in 1 2; do
    run "projected query, dense V" $wl SGC_PQ_BLOCKDIAG=0
    run "projected query, V head by head" $wl SGC_PQ_BLOCKDIAG=1
  done
done
run "per-pair K|V" cfg3_arkit SGC_PROJECTED_QUERY=0

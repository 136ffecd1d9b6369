#!/bin/bash
# alternated bench runs under different SGC_TUNE settings (argument: ';'-separated settings, "-" = none), three rounds
R=$GRAFT_REPO_ROOT; cd $R
IFS=';' read -ra SETS <<< "${1:--}"
for rnd in 1 2 3; do
  for s in "${SETS[@]}"; do
    if [ "$s" == "-" ]; then t=""; else t="$s"; fi
    v=$(SGC_TUNE="$t" timeout 300 python bench.py --no-cpu-baseline --no-strict-fp32 --steps 200 --warmup 50 --sustain 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], (d.get('sustained') or {}).get('value'), d['self_check']['mismatching'])")
    echo "round $rnd  [$s]  $v"
  done
done

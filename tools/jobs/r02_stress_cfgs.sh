#!/bin/bash
for w in cfg3_arkit cfg5_arkit_large; do
SGC_SELF_CHECK_RUNS=300 timeout 1200 python bench.py --workload $w --no-cpu-baseline --no-strict-fp32 --sustain 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['config']['workload'][:24], d['value'], d['self_check']['scene_runs'], d['self_check']['mismatching'], d['path_roofline']['frac'], d['roofline']['frac'])"
done

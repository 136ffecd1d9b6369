#!/bin/bash
# overlapped replays against serial eager launches, bit for bit, 300 scene runs per workload (the bench's self-check, lengthened)
for w in cfg2_scannet cfg4_scannet200_large cfg5_arkit_large; do
SGC_SELF_CHECK_RUNS=300 timeout 1200 python bench.py --workload $w --no-cpu-baseline --no-strict-fp32 --sustain 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['config']['workload'][:24], d['value'], 'runs', d['self_check']['scene_runs'], 'mismatching', d['self_check']['mismatching'], 'path', d['path_roofline']['frac'], 'gather', d['roofline']['frac'])"
done
SGC_SELF_CHECK_RUNS=300 timeout 1200 python bench.py --conv-mode bf16 --no-cpu-baseline --sustain 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bf16 mode', d['value'], 'runs', d['self_check']['scene_runs'], 'mismatching', d['self_check']['mismatching'])"

#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_modules.py -x -q 2>&1 | tail -3
for w in cfg2_scannet cfg4_scannet200_large; do
timeout 600 python bench.py --workload $w --no-cpu-baseline --no-strict-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['config']['workload'][:30], d['value'], d['ms_per_step'], d['self_check']['mismatching'], d['sustained']['value'], d['roofline']['frac'])"
done

#!/bin/bash
SGC_SELF_CHECK_RUNS=2000 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('self_check'))"
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_conv3d.py -x -q -k "ring" 2>&1 | tail -1; done

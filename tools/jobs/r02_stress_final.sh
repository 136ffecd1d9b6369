#!/bin/bash
SGC_SELF_CHECK_RUNS=3000 timeout 900 python bench.py --no-cpu-baseline --no-strict-fp32 --sustain 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('default', d['value'], d['self_check'])"
SGC_TUNE="halo_ring=2,topk_multi_min=1" SGC_SELF_CHECK_RUNS=3000 timeout 900 python bench.py --no-cpu-baseline --no-strict-fp32 --sustain 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ring+multi-topk', d['value'], d['self_check'])"
SGC_SELF_CHECK_RUNS=1000 timeout 900 python bench.py --workload cfg4_scannet200_large --no-cpu-baseline --no-strict-fp32 --sustain 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg4', d['value'], d['self_check'])"
for i in 1 2; do timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -1; done

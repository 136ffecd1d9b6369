#!/bin/bash
# 2 000 overlapped scene runs (four streams, graph replays) compared bit for bit with serial eager launches, on the final tree
mkdir -p gpurun_out
SGC_SELF_CHECK_RUNS=2000 timeout 1500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps(dict(value=d['value'], ms_per_step=d['ms_per_step'], self_check=d.get('self_check'))))" | tee gpurun_out/r04_self_check_soak.json

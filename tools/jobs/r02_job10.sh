#!/bin/bash
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict-fp32 --sustain 0 --breakdown 2>&1 >/tmp/bench.json | grep -E "^  sgc_(topk|linear_rows_headmajor)" 
python -c "
import json
d=json.loads(open('/tmp/bench.json').read().strip().split('\n')[-1]); print(d['value'],'scenes/s', d['self_check']['mismatching'])"
timeout 600 python tools/train_step_bench.py 2>&1 | tail -12

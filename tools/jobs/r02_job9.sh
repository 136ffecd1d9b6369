#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "topk or headmajor or tiled" 2>&1 | tail -4
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict-fp32 --sustain 0 --breakdown 2>&1 >/tmp/bench.json | grep -E "^  sgc_(topk|linear_rows_headmajor|conv3d|linear_rows_bf16)" 
python -c "
import json
d=json.loads(open('/tmp/bench.json').read().strip().split('\n')[-1]); print(d['value'],'scenes/s', d['self_check']['mismatching'])"

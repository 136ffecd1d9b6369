#!/bin/bash
one() { timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-strict-fp32 --sustain 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$*', '|', d['value'], 'scenes/s', d['ms_per_step'], 'ms self_check', d['self_check']['mismatching'])"; }
one --streams 4
one --streams 3
one --streams 5
one --streams 6
one --streams 8
one --streams 4 --input-layout nhwc
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict-fp32 --sustain 0 --breakdown 2>&1 >/dev/null | grep -E "^  sgc_(topk|bin_pairs)"

#!/bin/bash
for s in 2 3 4 5 6 8; do
timeout 600 python bench.py --streams $s --no-cpu-baseline --no-strict-fp32 --sustain 0 --steps 300 --warmup 40 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('streams', $s, d['value'], d['ms_per_step'])"
done

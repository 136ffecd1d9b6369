#!/bin/bash
for s in 3 4 5 6 8; do
for rep in 1 2; do
timeout 600 python bench.py --streams $s --no-cpu-baseline --no-strict-fp32 --sustain 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('streams', $s, d['value'], d['ms_per_step'])"
done; done

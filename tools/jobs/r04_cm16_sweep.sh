#!/bin/bash
# Cm = 16 bins in the real bench (eager HIP-event timing of the finest-level gather)
for w in cfg4_scannet200_large cfg5_arkit_large; do
for c in 27,30,3,3,1 18,24,3,3,1 16,22,3,3,1 16,16,3,3,1 20,30,3,3,1 14,20,3,3,1 20,20,3,3,1; do
SGC_TILED_CM16=$c timeout 600 python bench.py --workload $w --no-cpu-baseline --steps 60 --warmup 10 --no-strict-fp32 --sustain 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$w', '$c', d['value'], 'gather', d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done; done

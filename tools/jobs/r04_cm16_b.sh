#!/bin/bash
for w in cfg4_scannet200_large cfg5_arkit_large cfg2_scannet; do
for c in 16,22,3,3,1 14,20,3,3,1; do
SGC_TILED_CM16=$c timeout 600 python bench.py --workload $w --no-cpu-baseline --steps 60 --warmup 10 --no-strict-fp32 --sustain 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$w', '$c', d['value'], 'gather', d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done; done
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "tiled" 2>&1 | tail -2

#!/bin/bash
for w in cfg4_scannet200_large cfg5_arkit_large; do
for t in tile_xcd=0 tile_xcd=1; do
SGC_TUNE=$t timeout 600 python bench.py --workload $w --no-cpu-baseline --steps 60 --warmup 10 --no-strict-fp32 --sustain 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$w', '$t', d['value'], 'gather', d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done; done
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "tiled or gather or pairs" 2>&1 | tail -2

#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
for cfg in "SGC_TILED=0" "SGC_TILED=1" "SGC_TILED_CM32=20,16,3,3" "SGC_TILED_CM32=27,32,3,3" "SGC_TILED_CM32=16,16,3,3"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
r=d['roofline']; print(d['value'], 'scenes/s', r['frac'], r['avg_launch_us'], 'us', r['kernel'][:40], 'self_check', d['self_check']['mismatching'])"
done
echo "== cfg4"
for cfg in "SGC_TILED=0" "SGC_TILED=1"; do
  env $cfg timeout 600 python bench.py --workload cfg4_scannet200_large --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
r=d['roofline']; print(d['value'], 'scenes/s', r['frac'], r['avg_launch_us'], 'us', r['kernel'][:40], 'self_check', d['self_check']['mismatching'])"
done

#!/bin/bash
# time of the finest-level tiled gather under the two XCD deals (eager roofline pass of bench.py), alternated
for rnd in 1 2; do
  for xcd in 0 1; do
    SGC_TUNE=tile_xcd=$xcd python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-strict-fp32 --sustain 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('round $rnd tile_xcd=$xcd:', d['roofline']['avg_launch_us'], 'us frac', d['roofline']['frac'], 'scenes/s', d['value'])"
  done
done

#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "tiled or headmajor" 2>&1 | tail -2
one() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
r=d['roofline']; print('$*', '|', d['value'], 'scenes/s frac', r['frac'], r['avg_launch_us'], 'us self_check', d['self_check']['mismatching'])"; }
one SGC_TILED=0
one SGC_TILED_CM32=27,32,3,3,0
one SGC_TILED_CM32=16,22,3,3,0
one SGC_TILED_CM32=20,16,3,3,0
one SGC_TILED_CM32=20,22,3,3,0
one SGC_TILED_CM32=27,16,3,3,0
one SGC_TILED_CM32=16,22,3,3,0 SGC_TUNE=tile_hg=2,tile_nbuf=2
one SGC_TILED_CM32=16,22,2,2,0
one SGC_TILED_CM32=27,32,2,2,0
one SGC_TILED_CM32=40,21,3,3,0
one SGC_TILED_CM32=27,32,3,3,0 SGC_TUNE=tile_hg=2
one SGC_TILED_CM32=16,32,3,3,0
one SGC_TILED_CM32=27,22,3,3,0

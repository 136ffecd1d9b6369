#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "tiled or headmajor" 2>&1 | tail -2
one() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
r=d['roofline']; print('$*', '|', d['value'], 'scenes/s frac', r['frac'], r['avg_launch_us'], 'us self_check', d['self_check']['mismatching'])"; }
one SGC_TILED=0
one SGC_TILED_CM32=27,32,3,3,0
one SGC_TILED_CM32=16,22,3,3,0
one SGC_TILED_CM32=20,16,3,3,0
one SGC_TILED_CM32=16,16,3,3,0
one SGC_TILED_CM32=16,22,3,3,0 SGC_TUNE=tile_nw=16
one SGC_TILED_CM32=13,22,3,3,0
one SGC_TILED_CM32=16,19,3,3,0
EXTRA="--workload cfg4_scannet200_large --steps 10 --warmup 3"
one SGC_TILED_CM16=27,30,3,3,1
one SGC_TILED_CM16=27,11,3,3,1
one SGC_TILED_CM16=27,20,3,3,1
one SGC_TILED_CM16=40,20,3,3,1
one SGC_TILED_CM16=20,15,3,3,1

#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bf16_storage or tiled" 2>&1 | tail -8
one() { timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict-fp32 --sustain 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
r=d['roofline']; print('$*', '|', d['value'], 'scenes/s frac', r['frac'], r['avg_launch_us'], 'us alg', r['algorithmic_bytes'], d['dtype'][:20], 'self_check', d['self_check']['mismatching'])"; }
one --storage f32
one --storage bf16
SGC_TILED_CM32=27,32,3,3,0 one --storage bf16
SGC_TILED_CM32=20,22,3,3,1 one --storage bf16
SGC_TILED_CM32=27,22,3,3,0 one --storage bf16
one --storage bf16 --workload cfg4_scannet200_large --steps 10 --warmup 3
SGC_TILED_CM16=40,30,3,3,1 one --storage bf16 --workload cfg4_scannet200_large --steps 10 --warmup 3

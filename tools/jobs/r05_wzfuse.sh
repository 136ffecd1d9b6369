#!/bin/bash
# round 5: output transform of the Winograd-z form fused into the convolution launch (last-arriving workgroup): parity, then A/B
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_conv3d.py -x -q -k "winograd" 2>&1 | tail -3
for rep in 1 2; do
for f in 0 1; do
SGC_TUNE=wz_fuse=$f timeout 600 python bench.py --no-cpu-baseline --no-strict-fp32 --steps 60 --warmup 15 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cfg2 wz_fuse $f:', d['value'], 'sustained', d['sustained']['value'], 'mfma', d['roofline_mfma']['frac'], d['roofline_mfma']['avg_launch_us'], 'self_check', d['self_check']['mismatching'])"
done
done
for f in 0 1; do
SGC_TUNE=wz_fuse=$f timeout 600 python bench.py --workload cfg3_arkit --no-cpu-baseline --no-strict-fp32 --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cfg3 wz_fuse $f:', d['value'], 'sustained', d['sustained']['value'], 'self_check', d['self_check']['mismatching'])"
SGC_TUNE=wz_fuse=$f timeout 600 python bench.py --streams 1 --no-cpu-baseline --no-strict-fp32 --steps 60 --warmup 15 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cfg2 one stream wz_fuse $f:', d['value'], 'self_check', d['self_check']['mismatching'])"
done

#!/bin/bash
# round 5: Winograd-z with the input transform fused into the halo staging (no transformed copy): parity + A/B off / on per workload
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_conv3d.py -x -q -k "winograd" 2>&1 | tail -3
python tools/winograd_cost.py 2>&1 | grep -v amdgpu.ids | head -3
for rep in 1 2; do
for wz in 0 1; do
SGC_WINOGRAD_Z=$wz timeout 600 python bench.py --no-cpu-baseline --no-strict-fp32 --steps 60 --warmup 15 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cfg2 winograd_z $wz:', d['value'], 'sustained', d['sustained']['value'], 'mfma', d['roofline_mfma']['frac'], d['roofline_mfma']['avg_launch_us'], 'self_check', d['self_check']['mismatching'])"
done
done
for wl in cfg3_arkit cfg5_arkit_large cfg4_scannet200_large; do
n=${wl%%_*}
for cfg in "0 256" "1 256" "1 128"; do
set -- $cfg
SGC_WINOGRAD_Z=$1 SGC_WINOGRAD_Z_MIN_CH=$2 timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-strict-fp32 --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n winograd_z $1 min_ch $2:', d['value'], 'sustained', d['sustained']['value'], 'self_check', d['self_check']['mismatching'])"
done
done

#!/bin/bash
cd $GRAFT_REPO_ROOT
for wl in cfg4_scannet200_large cfg5_arkit_large cfg3_arkit cfg2_scannet; do
timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-strict-fp32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$wl', d['value'], 'sustained', d['sustained']['value'], 'gather', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'path', d['path_roofline']['frac'], 'self_check', d['self_check']['mismatching'])"
done
SGC_TUNE="tile_ds=0" timeout 600 python bench.py --workload cfg4_scannet200_large --no-cpu-baseline --no-strict-fp32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cfg4 tile_ds=0', d['value'], 'gather', d['roofline']['frac'], d['roofline']['avg_launch_us'])"
SGC_TUNE="tile_ds=0" timeout 600 python bench.py --workload cfg5_arkit_large --no-cpu-baseline --no-strict-fp32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cfg5 tile_ds=0', d['value'], 'gather', d['roofline']['frac'], d['roofline']['avg_launch_us'])"

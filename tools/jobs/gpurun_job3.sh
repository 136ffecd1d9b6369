#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg4 -- python3 $R/bench.py --workload cfg4_scannet200_large --steps 12 --warmup 3 --no-cpu-baseline --streams 1 > $R/gpurun_out/bench_cfg4.json 2> $R/gpurun_out/bench_cfg4.err
cut -c1-200 $R/gpurun_out/bench_cfg4.json

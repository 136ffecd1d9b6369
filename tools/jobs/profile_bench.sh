#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v12 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/bench_prof_v12.json 2> $R/gpurun_out/bench_prof_v12.err
cd $R
cut -c1-250 gpurun_out/bench_prof_v12.json
timeout 300 python bench.py --steps 40 --warmup 6 2>/dev/null > gpurun_out/bench_v12.json; cut -c1-250 gpurun_out/bench_v12.json

#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for S in 1 2; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sg$S -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams $S > $R/gpurun_out/bench_sg$S.json 2> $R/gpurun_out/bench_sg$S.err
cut -c1-200 $R/gpurun_out/bench_sg$S.json
done

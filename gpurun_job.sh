#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v2 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench_prof_v2.json 2> $R/gpurun_out/bench_prof_v2.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc2_fetch -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc2_write -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R
cat gpurun_out/bench_prof_v2.json | cut -c1-200

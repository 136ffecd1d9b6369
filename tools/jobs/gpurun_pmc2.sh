#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
timeout 200 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_gather_$C -- python3 $R/bench.py --graph tail --streams 1 --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_gather_$C.log 2>&1
echo rc $?
done

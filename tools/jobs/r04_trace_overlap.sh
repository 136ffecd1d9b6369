#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ov
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ov -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-strict-fp32 --sustain 0 --prewarm 0.2 > /dev/null 2>&1; echo rc $?
t=$(find /tmp/prof_ov -name "*kernel_trace.csv" | head -1)
head -1 $t
python3 $R/tools/trace_overlap.py $t | tee $R/gpurun_out/r04_trace_overlap.txt

#!/bin/bash
mkdir -p gpurun_out
python __graft_entry__.py --smoke 2>&1 | tail -3 | tee gpurun_out/smoke.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof.err
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/bench_prof.err
cat gpurun_out/bench_prof.json
find gpurun_out/prof_r1 -name "*stats*" | head
f=$(find gpurun_out/prof_r1 -name "*kernel_stats.csv" | head -1); head -40 "$f"
for t in 16 32 64; do SGC_CPU_THREADS=$t python - <<PY
import os, sys, json
sys.path.insert(0, '.')
import bench
from sgcdet_amd.scene import workload
print(json.dumps(bench.cpu_baseline(workload('cfg2_scannet'), 40, 0)))
PY
done 2>&1 | grep -v Warn | tee gpurun_out/cpu_threads.log

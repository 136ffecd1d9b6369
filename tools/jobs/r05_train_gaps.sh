#!/bin/bash
# round 5: where the GPU idles inside a training step (host-bound stretches): kernel trace of tools/train_step_bench.py -> tools/trace_gaps.py
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tr
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tr -- python3 $R/tools/train_step_bench.py --steps 8 > $R/gpurun_out/r05_train_gaps_line.json 2>/dev/null; echo rc $?
t=$(find /tmp/prof_tr -name "*kernel_trace.csv" | head -1)
ms=$(python3 -c "import json;print(json.load(open('$R/gpurun_out/r05_train_gaps_line.json'))['ms_per_step']*4)")
python3 $R/tools/trace_gaps.py "$t" 4 $ms > $R/gpurun_out/r05_train_gaps.txt 2>&1
rm -rf /tmp/prof_tr

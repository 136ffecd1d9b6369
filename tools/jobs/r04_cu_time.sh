#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_cu
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_cu -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-strict-fp32 --sustain 0 --prewarm 0.2 > /tmp/bench_cu.json 2>/dev/null; echo rc $?
t=$(find /tmp/prof_cu -name "*kernel_trace.csv" | head -1)
n=$(python3 -c "import csv,sys; csv.field_size_limit(1<<30); print(sum(1 for r in csv.DictReader(open('$t')) if 'topk_select_kernel' in r['Kernel_Name'] or 'topk_hist' in r['Kernel_Name'] and False) )")
python3 $R/tools/trace_cu_time.py $t $(python3 -c "
import csv; csv.field_size_limit(1<<30)
print(sum(1 for r in csv.DictReader(open('$t')) if 'vox_count_kernel' in r['Kernel_Name']) / 3.0)") | tee $R/gpurun_out/r04_cu_time.txt

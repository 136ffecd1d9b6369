#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
r=d['roofline']; print(d['value'], 'scenes/s', d['ms_per_step'], 'ms  frac', r['frac'], r['avg_launch_us'], 'us  strict', d['strict_fp32']['value'], 'sustained', d['sustained']['value'], 'self_check', d['self_check']['mismatching'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02a -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict-fp32 --sustain 0 > $R/gpurun_out/bench_prof_r02a.json 2> $R/gpurun_out/bench_prof_r02a.err
cd $R
f=$(find gpurun_out/prof_r02a -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r02a_kernel_stats.csv
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r02a_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.2f}% calls {r['Calls']:>6} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
lib = sum(float(r["TotalDurationNs"]) for r in rows if "at::" in r["Name"] or "rocprim" in r["Name"] or "Cijk" in r["Name"])
print("library kernels share", lib / tot)
PY
rm -rf gpurun_out/prof_r02a

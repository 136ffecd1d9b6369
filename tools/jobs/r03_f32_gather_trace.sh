#!/bin/bash
# strict-fp32 mode: is the tiled gather really slower with four streams, or do the HIP events mislead?  kernel trace of the command
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_f32
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_f32 -- python3 $R/bench.py --conv-mode f32 --steps 8 --warmup 4 --no-cpu-baseline --sustain 0 > $R/gpurun_out/r03_f32_under_rocprof.json 2>/dev/null
t=$(find /tmp/prof_f32 -name "*kernel_trace.csv" | head -1)
if [ -n "$t" ]; then
python3 - "$t" <<'PY'
import csv, sys
csv.field_size_limit(1 << 30)
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dfa3d_fwd_tile_kernel" in r["Kernel_Name"]]
d = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
d.sort()
print("tile kernel launches", len(d), "durations us (in launch order):", [round(x[1], 1) for x in d])
PY
fi
rm -rf /tmp/prof_f32

#!/bin/bash
# per-launch-shape durations of the MFMA kernels in a bench run (kernel trace): which layers of the neck cost what
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_lt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_lt -- python3 $R/bench.py --steps ${SGC_LT_STEPS:-40} --warmup 10 --no-cpu-baseline --no-strict-fp32 --sustain 0 $SGC_LT_ARGS > /dev/null 2>&1
t=$(find /tmp/prof_lt -name "*kernel_trace.csv" | head -1)
if [ -n "$t" ]; then
python3 - "$t" <<'PY'
import csv, sys, collections, re, os
csv.field_size_limit(1 << 30)
by = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    if os.environ.get("SGC_LT_ALL") is None and not any(k in n for k in ("igemm", "halo", "rows_gemm", "conv_epilogue", "level_tail")):
        continue
    g = tuple(int(r[k]) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else (int(r.get("Grid_Size", 0)),)
    w = tuple(int(r[k]) for k in ("Workgroup_Size_X",)) if "Workgroup_Size_X" in r else ()
    by[(n, g, w)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in by.values())
for (n, g, w), d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    d.sort()
    if sum(d) / tot < 0.004: continue
    print(f"{n[:62]:62s} grid {str(g):22s} wg {str(w):8s} n {len(d):5d} avg {sum(d)/len(d):8.1f} med {d[len(d)//2]:8.1f} share {sum(d)/tot:6.3f}")
PY
fi
rm -rf /tmp/prof_lt

#!/bin/bash
# kernel-trace of a bench command: per-kernel shares.  usage: r06_trace.sh TAG WORKLOAD [extra bench args]
# writes gpurun_out/${TAG}_kernel_stats.csv, ${TAG}_kernels.json, ${TAG}_bench_line.json   (TAG must not start with rNN_bench_: that glob is the bench lines)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
TAG=${1:-r06_bench_cfg2}
WL=${2:-cfg2_scannet}
shift; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $R/bench.py --workload $WL --steps 40 --warmup 10 --no-cpu-baseline --no-strict-fp32 --sustain 0 "$@" > $R/gpurun_out/${TAG}_bench_line.json 2> /dev/null; echo rocprof $TAG rc $?
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
t=$(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/${TAG}_kernel_stats.csv; fi
if [ -n "$t" ]; then
python3 - "$t" > $R/gpurun_out/${TAG}_kernels.json <<'PY'
import csv, json, sys, re, collections
csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
out = {}
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
by = collections.defaultdict(list)
for r in rows:
    by[re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    if sum(d) * 1e3 / tot >= 0.003:
        d.sort()
        out[key] = dict(launches=len(d), avg_us=round(sum(d) / len(d), 2), median_us=round(d[len(d) // 2], 2), max_us=round(d[-1], 2),
                        share_of_gpu_time=round(sum(d) * 1e3 / tot, 4))
out["_all"] = dict(launches=len(rows), total_ms=round(tot / 1e6, 2))
print(json.dumps(out, indent=1))
PY
fi
rm -rf /tmp/prof_$TAG

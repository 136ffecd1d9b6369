#!/bin/bash
# round-3 evidence for profiles/: the default bench line, the other workloads, the strict-fp32 and opt-in bf16 lines,
# kernel stats + per-kernel trace summary of the default command, the halo-convolution clock, the row-GEMM A/B
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python bench.py > gpurun_out/r03_bench_cfg2.json 2> gpurun_out/r03_bench_cfg2.err; echo bench rc $?
timeout 600 python bench.py --workload cfg3_arkit --no-cpu-baseline > gpurun_out/r03_bench_cfg3.json 2>/dev/null; echo cfg3 rc $?
timeout 600 python bench.py --workload cfg4_scannet200_large --no-cpu-baseline > gpurun_out/r03_bench_cfg4.json 2>/dev/null; echo cfg4 rc $?
timeout 600 python bench.py --workload cfg5_arkit_large --no-cpu-baseline > gpurun_out/r03_bench_cfg5.json 2>/dev/null; echo cfg5 rc $?
timeout 600 python bench.py --conv-mode f32 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r03_bench_cfg2_f32.json 2>/dev/null; echo f32 rc $?
timeout 600 python bench.py --conv-mode bf16 --no-cpu-baseline > gpurun_out/r03_bench_cfg2_bf16.json 2>/dev/null; echo bf16 rc $?
timeout 600 python bench.py --conv-mode bf16 --workload cfg5_arkit_large --no-cpu-baseline > gpurun_out/r03_bench_cfg5_bf16.json 2>/dev/null; echo bf16-5 rc $?
timeout 600 python bench.py --input-layout nhwc --no-cpu-baseline > gpurun_out/r03_bench_cfg2_nhwc.json 2>/dev/null; echo nhwc rc $?
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_r03
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r03 -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-strict-fp32 --sustain 0 > $R/gpurun_out/r03_bench_cfg2_under_rocprof.json 2> /dev/null; echo rocprof rc $?
f=$(find /tmp/prof_r03 -name "*kernel_stats.csv" | head -1)
t=$(find /tmp/prof_r03 -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/r03_bench_cfg2_kernel_stats.csv; fi
if [ -n "$t" ]; then
python3 - "$t" > $R/gpurun_out/r03_kernels_from_trace.json <<'PY'
import csv, json, sys
csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
out = {}
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
import re, collections
by = collections.defaultdict(list)
for r in rows:
    by[re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    if sum(d) * 1e3 / tot >= 0.004:
        d.sort()
        out[key] = dict(launches=len(d), avg_us=round(sum(d) / len(d), 2), median_us=round(d[len(d) // 2], 2), max_us=round(d[-1], 2),
                        share_of_gpu_time=round(sum(d) * 1e3 / tot, 4))
out["_all"] = dict(launches=len(rows), total_ms=round(tot / 1e6, 2))
print(json.dumps(out, indent=1))
PY
fi
rm -rf /tmp/prof_r03
cd $R
timeout 300 python tools/halo_clock.py > gpurun_out/r03_halo_clock.txt 2>&1; echo halo clock rc $?
timeout 400 python tools/rows_gemm_check.py > gpurun_out/r03_rows_gemm_ab.log 2>&1; echo rows gemm rc $?
python - <<'PY'
import json
for n in ("cfg2", "cfg3", "cfg4", "cfg5", "cfg2_f32", "cfg2_bf16", "cfg5_bf16", "cfg2_nhwc"):
    try:
        d = json.loads(open(f"gpurun_out/r03_bench_{n}.json").readline())
        print(n, d["value"], d["ms_per_step"], "gather", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "mfma", (d.get("roofline_mfma") or {}).get("frac"),
              "path", (d.get("path_roofline") or {}).get("frac"), "strict", (d.get("strict_fp32") or {}).get("value"), "sustained", (d.get("sustained") or {}).get("value"),
              "self_check", d["self_check"]["mismatching"], "calls/scene", d["config"].get("library_calls_per_scene"))
    except Exception as e:
        print(n, "failed", e)
PY
cat gpurun_out/r03_kernels_from_trace.json

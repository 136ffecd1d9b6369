#!/bin/bash
# final round-2 evidence for profiles/: kernel stats + per-kernel trace summary of the default bench command, the default
# bench line itself, and the training-step breakdown
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python bench.py > gpurun_out/r02_bench_cfg2.json 2> gpurun_out/r02_bench_cfg2.err; echo bench rc $?
timeout 600 python bench.py --workload cfg4_scannet200_large --no-cpu-baseline > gpurun_out/r02_bench_cfg4.json 2> gpurun_out/r02_bench_cfg4.err; echo bench4 rc $?
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r02 -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-strict-fp32 --sustain 0 > $R/gpurun_out/r02_bench_cfg2_under_rocprof.json 2> $R/gpurun_out/r02_bench_under_rocprof.err; echo rocprof rc $?
f=$(find /tmp/prof_r02 -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r02_bench_cfg2_kernel_stats.csv
t=$(find /tmp/prof_r02 -name "*kernel_trace.csv" | head -1)
python3 - "$t" > $R/gpurun_out/r02_kernels_from_trace.json <<'PY'
import csv, json, sys
csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
out = {}
for key in ("dfa3d_fwd_tile_kernel", "conv3d_halo_bf16x3_kernel<4, 4, 16, true>", "conv3d_halo_bf16x3_kernel<4, 4, 16, false>",
            "conv3d_halo_bf16x3_kernel<4, 8, 8, false>", "conv3d_igemm_bf16x3_kernel<128, 4, 2>", "topk_select_kernel", "bin_place_kernel"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if key in r["Kernel_Name"]]
    if d:
        d.sort()
        out[key] = dict(launches=len(d), avg_us=round(sum(d) / len(d), 2), median_us=round(d[len(d) // 2], 2), max_us=round(d[-1], 2))
print(json.dumps(out, indent=1))
PY
rm -rf /tmp/prof_r02
cd $R
timeout 300 python tools/train_step_bench.py --steps 5 --profile > gpurun_out/r02_train_step.json 2> gpurun_out/r02_train_step_kernels.txt
SGC_TRAIN_CONV=library timeout 300 python tools/train_step_bench.py --steps 5 > gpurun_out/r02_train_step_library.json 2>/dev/null
cat gpurun_out/r02_kernels_from_trace.json; head -1 gpurun_out/r02_train_step.json; head -1 gpurun_out/r02_train_step_library.json
python - <<'PY'
import json
for n in ("cfg2", "cfg4"):
    d = json.loads(open(f"gpurun_out/r02_bench_{n}.json").readline())
    print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["achieved"], d.get("strict_fp32", {}).get("value"), d.get("sustained", {}).get("value"), d["self_check"]["mismatching"])
PY

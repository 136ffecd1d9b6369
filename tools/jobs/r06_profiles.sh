#!/bin/bash
# round-6 evidence for profiles/: the default bench line + the driver's 20-step command, the other workloads, kernel stats + trace
# summary of the default command and of cfg4 / cfg5, per-layer tables in both launch geometries, HBM-side traffic of the gather
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python bench.py > gpurun_out/r06_bench_cfg2.json 2> gpurun_out/r06_bench_cfg2.err; echo bench rc $?
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_cfg2_driver_cmd.json 2>/dev/null; echo driver-cmd rc $?
timeout 600 python bench.py --workload cfg3_arkit --no-cpu-baseline > gpurun_out/r06_bench_cfg3.json 2>/dev/null; echo cfg3 rc $?
timeout 600 python bench.py --workload cfg4_scannet200_large --no-cpu-baseline > gpurun_out/r06_bench_cfg4.json 2>/dev/null; echo cfg4 rc $?
timeout 600 python bench.py --workload cfg5_arkit_large --no-cpu-baseline > gpurun_out/r06_bench_cfg5.json 2>/dev/null; echo cfg5 rc $?
timeout 600 python bench.py --workload cfg2_scannet_100v --no-cpu-baseline > gpurun_out/r06_bench_cfg2_100v.json 2>/dev/null; echo cfg2-100v rc $?
timeout 600 python bench.py --input-layout nhwc --no-cpu-baseline > gpurun_out/r06_bench_cfg2_nhwc.json 2>/dev/null; echo nhwc rc $?
timeout 600 python bench.py --winograd off --no-cpu-baseline > gpurun_out/r06_bench_cfg2_direct.json 2>/dev/null; echo direct rc $?
timeout 600 python bench.py --conv-mode f32 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r06_bench_cfg2_f32.json 2>/dev/null; echo f32 rc $?
timeout 600 python bench.py --conv-mode fp16 --storage bf16 --no-cpu-baseline > gpurun_out/r06_bench_cfg2_fp16_bf16maps.json 2>/dev/null; echo fp16+storage rc $?
# per-layer tables: kernel-alone times in the geometry of the timed region (throughput) and at the latency-optimal splits
for geo in as-timed latency; do
timeout 600 python bench.py --breakdown --eager-geometry $geo --no-cpu-baseline --no-strict-fp32 --sustain 0 --steps 10 --warmup 3 > /dev/null 2> gpurun_out/r06_neck_layers_$geo.raw
grep -v "amdgpu.ids\|warn\|Warning" gpurun_out/r06_neck_layers_$geo.raw | cut -c1-160 > gpurun_out/r06_neck_layers_$geo.txt; rm -f gpurun_out/r06_neck_layers_$geo.raw
done
timeout 600 python bench.py --workload cfg5_arkit_large --breakdown --no-cpu-baseline --no-strict-fp32 --sustain 0 --steps 10 --warmup 3 > /dev/null 2> gpurun_out/r06_cfg5_breakdown.raw
grep -v "amdgpu.ids\|warn\|Warning" gpurun_out/r06_cfg5_breakdown.raw | cut -c1-160 > gpurun_out/r06_cfg5_breakdown.txt; rm -f gpurun_out/r06_cfg5_breakdown.raw
bash tools/jobs/r06_trace.sh r06_trace_cfg2 cfg2_scannet > /dev/null 2>&1; echo trace cfg2 rc $?
bash tools/jobs/r06_trace.sh r06_trace_cfg5 cfg5_arkit_large --steps 10 --warmup 3 > /dev/null 2>&1; echo trace cfg5 rc $?
bash tools/jobs/r06_trace.sh r06_trace_cfg4 cfg4_scannet200_large --steps 10 --warmup 3 > /dev/null 2>&1; echo trace cfg4 rc $?
bash tools/jobs/r06_trace.sh r06_trace_cfg2_100v cfg2_scannet_100v --steps 20 --warmup 5 > /dev/null 2>&1; echo trace cfg2-100v rc $?
# HBM-side traffic of the tiled gather, finest level of config 2 (one counter per pass; FETCH_SIZE is doubled per the gfx950 correction)
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_x
  timeout 400 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_x -- python3 $R/bench.py --graph tail --streams 1 --steps 6 --warmup 2 --no-cpu-baseline --no-strict-fp32 --sustain 0 > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmc_x dfa3d_fwd_tile_kernel 2 > $R/gpurun_out/r06_pmc_gather_${ctr}.json
done
cd $R
python3 - <<'PY' > gpurun_out/r06_gather_tile_pmc_hbm.json
import json
f = json.load(open("gpurun_out/r06_pmc_gather_FETCH_SIZE.json")); w = json.load(open("gpurun_out/r06_pmc_gather_WRITE_SIZE.json"))
fetch_kb, write_kb = f.get("FETCH_SIZE", 0.0), w.get("WRITE_SIZE", 0.0)
out = dict(kernel="sgc::dfa3d_fwd_tile_kernel, finest level of config 2 (tools/jobs/r06_profiles.sh)", FETCH_SIZE_KB=fetch_kb, WRITE_SIZE_KB=write_kb,
           hbm_bytes_per_launch=int((2 * fetch_kb + write_kb) * 1024), dispatches=f.get("_dispatches"),
           note="FETCH_SIZE doubled (gfx950: 128-byte requests tallied at 64 B), WRITE_SIZE as read; separate --pmc passes")
print(json.dumps(out, indent=1))
PY
timeout 600 python tools/train_step_bench.py --steps 40 2>/dev/null | tail -1 > gpurun_out/r06_train_step.json
SGC_TRAIN_BWD=0 timeout 600 python tools/train_step_bench.py --steps 40 2>/dev/null | tail -1 >> gpurun_out/r06_train_step.json
timeout 600 python tools/train_step_bench.py --steps 10 --profile >> gpurun_out/r06_train_step.json 2> gpurun_out/r06_train_step_kernels.raw; echo train rc $?
timeout 600 python tools/train_step_bench.py --steps 10 --glue 2>&1 | grep -v "amdgpu.ids\|warn\|Warning" | cut -c1-260 > gpurun_out/r06_train_step_glue.txt
grep -v "amdgpu.ids\|warn\|Warning" gpurun_out/r06_train_step_kernels.raw | cut -c1-200 > gpurun_out/r06_train_step_kernels.txt; rm -f gpurun_out/r06_train_step_kernels.raw
python - <<'PY'
import json
for n in ("cfg2", "cfg2_driver_cmd", "cfg2_100v", "cfg3", "cfg4", "cfg5", "cfg2_nhwc", "cfg2_direct", "cfg2_f32", "cfg2_fp16_bf16maps"):
    try:
        d = json.loads(open(f"gpurun_out/r06_bench_{n}.json").readline())
        print(n, d["value"], d["ms_per_scene"], "gather", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "mfma", (d.get("roofline_mfma") or {}).get("frac"), (d.get("roofline_mfma") or {}).get("avg_launch_us"),
              "path", (d.get("path_roofline") or {}).get("frac"), (d.get("path_roofline") or {}).get("frac_of_power_limited_floor"), "strict", (d.get("strict_fp32") or {}).get("value"), "sustained", (d.get("sustained") or {}).get("value"),
              "self_check", d["self_check"]["mismatching"], "calls/scene", d["config"].get("library_calls_per_scene"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(n, "failed", e)
PY
cat gpurun_out/r06_gather_tile_pmc_hbm.json gpurun_out/r06_train_step.json

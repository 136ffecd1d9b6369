#!/bin/bash
# round-4 evidence for profiles/: the default bench line + the driver's 20-step command, the other workloads, the strict-fp32 and
# opt-in bf16 / fp16 lines, kernel stats + per-kernel trace summary of the default command, SQ counters of the halo kernel,
# HBM-side traffic of the gather, the training step
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python bench.py > gpurun_out/r04_bench_cfg2.json 2> gpurun_out/r04_bench_cfg2.err; echo bench rc $?
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench_cfg2_driver_cmd.json 2>/dev/null; echo driver-cmd rc $?
timeout 600 python bench.py --workload cfg3_arkit --no-cpu-baseline > gpurun_out/r04_bench_cfg3.json 2>/dev/null; echo cfg3 rc $?
timeout 600 python bench.py --workload cfg4_scannet200_large --no-cpu-baseline > gpurun_out/r04_bench_cfg4.json 2>/dev/null; echo cfg4 rc $?
timeout 600 python bench.py --workload cfg5_arkit_large --no-cpu-baseline > gpurun_out/r04_bench_cfg5.json 2>/dev/null; echo cfg5 rc $?
timeout 600 python bench.py --conv-mode f32 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r04_bench_cfg2_f32.json 2>/dev/null; echo f32 rc $?
timeout 600 python bench.py --conv-mode bf16 --no-cpu-baseline > gpurun_out/r04_bench_cfg2_bf16.json 2>/dev/null; echo bf16 rc $?
timeout 600 python bench.py --conv-mode fp16 --no-cpu-baseline > gpurun_out/r04_bench_cfg2_fp16.json 2>/dev/null; echo fp16 rc $?
timeout 600 python bench.py --conv-mode fp16 --workload cfg5_arkit_large --no-cpu-baseline > gpurun_out/r04_bench_cfg5_fp16.json 2>/dev/null; echo fp16-5 rc $?
timeout 600 python bench.py --input-layout nhwc --no-cpu-baseline > gpurun_out/r04_bench_cfg2_nhwc.json 2>/dev/null; echo nhwc rc $?
# the reduced-precision twin of BASELINE configs #2 / #5: fp16 products + bf16 value map and depth maps in the tiled gather
timeout 600 python bench.py --conv-mode fp16 --storage bf16 --no-cpu-baseline > gpurun_out/r04_bench_cfg2_fp16_bf16maps.json 2>/dev/null; echo fp16+storage rc $?
timeout 600 python bench.py --conv-mode fp16 --storage bf16 --workload cfg5_arkit_large --no-cpu-baseline > gpurun_out/r04_bench_cfg5_fp16_bf16maps.json 2>/dev/null; echo fp16+storage-5 rc $?
bash tools/jobs/r04_trace.sh r04 > /dev/null 2>&1; echo trace rc $?
# SQ counters of the halo kernel (90-GF layer), separate passes
cd /tmp && export TMPDIR=/tmp
for part in a b; do
  if [ $part == a ]; then C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; else C="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; fi
  rm -rf /tmp/pmc_h$part
  timeout 150 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_h$part -- python3 $R/tools/conv_one.py 256 256 40 40 16 > /dev/null 2>&1; echo pmc halo $part rc $?
done
python3 - <<PY > $R/gpurun_out/r04_pmc_conv_halo.json
import json, subprocess
out = {}
for part in "ab":
    out.update(json.loads(subprocess.run(["python3", "$R/tools/pmc_summary.py", f"/tmp/pmc_h{part}", "conv3d_halo", "2"], capture_output=True, text=True).stdout))
print(json.dumps(out, indent=1))
PY
# HBM-side traffic of the tiled gather, finest level of config 2 (one counter per pass; FETCH_SIZE is doubled per the gfx950 correction)
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_x
  timeout 400 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_x -- python3 $R/bench.py --graph tail --streams 1 --steps 6 --warmup 2 --no-cpu-baseline --no-strict-fp32 --sustain 0 > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmc_x dfa3d_fwd_tile_kernel 2 > $R/gpurun_out/r04_pmc_gather_${ctr}.json
done
cd $R
python3 - <<'PY' > gpurun_out/r04_gather_tile_pmc_hbm.json
import json
f = json.load(open("gpurun_out/r04_pmc_gather_FETCH_SIZE.json")); w = json.load(open("gpurun_out/r04_pmc_gather_WRITE_SIZE.json"))
# the three levels launch the tiled kernel once each at config 2 only for the finest one (min_pixels): the mean is over its dispatches
fetch_kb, write_kb = f.get("FETCH_SIZE", 0.0), w.get("WRITE_SIZE", 0.0)
out = dict(kernel="sgc::dfa3d_fwd_tile_kernel, finest level of config 2 (tools/jobs/r04_profiles.sh)", FETCH_SIZE_KB=fetch_kb, WRITE_SIZE_KB=write_kb,
           hbm_bytes_per_launch=int((2 * fetch_kb + write_kb) * 1024), dispatches=f.get("_dispatches"),
           note="FETCH_SIZE doubled (gfx950: 128-byte requests tallied at 64 B), WRITE_SIZE as read; separate --pmc passes")
print(json.dumps(out, indent=1))
PY
timeout 600 python tools/train_step_bench.py --steps 10 --profile > gpurun_out/r04_train_step.json 2> gpurun_out/r04_train_step_kernels.raw; echo train rc $?
grep -v "amdgpu.ids\|warn\|Warning" gpurun_out/r04_train_step_kernels.raw | cut -c1-200 > gpurun_out/r04_train_step_kernels.txt; rm -f gpurun_out/r04_train_step_kernels.raw
timeout 300 python tools/wgrad_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_wgrad_ab.txt
timeout 300 python tools/halo_knob_ab.py halo_stagger 0,1 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_halo_schedule_ab.log
timeout 300 python tools/halo_fixed_cost.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_halo_fixed_cost.txt
timeout 300 python tools/small_grid_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_small_grid_ab.txt
python - <<'PY'
import json
for n in ("cfg2", "cfg2_driver_cmd", "cfg3", "cfg4", "cfg5", "cfg2_f32", "cfg2_bf16", "cfg2_fp16", "cfg5_fp16", "cfg2_nhwc"):
    try:
        d = json.loads(open(f"gpurun_out/r04_bench_{n}.json").readline())
        print(n, d["value"], d["ms_per_step"], "gather", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "mfma", (d.get("roofline_mfma") or {}).get("frac"),
              "path", (d.get("path_roofline") or {}).get("frac"), "strict", (d.get("strict_fp32") or {}).get("value"), "sustained", (d.get("sustained") or {}).get("value"),
              "self_check", d["self_check"]["mismatching"], "calls/scene", d["config"].get("library_calls_per_scene"))
    except Exception as e:
        print(n, "failed", e)
PY
cat gpurun_out/r04_pmc_conv_halo.json gpurun_out/r04_gather_tile_pmc_hbm.json gpurun_out/r04_train_step.json

#!/bin/bash
# round 5, first job: the headline on today's box (unchanged tree) + the evidence the round-4 review asked for on the C = 128
# configs (BASELINE cfg #4 / #5): per-kernel stats and the per-layer --breakdown table
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_base_cfg2_driver_cmd.json 2>/dev/null; echo base rc $?
for w in cfg5_arkit_large cfg4_scannet200_large cfg3_arkit; do
n=${w%%_*}
timeout 900 python bench.py --workload $w --no-cpu-baseline --no-strict-fp32 --breakdown --steps 20 --warmup 5 --sustain 0 > gpurun_out/r05_base_${n}.json 2> gpurun_out/r05_base_${n}_breakdown.raw; echo $w rc $?
grep -v "amdgpu.ids\|warn\|Warning" gpurun_out/r05_base_${n}_breakdown.raw | cut -c1-200 > gpurun_out/r05_base_${n}_breakdown.txt; rm -f gpurun_out/r05_base_${n}_breakdown.raw
done
bash tools/jobs/r05_trace.sh r05_base_cfg5 cfg5_arkit_large --steps 10 --warmup 3
bash tools/jobs/r05_trace.sh r05_base_cfg4 cfg4_scannet200_large --steps 10 --warmup 3
python - <<'PY'
import json
for n in ("cfg2_driver_cmd", "cfg3", "cfg4", "cfg5"):
    try:
        d = json.loads(open(f"gpurun_out/r05_base_{n}.json").readline())
        print(n, d["value"], d["ms_per_step"], "gather", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "mfma", (d.get("roofline_mfma") or {}).get("frac"),
              "path", (d.get("path_roofline") or {}).get("frac"), "sustained", (d.get("sustained") or {}).get("value"),
              "self_check", d["self_check"]["mismatching"], "calls/scene", d["config"].get("library_calls_per_scene"))
    except Exception as e:
        print(n, "failed", e)
PY

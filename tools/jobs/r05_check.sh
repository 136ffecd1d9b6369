#!/bin/bash
# full -m gpu suite + smoke + the driver's command on the current tree; cfg3 alone in the latency geometry (wave-quantisation split)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r05_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_check_driver_cmd.json 2>/dev/null; echo driver rc $?
timeout 600 python bench.py --workload cfg3_arkit --streams 1 --no-cpu-baseline --no-strict-fp32 --steps 40 --warmup 10 > gpurun_out/r05_cfg3_1stream_fix1.json 2>/dev/null
SGC_TUNE="halo_wave_fix=0" timeout 600 python bench.py --workload cfg3_arkit --streams 1 --no-cpu-baseline --no-strict-fp32 --steps 40 --warmup 10 > gpurun_out/r05_cfg3_1stream_fix0.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_check_driver_cmd.json").readline())
print("driver cmd", d["value"], "sustained", d["sustained"]["value"], "self_check", d["self_check"]["mismatching"])
for f in (0, 1):
    d = json.loads(open(f"gpurun_out/r05_cfg3_1stream_fix{f}.json").readline())
    print("cfg3 one stream, halo_wave_fix", f, d["value"], "mfma", d["roofline_mfma"]["frac"], d["roofline_mfma"]["avg_launch_us"], "self_check", d["self_check"]["mismatching"])
PY

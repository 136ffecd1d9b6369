#!/bin/bash
# quick state check: conv + kernel + module GPU tests, the driver command (20 steps x 4 scenes) and the default line (100 steps x 4 scenes)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench_20.json 2> gpurun_out/r04_bench_20.err; echo rc $?
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04_bench_400.json 2> gpurun_out/r04_bench_400.err; echo rc $?
python - <<'PY'
import json
for f in ("gpurun_out/r04_bench_20.json", "gpurun_out/r04_bench_400.json"):
    d = json.loads(open(f).readline())
    print(f, d["value"], d["ms_per_step"], d.get("sustained", {}).get("value"), d.get("path_roofline", {}).get("frac"), d["roofline"]["frac"], d.get("roofline_mfma"), d["self_check"])
PY

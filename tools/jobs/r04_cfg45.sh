#!/bin/bash
mkdir -p gpurun_out
for w in cfg4_scannet200_large cfg5_arkit_large cfg3_arkit; do
n=${w%%_*}
timeout 900 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r04_bench_$n.json 2>gpurun_out/r04_bench_$n.err; echo $w rc $?
done
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -x -q -k "tiled or full_view or hot_path" 2>&1 | tail -3
python - <<'PY'
import json
for n in ("cfg3", "cfg4", "cfg5"):
    d = json.loads(open(f"gpurun_out/r04_bench_{n}.json").readline())
    print(n, d["value"], "gather", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "path", (d.get("path_roofline") or {}).get("frac"), "sustained", (d.get("sustained") or {}).get("value"), d["self_check"]["mismatching"])
PY

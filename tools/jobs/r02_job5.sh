#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/valid_stats.py cfg2_scannet 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_modules.py -q -m gpu -k "full_view or full_size" 2>&1 | tail -6
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_full.json 2> gpurun_out/r02_bench_full.err; echo rc $?; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02_bench_full.json").read().strip().split("\n")[-1])
for k in ("value", "ms_per_step", "strict_fp32", "sustained", "self_check"):
    print(k, d.get(k))
print("roofline", {k: v for k, v in d["roofline"].items() if k not in ("measured",)})
print("cpu", json.dumps(d.get("cpu_baseline"))[:1500])
PY
tail -3 gpurun_out/r02_bench_full.err

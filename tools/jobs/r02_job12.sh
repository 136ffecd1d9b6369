#!/bin/bash
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
timeout 900 python bench.py --steps 20 --warmup 5 --breakdown > gpurun_out/r02_bench_b.json 2> gpurun_out/r02_bench_b.err
grep -E "^  sgc_(topk|bin_pairs|linear_rows_headmajor)" gpurun_out/r02_bench_b.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02_bench_b.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["strict_fp32"], d["sustained"], d["self_check"]["mismatching"])
print(json.dumps(d["roofline"])[:600])
print(json.dumps(d["cpu_baseline"])[:1200])
PY

#!/bin/bash
timeout 600 python bench.py --views 100 --no-cpu-baseline --no-strict-fp32 > gpurun_out/r02_bench_cfg2_100views.json 2>/dev/null; echo rc $?
timeout 600 python bench.py --storage bf16 --no-cpu-baseline --no-strict-fp32 > gpurun_out/r02_bench_cfg2_bf16_storage.json 2>/dev/null; echo rc $?
timeout 600 python bench.py --input-layout nhwc --no-cpu-baseline --no-strict-fp32 > gpurun_out/r02_bench_cfg2_nhwc.json 2>/dev/null; echo rc $?
timeout 900 python bench.py --workload cfg5_arkit_large --no-cpu-baseline --no-strict-fp32 > gpurun_out/r02_bench_cfg5.json 2>/dev/null; echo rc $?
timeout 600 python bench.py --workload cfg3_arkit --no-cpu-baseline --no-strict-fp32 > gpurun_out/r02_bench_cfg3.json 2>/dev/null; echo rc $?
python - <<'PY'
import json
for n in ("cfg2_100views", "cfg2_bf16_storage", "cfg2_nhwc", "cfg3", "cfg5"):
    try:
        d = json.loads(open(f"gpurun_out/r02_bench_{n}.json").readline())
        print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], round(d["roofline"]["achieved"]), d["self_check"]["mismatching"], d["sustained"]["value"], d["dtype"][:30])
    except Exception as e:
        print(n, "failed", e)
PY
